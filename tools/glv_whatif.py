#!/usr/bin/env python3
"""What would GLV buy?  Emulates its pipeline shape -- 2N points with scalars below 2^126 in 8 windows of 16 bits -- on the
existing kernels (MSM_HIP_EXPERIMENT_WINDOWS=8) and compares with the normal N-point, 16-window run.  The emulation pays a
2N-point conversion and 2N-scalar decomposition that real GLV would not (~+60 us at N = 2^20)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build
def med(ctx, b, s, n, reps=60):
    for _ in range(5): ctx.msm_device(b.data_ptr(), s.data_ptr(), n)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ctx.msm_device(b.data_ptr(), s.data_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
    ctx.set_stage_timing(True); ctx.msm_device(b.data_ptr(), s.data_ptr(), n); tm = ctx.timings(); ctx.set_stage_timing(False)
    return float(np.median(ts)), {k: round(v, 3) for k, v in tm.items() if k.endswith("_ms")}
for logn in [int(x) for x in (sys.argv[1:] or ["17", "20"])]:
    n = 1 << logn
    d_b = torch.empty(2 * n * 16, dtype=torch.int32, device="cuda"); d_s = torch.empty(2 * n * 8, dtype=torch.int32, device="cuda")
    os.environ.pop("MSM_HIP_EXPERIMENT_WINDOWS", None)
    ctx = mh.MsmContext(window_bits=16 if logn >= 18 else 0)
    GEN.generate_device(1, 2, 2 * n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    base = med(ctx, d_b, d_s, n); ctx.close()
    v = d_s.view(2 * n, 8); v[:, 4:] = 0; v[:, 3] &= 0x3FFFFFFF; torch.cuda.synchronize()   # scalars < 2^126
    for c in ([16] if logn >= 18 else [13, 15, 16]):
        W = 126 // c + 1
        os.environ["MSM_HIP_EXPERIMENT_WINDOWS"] = str(W)
        ctx = mh.MsmContext(window_bits=c)
        r = med(ctx, d_b, d_s, 2 * n); ctx.close()
        print(f"N=2^{logn}: today {base[0]:.3f} ms {base[1]}\n   GLV shape c={c} W={W}: {r[0]:.3f} ms {r[1]}")
