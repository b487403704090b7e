#!/usr/bin/env python3
"""Per-call latency distribution of msm_device (resident operands): min / median / p90 / p99 / max over many calls."""
import os, sys, time
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build
for logn in [int(x) for x in (sys.argv[1:] or ["16", "20"])]:
    n = 1 << logn
    ctx = mh.MsmContext(max_points=n)
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda")
    GEN.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    for _ in range(5): ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    ts = []
    for _ in range(int(os.environ.get("JITTER_CALLS", "400"))):
        t0 = time.perf_counter(); ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
    a = np.sort(np.array(ts))
    print(f"n=2^{logn}: min {a[0]:.3f} median {a[len(a)//2]:.3f} mean {a.mean():.3f} p90 {a[int(len(a)*0.9)]:.3f} p99 {a[int(len(a)*0.99)]:.3f} max {a[-1]:.3f} ms; calls > 1.5x median: {(a > 1.5 * a[len(a)//2]).sum()}")
    ctx.close()
