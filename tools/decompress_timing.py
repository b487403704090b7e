#!/usr/bin/env python3
"""Row f3 timing: arkworks compressed images -> resident bases on the GPU vs the CPU oracle, and the file harness."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build
from mopro_msm_hip import instances as inst
from oracle import bn254_oracle as orc

ctx = mh.MsmContext()
for logn in (16, 20, 22):
    n = 1 << logn
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda")
    GEN.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    bases = d_b.cpu().numpy().view(np.uint32).reshape(n, 16); scalars = d_s.cpu().numpy().view(np.uint32).reshape(n, 8)
    t0 = time.perf_counter(); img = mh.compress_points(bases, mh.FORM_MONT); t_c = time.perf_counter() - t0
    ctx.upload_compressed(img)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.upload_compressed(img); ts.append(time.perf_counter() - t0)
    t_up = min(ts)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.upload_bases(bases, mh.FORM_MONT); ts.append(time.perf_counter() - t0)
    t_raw = min(ts)
    m = min(n, 1 << 14)
    t0 = time.perf_counter(); orc.g1_decompress(img[: 32 * m]); t_o = (time.perf_counter() - t0) / m * n
    ctx.upload_compressed(img)
    r1 = ctx.msm_resident(scalars); r2 = ctx.msm(bases, scalars, mh.FORM_MONT)
    assert (r1.affine_std == r2.affine_std).all()
    print(f"n=2^{logn}: host compress {t_c*1e3:.1f} ms | upload_compressed (H2D 32 B/pt + GPU sqrt) {t_up*1e3:.2f} ms | "
          f"upload_bases (H2D 64 B/pt + convert) {t_raw*1e3:.2f} ms | CPU oracle decompress, 1 thread (extrapolated from {m}) {t_o*1e3:.0f} ms")
with tempfile.TemporaryDirectory() as d:
    res = []
    for size in (8, 12, 16, 18, 20):
        p = os.path.join(d, f"{size}x4")
        t0 = time.perf_counter(); r = inst.run_benchmark(size, 4, p, ctx); t = time.perf_counter() - t0
        res.append(r); print("run_benchmark", r, f"(whole call incl. generating vectors {t:.2f} s)")
    inst.write_csv(os.path.join(ROOT, "gpurun_out", "hip_benchmark.txt"), res)
