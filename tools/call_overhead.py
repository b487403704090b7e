#!/usr/bin/env python3
"""Python-side overhead of one msm_device call: Python wall clock minus the C ABI's own wall clock (msm_timings_t.total_ms)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build
for logn in (12, 16, 20):
    n = 1 << logn
    ctx = mh.MsmContext(max_points=n)
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda")
    GEN.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    for _ in range(5): ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    py, cc, fin = [], [], []
    for _ in range(200):
        t0 = time.perf_counter(); ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); py.append((time.perf_counter() - t0) * 1e3)
        tm = ctx.timings(); cc.append(tm["total_ms"]); fin.append(tm["finish_ms"])
    print(f"n=2^{logn}: python wall median {np.median(py):.4f} ms, C ABI wall median {np.median(cc):.4f} ms, host finish {np.median(fin)*1e3:.1f} us, "
          f"python overhead {1e3*(np.median(py)-np.median(cc)):.1f} us")
    ctx.close()
