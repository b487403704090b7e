#!/usr/bin/env python3
"""Soak run: thousands of calls of mixed sizes on one context -- results must repeat bit for bit and neither host nor
device memory may grow (workspace is grow-only up to the largest size, then constant)."""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
nmax = 1 << 18
ctx = mh.MsmContext()
d_b = torch.empty(nmax * 16, dtype=torch.int32, device="cuda"); d_s = torch.empty(nmax * 8, dtype=torch.int32, device="cuda")
GEN.generate_device(11, 12, nmax, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
sizes = [1, 17, 1000, 4096, 30000, 1 << 16, 100003, nmax]
ref = {n: ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n).jacobian_mont.copy() for n in sizes}
ref_aff = {n: mh.combine_partials(ref[n].reshape(1, 24)).affine_std.copy() for n in sizes}
free0, _ = torch.cuda.mem_get_info(); rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
rng = np.random.default_rng(5)
t0 = time.time(); bad = 0
for i in range(calls):
    n = sizes[int(rng.integers(0, len(sizes)))]
    r = ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    if not (r.affine_std == ref_aff[n]).all(): bad += 1
free1, _ = torch.cuda.mem_get_info(); rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print(f"soak: {calls} calls in {time.time()-t0:.1f} s, mismatches {bad}, device memory delta {(free0-free1)/2**20:.1f} MiB, host max-RSS delta {(rss1-rss0)/1024:.1f} MiB")
sys.exit(1 if bad else 0)
