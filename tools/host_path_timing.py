#!/usr/bin/env python3
"""Time the HOST-pointer entry (msm_bn254_g1: PCIe included) single-shot vs streamed, for DESIGN.md section 7."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build
for logn in (20, 22):
    n = 1 << logn
    with mh.MsmContext(stream_chunk_log2=28) as c0:  # never streams
        d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
        GEN.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr())
        hb = d_b.cpu().numpy().view(np.uint32).reshape(n, 16); hs = d_s.cpu().numpy().view(np.uint32).reshape(n, 8)
        dev = c0.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
        for _ in range(2): r0 = c0.msm(hb, hs, mh.FORM_MONT)
        t = time.perf_counter(); r0 = c0.msm(hb, hs, mh.FORM_MONT); t_single = (time.perf_counter() - t) * 1e3
        t = time.perf_counter(); c0.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); t_dev = (time.perf_counter() - t) * 1e3
    for lg in (18, 19, 20):
        if n < 2 << lg: continue
        with mh.MsmContext(stream_chunk_log2=lg) as c1:
            for _ in range(2): r1 = c1.msm(hb, hs, mh.FORM_MONT)
            t = time.perf_counter(); r1 = c1.msm(hb, hs, mh.FORM_MONT); t_str = (time.perf_counter() - t) * 1e3
        assert (r1.affine_std == dev.affine_std).all() and (r0.affine_std == dev.affine_std).all()
        print(f"N=2^{logn}: resident {t_dev:.2f} ms | host single-shot {t_single:.2f} ms | host streamed 2^{lg}-point chunks {t_str:.2f} ms", flush=True)
