"""Which process-level state changes the kernels' speed?  Calibration peaks (k_calibrate, same ISA every time) and a 2^20 MSM after
each step: hooks context alone, + a product context, + torch tensors, ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
hk = th.HooksContext()
def cal(tag):
    for _ in range(2): mad, fpm = hk.calibrate()
    print(f"{tag:40s} mad {mad/1e9:9.1f} G/s  fp_mul {fpm/1e9:7.2f} G/s", flush=True)
cal("hooks context only")
import torch
x = torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
cal("+ torch initialised")
n = 1 << 20
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
hk.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
cal("+ instance generated")
def msm(ctx, tag):
    for _ in range(5): ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    t0 = time.perf_counter()
    for _ in range(20): ctx.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    ms = (time.perf_counter() - t0) * 1e3 / 20
    print(f"{tag:40s} msm 2^20 {ms:.4f} ms  acc {ctx.timings()['accumulate_ms']:.4f}", flush=True)
msm(hk, "hooks context msm")
cal("after hooks msm")
c = mh.MsmContext()
c.set_kernel_timing(1)
cal("+ product context created")
msm(c, "product context msm")
cal("after product msm")
msm(hk, "hooks context msm again")
c.close()
cal("product context closed")
msm(hk, "hooks context msm, product closed")
