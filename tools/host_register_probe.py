#!/usr/bin/env python3
"""What does pinning the caller's pageable memory in place cost (hipHostRegister / hipHostUnregister per call)?  If it were cheap, a pageable host call could
register, copy at the pinned rate (53-55 GB/s instead of ~40) and unregister.  Round 6 probe; result in profiles/NOTES_r6.md."""
import time
import numpy as np
import torch

rt = torch.cuda.cudart()
torch.cuda.init()
for mb in (32, 64, 96):
    a = np.ones(mb << 20, np.uint8)
    ts, tu = [], []
    for _ in range(7):
        t0 = time.perf_counter(); rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); t1 = time.perf_counter()
        rc2 = rt.cudaHostUnregister(a.ctypes.data); t2 = time.perf_counter()
        ts.append((t1 - t0) * 1e3); tu.append((t2 - t1) * 1e3)
    print(f"{mb} MB: register median {sorted(ts)[3]:.3f} ms (min {min(ts):.3f}), unregister median {sorted(tu)[3]:.3f} ms, rc {rc} {rc2}", flush=True)

# ... and does a registered pageable buffer then travel at the pinned rate?  The 2^20 host call on plain numpy arrays, on the same arrays registered in
# place, and on torch pinned memory
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th

gen = th.HooksContext()
n = 1 << 20
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
tb, ts_ = d_b.cpu(), d_s.cpu()
hb, hs = tb.numpy().view(np.uint32).reshape(n, 16).copy(), ts_.numpy().view(np.uint32).reshape(n, 8).copy()
pb, ps = tb.pin_memory(), ts_.pin_memory()
hbp, hsp = pb.numpy().view(np.uint32).reshape(n, 16), ps.numpy().view(np.uint32).reshape(n, 8)


def timed(fn, reps=9):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.15:
        fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append((time.perf_counter() - t0) * 1e3)
    return sorted(t)[len(t) // 2]


with mh.MsmContext() as c:
    for rnd in range(2):
        print(f"2^20 pageable numpy            {timed(lambda: c.msm(hb, hs, mh.FORM_MONT)):.3f} ms", flush=True)

        def registered():
            rt.cudaHostRegister(hb.ctypes.data, hb.nbytes, 0); rt.cudaHostRegister(hs.ctypes.data, hs.nbytes, 0)
            r = c.msm(hb, hs, mh.FORM_MONT)
            rt.cudaHostUnregister(hb.ctypes.data); rt.cudaHostUnregister(hs.ctypes.data)
            return r
        print(f"2^20 registered per call       {timed(registered):.3f} ms", flush=True)
        rt.cudaHostRegister(hb.ctypes.data, hb.nbytes, 0); rt.cudaHostRegister(hs.ctypes.data, hs.nbytes, 0)
        print(f"2^20 registered once           {timed(lambda: c.msm(hb, hs, mh.FORM_MONT)):.3f} ms", flush=True)
        rt.cudaHostUnregister(hb.ctypes.data); rt.cudaHostUnregister(hs.ctypes.data)
        print(f"2^20 torch pinned              {timed(lambda: c.msm(hbp, hsp, mh.FORM_MONT)):.3f} ms", flush=True)
