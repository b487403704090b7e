import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpu-acceleration_amd')]
import numpy as np, torch, mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build
for logn, wb, fl in ((16, 16, mh.FLAG_UNSIGNED_DIGITS), (16, 0, 0), (20, 16, mh.FLAG_UNSIGNED_DIGITS), (20, 17, 0)):
    n = 1 << logn
    c = mh.MsmContext(window_bits=wb, flags=fl)
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda")
    GEN.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    for _ in range(3): r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    t0 = time.perf_counter()
    for _ in range(20): r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    print(f"2^{logn} window_bits={wb} flags={fl}: {(time.perf_counter()-t0)/20*1e3:.3f} ms")
    c.close()
