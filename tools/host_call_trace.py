"""A few host-pointer calls at 2^20 from pinned (pull / copy) and pageable memory -- run under `rocprofv3 --kernel-trace --stats`
to see which transfers run as kernels (blit) and which on the DMA engines."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
mode = sys.argv[1]
n = 1 << 20
gen = th.HooksContext()
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
tb, ts = d_b.cpu(), d_s.cpu()
if mode != "pageable":
    tb, ts = tb.pin_memory(), ts.pin_memory()
hb, hs = tb.numpy().view(np.uint32).reshape(n, 16), ts.numpy().view(np.uint32).reshape(n, 8)
with mh.MsmContext() as c:
    for _ in range(6):
        c.msm(hb, hs, mh.FORM_MONT)
    print(mode, c.timings())
