"""One resident batch under rocprofv3 (--kernel-trace --memory-copy-trace): the timeline of the two pipelines.  argv: log2(n) count"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << lg
gen = th.HooksContext()
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
gen.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
c = mh.MsmContext()
c.upload_bases(d_b.cpu().numpy().view(np.uint32).reshape(n, 16), mh.FORM_MONT)
hs = d_s.cpu().pin_memory().numpy().view(np.uint32).reshape(n, 8)
for _ in range(2): c.msm_resident_batch([hs] * 4)
ts = []
for _ in range(8):
    t0 = time.perf_counter(); c.msm_resident(hs); ts.append((time.perf_counter() - t0) * 1e3)
print("single calls ms:", " ".join(f"{t:.3f}" for t in ts))
t0 = time.perf_counter()
c.msm_resident_batch([hs] * B)
print(f"batch of {B}: {(time.perf_counter() - t0) * 1e3 / B:.4f} ms per MSM")
c.close(); gen.close()
