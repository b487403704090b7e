"""Per-step time of the 2^20 MSM after the GPU sat idle (0.5 s): how many steps does the clock ramp take?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
hk = th.HooksContext()
n = 1 << 20
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
hk.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
c = mh.MsmContext(max_points=n)
for idle in (0.5, 0.05, 0.005):
    for rep in range(2):
        time.sleep(idle)
        ts = []
        for _ in range(120):
            t0 = time.perf_counter(); c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"idle {idle:5.3f}s: steps 1-5 " + " ".join(f"{t:.3f}" for t in ts[:5]) + " | 6-10 avg %.3f | 11-20 avg %.3f | 21-40 %.3f | 41-80 %.3f | 81-120 %.3f"
              % (sum(ts[5:10]) / 5, sum(ts[10:20]) / 10, sum(ts[20:40]) / 20, sum(ts[40:80]) / 40, sum(ts[80:]) / 40), flush=True)
