#!/usr/bin/env python3
"""Does the mere EXISTENCE of other contexts (their streams = HSA queues) slow a context's resident batch?  K idle contexts are created
(each with its second pipeline, made by one small batch call), then a fresh context runs the batch at 2^LOG_N; K = 0, 1, 2, 3, 5, 8.
usage: tools/idle_contexts_probe.py [log_n] [table|plain]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
flags = mh.FLAG_WINDOW_TABLE if (len(sys.argv) > 2 and sys.argv[2] == "table") else 0
n = 1 << lg
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
with th.HooksContext() as gen:
    gen.generate_device(61, 62, n, d_b.data_ptr(), d_s.data_ptr())
torch.cuda.synchronize()
hb = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
hs = d_s.cpu().pin_memory().numpy().view(np.uint32).reshape(n, 8)
small_b, small_s = hb[:4096], hs[:4096]
idle = []
for K in (0, 1, 2, 3, 5, 8):
    while len(idle) < K:
        c = mh.MsmContext()
        c.upload_bases(small_b, mh.FORM_MONT)
        c.msm_resident_batch([small_s, small_s])  # creates its second pipeline
        idle.append(c)
    with mh.MsmContext(flags=flags) as c:
        c.upload_bases(hb, mh.FORM_MONT)
        for _ in range(3): c.msm_resident_batch([hs] * 8, want_affine=False)
        best = 1e9
        for _ in range(4):
            t = time.perf_counter(); c.msm_resident_batch([hs] * 8, want_affine=False); best = min(best, (time.perf_counter() - t) * 1e3 / 8)
        ts = []
        for _ in range(10):
            t = time.perf_counter(); c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t) * 1e3)
    print(f"2^{lg} {'table' if flags else 'plain'}: {K} idle contexts alive: batch {best:.4f} ms per MSM, device call {sorted(ts)[len(ts)//2]:.4f} ms", flush=True)
for c in idle: c.close()
