#!/usr/bin/env python3
"""One resident batch (8 MSMs, two in flight) for a kernel timeline: tools/batch_timeline.sh runs this under rocprofv3.
usage: batch_trace.py LOG_N [table|plain|COUNT] [NAME=VALUE ...]   (knobs are set in os.environ before the context is created;
round 2's form `LOG_N COUNT` = plain, COUNT MSMs per batch)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
lg, kind = int(sys.argv[1]), sys.argv[2] if len(sys.argv) > 2 else "plain"
B = 8
if kind.isdigit():
    B, kind = int(kind), "plain"
for kv in sys.argv[3:]:
    k, _, v = kv.partition("=")
    os.environ[k] = v
n = 1 << lg
d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
with th.HooksContext() as gen:
    gen.generate_device(31, 32, n, d_b.data_ptr(), d_s.data_ptr())
torch.cuda.synchronize()
hb = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
hs = d_s.cpu().pin_memory().numpy().view(np.uint32).reshape(n, 8)
with mh.MsmContext(flags=mh.FLAG_WINDOW_TABLE if kind == "table" else 0) as c:
    c.upload_bases(hb, mh.FORM_MONT)
    for _ in range(3):
        c.msm_resident_batch([hs] * B, want_affine=False)
    t = time.perf_counter()
    c.msm_resident_batch([hs] * B, want_affine=False)
    print("batch of %d: %.4f ms per MSM" % (B, (time.perf_counter() - t) * 1e3 / B))
