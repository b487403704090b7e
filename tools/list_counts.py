#!/usr/bin/env python3
"""Round 6 (NOTES_r6 section 14): list counters (long / mid / two-piece buckets, pieces, partial sums) and the combine / accumulate / plan stage times of skewed cases
at 2^20, for one or more MSM_HIP_MID_LANE_MIN values (hooks build): python3 tools/list_counts.py 32768 0 1000000000"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from mopro_msm_hip import testhooks as th
n = 1 << 20
def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32).reshape(-1).copy()).cuda()
with th.HooksContext() as c:
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    c.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    s = d_s.cpu().numpy().view(np.uint32).reshape(n, 8)
    b = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
idx = np.arange(n)
cases = [("uniform", b, s), ("all-equal", b, np.tile(s[:1], (n, 1))), ("3-distinct", b, s[idx % 3]), ("256-distinct", b, s[idx % 256])]
for T in (8, 32, 128):
    L = n // T
    cases.append((f"T={T}", np.tile(b[:L], (T, 1)), np.tile(s[:L], (T, 1))))
for mlm in sys.argv[1:] or ["32768"]:
    os.environ["MSM_HIP_MID_LANE_MIN"] = mlm
    with th.HooksContext() as c:
        for label, bb, ss in cases:
            tb, ts_ = dev(bb), dev(ss)
            for _ in range(30):
                c.msm_device(tb.data_ptr(), ts_.data_ptr(), n)
            c.set_stage_timing(True)
            cm = []
            for _ in range(5):
                c.msm_device(tb.data_ptr(), ts_.data_ptr(), n)
                t = c.timings()
                cm.append((round(t["combine_ms"], 4), round(t["accumulate_ms"], 4), round(t["plan_ms"], 4)))
            c.set_stage_timing(False)
            print("mid_lane_min", mlm, label, c.list_counts(), "combine/acc/plan", sorted(cm)[2], flush=True)
