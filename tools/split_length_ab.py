#!/usr/bin/env python3
"""Round 6 (NOTES_r6 section 14): the run length of very long buckets (make_piece_plan: psplit) A/B through MSM_HIP_SPLIT_TARGET (hooks build), whole calls on
skewed scalars at 2^20, contexts side by side on one box: python3 tools/split_length_ab.py 524288 385000 262144   (pieces aimed at: pairs / target = length)"""
import os, sys, time
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
cases = [("uniform", b, s), ("all-equal", b, np.tile(s[:1], (n, 1))), ("2-distinct", b, s[idx % 2]), ("3-distinct", b, s[idx % 3]), ("3-blocked", b, s[(idx * 3) // n]), ("256-distinct", b, s[idx % 256])]
devs = [(l, dev(bb), dev(ss)) for l, bb, ss in cases]
ctxs = {}
for tgt in sys.argv[1:]:
    os.environ["MSM_HIP_SPLIT_TARGET"] = tgt
    ctxs[tgt] = th.HooksContext()
for rnd in range(3):
    for label, tb, ts_ in devs:
        row = []
        for tgt, c in ctxs.items():
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.1:
                c.msm_device(tb.data_ptr(), ts_.data_ptr(), n)
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                c.msm_device(tb.data_ptr(), ts_.data_ptr(), n)
                ts.append((time.perf_counter() - t0) * 1e3)
            row.append(f"{tgt}: {sorted(ts)[7]:.3f}")
        print(rnd, f"{label:14s}", "  ".join(row), flush=True)
