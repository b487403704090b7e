#!/usr/bin/env python3
"""Round 6 (NOTES_r6 section 14): the run length of very long buckets A/B through knobs of the hooks build, whole calls on skewed scalars at 2^20, one context per
setting side by side on one box:   python3 tools/split_length_ab.py MSM_HIP_SPLIT_TARGET=524288 MSM_HIP_SPLIT_TARGET=262144      (host plan: pairs / target = length)
                                   python3 tools/split_length_ab.py MSM_HIP_SPLIT_SHIFT=31 MSM_HIP_SPLIT_SHIFT=16 MSM_HIP_SPLIT_SHIFT=17   (device: entries >> shift; 31 = off)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from mopro_msm_hip import testhooks as th
n = 1 << int(os.environ.get("AB_LOGN", "20"))  # (AB_LOGN: another size)
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
cases = [("uniform", b, s), ("all-equal", b, np.tile(s[:1], (n, 1))), ("2-distinct", b, s[idx % 2]), ("3-distinct", b, s[idx % 3]), ("3-blocked", b, s[(idx * 3) // n]), ("256-distinct", b, s[idx % 256]),
         ("small<2^32", b, np.pad(s[:, :1], ((0, 0), (0, 7))))]
rng = np.random.default_rng(7)  # the witness-like mix of tools/adversarial_timing.py: 40 % zeros, 30 % ones, 10 % below 2^16, 20 % uniform
u = rng.random(n)
mix = s.copy()
mix[u < 0.8] = 0
mix[(u >= 0.4) & (u < 0.7), 0] = 1
sel = (u >= 0.7) & (u < 0.8)
mix[sel, 0] = s[sel, 0] & 0xFFFF
cases.append(("witness-like", b, mix))
for T in (8, 32, 128):
    L = n // T
    cases.append((f"fixture T={T}", np.tile(b[:L], (T, 1)), np.tile(s[:L], (T, 1))))
devs = [(l, dev(bb), dev(ss)) for l, bb, ss in cases]
ctxs = {}
for setting in sys.argv[1:]:
    name, val = setting.split("=", 1)
    os.environ[name] = val
    ctxs[setting.replace("MSM_HIP_", "")] = th.HooksContext()
    del os.environ[name]
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
