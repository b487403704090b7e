#!/usr/bin/env python3
"""Timing of skewed scalar distributions at N = 2^20 (robustness check; results in profiles/NOTES_r1.md)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch, mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build
n = 1 << 20
with mh.MsmContext() as c:
    c.set_stage_timing(True)
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    GEN.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr())
    s = d_s.cpu().numpy().view(np.uint32).reshape(n, 8)
    idx = np.arange(n)
    cases = [("uniform", s), ("all-equal", np.tile(s[:1], (n, 1))), ("2-distinct-interleaved", s[idx % 2]), ("3-distinct-interleaved", s[idx % 3]),
             ("3-distinct-blocked", s[(idx * 3) // n]), ("256-distinct", s[idx % 256]), ("small<2^32", np.pad(s[:, :1], ((0, 0), (0, 7))))]
    # a witness-like mix: 40 % zeros, 30 % ones, 10 % below 2^16, 20 % uniform
    rng = np.random.default_rng(7)
    u = rng.random(n)
    mix = s.copy()
    mix[u < 0.8] = 0
    mix[(u >= 0.4) & (u < 0.7), 0] = 1
    sel = (u >= 0.7) & (u < 0.8)
    mix[sel, 0] = s[sel, 0] & 0xFFFF
    cases.append(("witness-like mix", mix))
    for label, arr in cases:
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int32).reshape(-1).copy()).cuda()
        for _ in range(2): c.msm_device(d_b.data_ptr(), t.data_ptr(), n)
        t0 = time.perf_counter(); r = c.msm_device(d_b.data_ptr(), t.data_ptr(), n); dt = (time.perf_counter() - t0) * 1e3
        print(f"{label:26s} {dt:7.2f} ms", {k: round(v, 2) for k, v in c.timings().items() if k.endswith("_ms") and v > 0.005}, flush=True)
