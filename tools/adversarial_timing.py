#!/usr/bin/env python3
"""Timing of skewed scalar distributions and of the reference's fixture shape at N = 2^20, beside uniform scalars, every case after the clock ramp
(150 ms of back-to-back calls before the timed ones, median of 15).  Rows:
  scalar skew     all equal / 2, 3, 256 distinct / small / a witness-like mix, on distinct bases k_i * G
  reference shape test_utils::generate_random_bases_and_scalars (metal_msm.rs:706-730): one (base, scalar) sequence repeated T = 8 / 128 times --
                  every bucket holds T copies of each of its points (P + P in the accumulation); and the 63 points the reference tree holds, tiled
Prints ms per MSM, the ratio to uniform and the stage times of one extra call with stage events on."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th

GEN = th.HooksContext()  # the synthetic-instance generator lives in the hooks build
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ONLY = sys.argv[2] if len(sys.argv) > 2 else None  # substring of a row's label: that row alone (to profile one case: rocprofv3 ... -- python3 tools/adversarial_timing.py 20 T=128)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32).reshape(-1).copy()).cuda()


def timed(c, d_b, d_s):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.15:  # clock ramp
        c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


with mh.MsmContext() as c:
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0")
    d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    GEN.generate_device(1, 2, n, d_b.data_ptr(), d_s.data_ptr())
    torch.cuda.synchronize()
    s = d_s.cpu().numpy().view(np.uint32).reshape(n, 8)
    b = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
    idx = np.arange(n)
    cases = [("uniform", b, s), ("all-equal", b, np.tile(s[:1], (n, 1))), ("2-distinct-interleaved", b, s[idx % 2]), ("3-distinct-interleaved", b, s[idx % 3]),
             ("3-distinct-blocked", b, s[(idx * 3) // n]), ("256-distinct", b, s[idx % 256]), ("small<2^32", b, np.pad(s[:, :1], ((0, 0), (0, 7))))]
    # a witness-like mix: 40 % zeros, 30 % ones, 10 % below 2^16, 20 % uniform
    rng = np.random.default_rng(7)
    u = rng.random(n)
    mix = s.copy()
    mix[u < 0.8] = 0
    mix[(u >= 0.4) & (u < 0.7), 0] = 1
    sel = (u >= 0.7) & (u < 0.8)
    mix[sel, 0] = s[sel, 0] & 0xFFFF
    cases.append(("witness-like mix", b, mix))
    for T in (8, 128):  # the reference's fixture shape: the first n / T pairs repeated T times
        L = n // T
        cases.append((f"reference fixture T={T}", np.tile(b[:L], (T, 1)), np.tile(s[:L], (T, 1))))
    try:
        from conftest import load_srs_sets, load_zkey_points
        zb, zinf, _, _, _ = load_zkey_points()
        pts = [zb[i] for i in range(len(zb)) if not zinf[i]]
        for _f, _k, _om, g, gl in load_srs_sets():
            pts += list(g) + list(gl)
        pts = np.stack(pts).astype(np.uint32)
        cases.append(("63 reference-held points tiled", np.ascontiguousarray(pts[idx % len(pts)]), s))
    except Exception as e:  # noqa: BLE001
        print("reference-held points not available:", e)
    base = None
    for label, bb, ss in cases:
        if ONLY and ONLY not in label:
            continue
        tb, ts_ = dev(bb), dev(ss)
        c.set_stage_timing(False)
        ms = timed(c, tb, ts_)
        if base is None:
            base = ms
        c.set_stage_timing(True)
        c.msm_device(tb.data_ptr(), ts_.data_ptr(), n)
        c.msm_device(tb.data_ptr(), ts_.data_ptr(), n)
        st = {k: round(v, 3) for k, v in c.timings().items() if k.endswith("_ms") and v > 0.005 and k != "total_ms"}
        print(f"{label:32s} {ms:7.3f} ms  x{ms / base:5.3f} of uniform  {st}", flush=True)
        del tb, ts_
