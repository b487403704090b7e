#!/usr/bin/env python3
"""Sizes that are not powers of two: msm_bn254_g1_device with the chunk length of k_accumulate fitted to whole rounds of workgroups
(default) against the power-of-two length (MSM_HIP_CHUNK_ROUNDS=0).  Interleaved contexts, median of 15 calls, 4 rounds; stage times of one
call.   usage: tools/odd_size_ab.py [n ...]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
sizes = [int(x) for x in sys.argv[1:]] or [(1 << 19) + 12345, 600000, 750000, 1000000, (1 << 20) + 1, 1200000, 1500000, 1800000, 3000000]
gen = th.HooksContext()
for n in sizes:
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(41, 42, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    ctxs = {}
    for v in ("0", "1"):
        os.environ["MSM_HIP_CHUNK_ROUNDS"] = v
        ctxs[v] = mh.MsmContext()
    os.environ.pop("MSM_HIP_CHUNK_ROUNDS")
    med = {v: [] for v in ctxs}; acc = {}; ref = None; same = True
    for rnd in range(4):
        for v, c in ctxs.items():
            for _ in range(4): r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
            ts = []
            for _ in range(15):
                t = time.perf_counter(); r = c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t) * 1e3)
            med[v].append(statistics.median(ts))
            if ref is None: ref = r.affine_std.copy()
            same = same and bool((r.affine_std == ref).all())
    for v, c in ctxs.items():
        c.set_stage_timing(True); c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); acc[v] = c.timings()["accumulate_ms"]; c.set_stage_timing(False)
    pl = mh.plan(n)
    print(f"n {n:8d} c {pl.window_bits} W {pl.num_windows} glv {pl.glv}: power of two {statistics.median(med['0']):.4f} ms (k_accumulate {acc['0']:.3f})  "
          f"fitted {statistics.median(med['1']):.4f} ms (k_accumulate {acc['1']:.3f})  same={same}", flush=True)
    for c in ctxs.values(): c.close()
