#!/usr/bin/env python3
"""Sizes that are not powers of two: msm_bn254_g1_device under two or more values of a context-creation knob, interleaved contexts, median of
15 calls x 4 rounds, k_accumulate time of one call, results compared.
usage: tools/odd_size_ab.py [--env MSM_HIP_CHUNK_ROUNDS] [--vals 0,1] [--wb 0] [--flags 0] [n ...]
  default: the chunk length of k_accumulate fitted to whole rounds of workgroups (1, default) against the power-of-two length (0);
  --env MSM_HIP_GLV_MAX_LOG2 --vals 20,21: where the GLV split stops.  A value `-` leaves the variable unset."""
import argparse, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
ap = argparse.ArgumentParser()
ap.add_argument("--env", default="MSM_HIP_CHUNK_ROUNDS")
ap.add_argument("--vals", default="0,1")
ap.add_argument("--wb", type=int, default=0)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("sizes", nargs="*", type=int)
a = ap.parse_args()
sizes = a.sizes or [(1 << 19) + 12345, 600000, 750000, 1000000, (1 << 20) + 1, 1200000, 1500000, 1800000, 3000000]
vals = a.vals.split(",")
gen = th.HooksContext()
for n in sizes:
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(41, 42, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    ctxs, plans = {}, {}
    for v in vals:
        if v == "-": os.environ.pop(a.env, None)
        else: os.environ[a.env] = v
        ctxs[v] = mh.MsmContext(window_bits=a.wb, flags=a.flags)
        plans[v] = mh.plan(n, a.wb, a.flags)
    os.environ.pop(a.env, None)
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
    print(f"n {n:8d}: " + "  ".join(f"{a.env}={v} (c {plans[v].window_bits} W {plans[v].num_windows} glv {plans[v].glv}) {statistics.median(med[v]):.4f} ms, "
                                    f"k_accumulate {acc[v]:.3f}" for v in vals) + f"  same={same}", flush=True)
    for c in ctxs.values(): c.close()
