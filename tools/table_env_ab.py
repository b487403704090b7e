#!/usr/bin/env python3
"""Interleaved A/B of a knob read at context creation on the RESIDENT path with its window table (single calls and batch).
tools/table_env_ab.py NAME v1,v2 [log_n,...]      (AB_TABLE=0 in the environment: the same resident calls without the table)"""
import os, sys, time, statistics
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
name, vals = sys.argv[1], sys.argv[2].split(",")
sizes = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "12,14,16,17,18,20").split(",")]  # log2(n), or n itself when > 64
gen = th.HooksContext()
FLAGS = mh.FLAG_WINDOW_TABLE if os.environ.get("AB_TABLE", "1") == "1" else 0
for lg in sizes:
    n = 1 << lg if lg <= 64 else lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(31, 32, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    hb = d_b.cpu().numpy().view(np.uint32).reshape(n, 16)
    hs = d_s.cpu().pin_memory().numpy().view(np.uint32).reshape(n, 8)
    ctxs = {}
    for v in vals:
        if v == "-": os.environ.pop(name, None)
        else: os.environ[name] = v
        ctxs[v] = mh.MsmContext(flags=FLAGS)
        ctxs[v].upload_bases(hb, mh.FORM_MONT)
    os.environ.pop(name, None)
    rows = {v: [] for v in vals}; brows = {v: [] for v in vals}; ref = None
    for rnd in range(4):
        for v in vals:
            c = ctxs[v]
            for _ in range(5): r = c.msm_resident(hs)
            ts = []
            for _ in range(20):
                t = time.perf_counter(); r = c.msm_resident(hs); ts.append((time.perf_counter() - t) * 1e3)
            a = r.affine_std.copy()
            if ref is None: ref = a
            rows[v].append((statistics.median(ts), bool((a == ref).all())))
            t = time.perf_counter(); rb = c.msm_resident_batch([hs] * 8, want_affine=False); brows[v].append((time.perf_counter() - t) * 1e3 / 8)
            if rnd == 0:
                rb = c.msm_resident_batch([hs] * 5, want_affine=True)
                rows[v].append((rows[v][-1][0], all(bool((x.affine_std == ref).all()) for x in rb)))
    print(f"{'2^%d' % lg if lg <= 64 else lg} {'table' if FLAGS else 'plain'}: " + "  ".join(f"{name}={v}: single {statistics.median(y[0] for y in rows[v]):.4f} batch {min(brows[v]):.4f} ms" for v in vals)
          + f"  same={all(y[1] for v in vals for y in rows[v])}", flush=True)
    for c in ctxs.values(): c.close()
