#!/usr/bin/env python3
"""msm_bn254_g1_device over sizes BETWEEN the powers of two (eighth-octave steps): looks for steps in the time per point that a planner
threshold, a chunk-length rule or a sort-geometry switch leaves behind.   usage: tools/size_sweep.py [lo_log2 hi_log2 steps_per_octave]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 14
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 22
spo = int(sys.argv[3]) if len(sys.argv) > 3 else 8
nmax = 1 << hi
gen = th.HooksContext()
d_b = torch.empty(nmax * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(nmax * 8, dtype=torch.int32, device="cuda:0")
gen.generate_device(51, 52, nmax, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
sizes = sorted({int(round(2 ** (lo + k / spo))) + (1 if k % spo == 0 and k else 0) for k in range((hi - lo) * spo + 1)} | {1 << hi})
prev = None
with mh.MsmContext() as c:
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.3: c.msm_device(d_b.data_ptr(), d_s.data_ptr(), 1 << 18)
    for n in sizes:
        n = min(n, nmax)
        for _ in range(3): c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n)
        ts = []
        for _ in range(9):
            t = time.perf_counter(); c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t) * 1e3)
        ms = statistics.median(ts)
        c.set_stage_timing(True); c.msm_device(d_b.data_ptr(), d_s.data_ptr(), n); tm = c.timings(); c.set_stage_timing(False)
        pl = mh.plan(n)
        step = "" if prev is None else "  %+5.1f %% time for %+5.1f %% points" % ((ms / prev[1] - 1) * 100, (n / prev[0] - 1) * 100)
        print("n %8d c %2d W %2d glv %d  %.4f ms  %.3f ns/point  sort %.3f acc %.3f reduce %.3f%s" %
              (n, pl.window_bits, pl.num_windows, pl.glv, ms, ms * 1e6 / n, tm["sort_ms"], tm["accumulate_ms"], tm["reduce_ms"], step), flush=True)
        prev = (n, ms)
