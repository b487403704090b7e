#!/usr/bin/env python3
"""Interleaved A/B of a knob read at context creation on the HOST-pointer call (pinned memory): two contexts side by side.
tools/host_env_ab.py NAME v1,v2 [log_n,...]   ('-' = unset)"""
import os, sys, time, statistics
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gpu-acceleration_amd")]
import numpy as np, torch
import mopro_msm_hip as mh
from mopro_msm_hip import testhooks as th
name, vals = sys.argv[1], sys.argv[2].split(",")
sizes = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "19,20,22").split(",")]
gen = th.HooksContext()
ctxs = {}
for v in vals:
    if v == "-": os.environ.pop(name, None)
    else: os.environ[name] = v
    ctxs[v] = mh.MsmContext()
os.environ.pop(name, None)
for lg in sizes:
    n = 1 << lg
    d_b = torch.empty(n * 16, dtype=torch.int32, device="cuda:0"); d_s = torch.empty(n * 8, dtype=torch.int32, device="cuda:0")
    gen.generate_device(21, 22, n, d_b.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
    hb = d_b.cpu().pin_memory().numpy().view(np.uint32).reshape(n, 16); hs = d_s.cpu().pin_memory().numpy().view(np.uint32).reshape(n, 8)
    pb, ps_ = d_b.cpu().numpy().view(np.uint32).reshape(n, 16), d_s.cpu().numpy().view(np.uint32).reshape(n, 8)
    for kind, (b, s) in (("pinned", (hb, hs)), ("pageable", (pb, ps_))):
        rows = {v: [] for v in vals}; ref = None
        for rnd in range(4):
            for v in vals:
                c = ctxs[v]
                for _ in range(3): r = c.msm(b, s, mh.FORM_MONT)
                ts = []
                for _ in range(12):
                    t = time.perf_counter(); r = c.msm(b, s, mh.FORM_MONT); ts.append((time.perf_counter() - t) * 1e3)
                a = r.affine_std.copy()
                if ref is None: ref = a
                rows[v].append((statistics.median(ts), bool((a == ref).all())))
        print(f"2^{lg} {kind:8s} " + "  ".join(f"{name}={v}: {statistics.median(y[0] for y in rows[v]):.4f} ms" for v in vals)
              + f"  same={all(y[1] for v in vals for y in rows[v])}", flush=True)
