#!/usr/bin/env python3
"""Interleaved A/B of one environment knob: tools/sweep_env.py NAME v1 v2 ... [-- bench args]; median of 3 rounds per value."""
import json, os, subprocess, sys, statistics
import os
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--"); extra = args[k + 1:]; args = args[:k]
name, vals = args[0], args[1:]
res = {v: [] for v in vals}
for rnd in range(3):
    for v in vals:
        env = dict(os.environ, **{name: v})
        p = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "3", "--no-cpu-baseline"] + extra, capture_output=True, text=True, env=env)
        j = json.loads(p.stdout.strip().splitlines()[-1])
        res[v].append((j["value"], j["stage_ms_untimed_diagnostic_step"]["reduce_ms"], j["bit_exact"]))
for v in vals:
    r = res[v]
    print(name, v, "median ms", statistics.median(x[0] for x in r), "reduce", round(statistics.median(x[1] for x in r), 3), [x[0] for x in r], all(x[2] for x in r), flush=True)
