#!/usr/bin/env python3
"""Interleaved A/B of the GLV split against the unsplit pipeline: median bench ms per size (3 rounds each)."""
import json, subprocess, sys, statistics, os
import os
os.environ.setdefault("MSM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-acceleration_amd", "libmsm_hip_hooks.so"))  # the A/B knobs this script sets are read by the HOOKS build only (round 5)
for n in [int(x) for x in (sys.argv[1:] or "16 17 18 19 20 21 22".split())]:
    res = {"glv": [], "plain": []}
    for rnd in range(3):
        for name, extra in (("glv", []), ("plain", ["--no-glv"])):
            p = subprocess.run([sys.executable, "bench.py", "--log-n", str(n), "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-host-legs"] + extra,
                               env=dict(os.environ, MSM_HIP_GLV_MAX_LOG2="23"),
                               capture_output=True, text=True)
            j = json.loads(p.stdout.strip().splitlines()[-1])
            assert j["bit_exact"]
            res[name].append(j["value"])
    g, q = statistics.median(res["glv"]), statistics.median(res["plain"])
    print(f"logN {n}: GLV {g:.4f} ms  plain {q:.4f} ms  ({100 * (g / q - 1):+.1f} %)  {res}", flush=True)
