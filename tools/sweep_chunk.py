#!/usr/bin/env python3
"""Interleaved A/B of MSM_HIP_CHUNK_LEN values in one session (3 rounds each) -- prints median bench value per L."""
import json, os, subprocess, sys, statistics
Ls = [int(x) for x in (sys.argv[1:] or "32 44 48 52 64".split())]
res = {L: [] for L in Ls}
for rnd in range(3):
    for L in Ls:
        env = dict(os.environ, MSM_HIP_CHUNK_LEN=str(L))
        p = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True, env=env)
        j = json.loads(p.stdout.strip().splitlines()[-1])
        res[L].append((j["value"], j["roofline"]["avg_kernel_ms"], j["stage_ms_untimed_diagnostic_step"]["reduce_ms"]))
for L in Ls:
    v = res[L]
    print("L", L, "median ms", statistics.median(x[0] for x in v), "acc", round(statistics.median(x[1] for x in v), 3), "reduce",
          round(statistics.median(x[2] for x in v), 3), [x[0] for x in v], flush=True)
