#!/usr/bin/env python3
"""Window-size x chunk-length grid at the per-GPU sizes of the multi-GPU headline runs (2^17..2^19)."""
import json, os, subprocess, sys
sizes = [int(x) for x in (sys.argv[1:] or "17 18 19".split())]
for n in sizes:
    for c in (0, 12, 13, 14, 15, 16):
        for L in (16, 32):
            env = dict(os.environ, MSM_HIP_CHUNK_LEN=str(L))
            p = subprocess.run([sys.executable, "bench.py", "--log-n", str(n), "--steps", "20", "--warmup", "3", "--no-cpu-baseline",
                                "--window-bits", str(c)], capture_output=True, text=True, env=env)
            try:
                j = json.loads(p.stdout.strip().splitlines()[-1])
                st = j["stage_ms_untimed_diagnostic_step"]
                print("logN", n, "c", c, "->", j["config"]["window_bits"], "L", L, "ms", j["value"], "exact", j["bit_exact"],
                      "sort", round(st["sort_ms"], 3), "acc", round(st["accumulate_ms"], 3), "reduce", round(st["reduce_ms"], 3),
                      "finish", round(st["finish_ms"], 3), flush=True)
            except Exception as e:
                print("logN", n, "c", c, "L", L, "FAILED", e, p.stderr[-300:], flush=True)
