#!/usr/bin/env python3
"""Window-size sweep per instance size (planner tuning): prints bench ms per (logN, c)."""
import json, os, subprocess, sys
sizes = [int(x) for x in (sys.argv[1:] or "10 12 14 15 16".split())]
for n in sizes:
    row = []
    for c in [int(x) for x in (os.environ.get("CS") or "0,7,8,9,10,11,12,13,14,15,16").split(",")]:
        if c and (1 << (c - 1)) > (1 << n) * 4:
            continue
        p = subprocess.run([sys.executable, "bench.py", "--log-n", str(n), "--steps", "60", "--warmup", "5", "--no-cpu-baseline", "--no-host-legs",
                            "--window-bits", str(c)], capture_output=True, text=True)
        try:
            j = json.loads(p.stdout.strip().splitlines()[-1])
            row.append("c%d%s=%.3f%s" % (j["config"]["window_bits"], "*" if c == 0 else "", j["value"], "" if j["bit_exact"] else "(WRONG)"))
        except Exception as e:
            row.append("c%d=FAIL" % c)
    print("logN", n, " ".join(row), flush=True)
