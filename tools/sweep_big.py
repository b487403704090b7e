#!/usr/bin/env python3
"""Window size at large N: bench ms for c = 16, 17, 18 at 2^21..2^24 (is the 32768-bucket cap still the right plan?)."""
import json, subprocess, sys
for n in [int(x) for x in (sys.argv[1:] or "21 22 23 24".split())]:
    row = []
    for c in (16, 17, 18):
        p = subprocess.run([sys.executable, "bench.py", "--log-n", str(n), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--window-bits", str(c)],
                           capture_output=True, text=True)
        try:
            j = json.loads(p.stdout.strip().splitlines()[-1]); st = j["stage_ms_untimed_diagnostic_step"]
            row.append("c%d=%.3f%s (sort %.2f acc %.2f red %.2f)" % (c, j["value"], "" if j["bit_exact"] else "(WRONG)", st["sort_ms"], st["accumulate_ms"], st["reduce_ms"]))
        except Exception as e:
            row.append("c%d=FAIL %s" % (c, p.stderr[-200:]))
    print("logN", n, " | ".join(row), flush=True)
