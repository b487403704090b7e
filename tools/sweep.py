#!/usr/bin/env python3
"""Run bench.py over a range of instance sizes on one GPU and print one summary line per size."""
import json, subprocess, sys
sizes = [int(x) for x in (sys.argv[1:] or "10 12 14 16 18 20 22 23 24".split())]
for n in sizes:
    p = subprocess.run([sys.executable, "bench.py", "--log-n", str(n), "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-host-legs"],
                       capture_output=True, text=True)
    try:
        j = json.loads(p.stdout.strip().splitlines()[-1])
        st = {k: round(v, 3) for k, v in j["stage_ms_untimed_diagnostic_step"].items()}
        print("logN", n, "ms", j["value"], "exact", j["bit_exact"], "c", j["config"]["window_bits"], "acc_ms",
              j["roofline"]["avg_kernel_ms"], "frac", j["roofline"]["frac"], st, flush=True)
    except Exception as e:
        print("logN", n, "FAILED", e, p.stderr[-600:], flush=True)
