#!/usr/bin/env python3
"""Interleaved A/B of product-library builds on ONE box: tools/ab_libs.py [--rounds R] name1 name2 ... [-- bench args]
(name = tools/_ab/libmsm_hip_<name>.so, `base` = the library as built in-tree).  Every round runs bench.py once per library, whole processes;
prints per library the medians of ms per step, k_accumulate ms, its Mcycles and the sustained clock."""
import json, os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--"); extra = args[k + 1:]; args = args[:k]
rounds = 3
if args and args[0] == "--rounds":
    rounds = int(args[1]); args = args[2:]
names = args
res = {v: [] for v in names}
for rnd in range(rounds):
    for v in (names if rnd % 2 == 0 else names[::-1]):
        env = dict(os.environ)
        env.pop("MSM_HIP_LIB", None)
        for part in v.split(","):  # "base", a library name, or NAME=value environment settings, comma-separated (e.g. pf2,MSM_HIP_CHUNK_LEN=48)
            if "=" in part:
                k_, v_ = part.split("=", 1)
                env[k_] = v_
            elif part != "base":
                env["MSM_HIP_LIB"] = os.path.join(ROOT, "tools", "_ab", "libmsm_hip_%s.so" % part)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-host-legs"] + extra,
                           capture_output=True, text=True, env=env)
        try:
            j = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception:
            print(v, "FAILED", p.stdout[-500:], p.stderr[-1500:], flush=True)
            continue
        st = j["stage_ms_untimed_diagnostic_step"]
        res[v].append((j["value"], j["roofline"]["avg_kernel_ms"], (j.get("clock") or {}).get("k_accumulate_mcycles"), (j.get("clock") or {}).get("sclk_ghz_timed_loop"),
                       st.get("sort_ms"), st.get("reduce_ms"), j["bit_exact"]))
med = lambda xs: statistics.median([x for x in xs if x is not None]) if any(x is not None for x in xs) else None
for v in names:
    r = res[v]
    if not r:
        continue
    print("%-28s ms/step %.4f  k_accumulate %.4f ms  %.4f Mcycles  sclk %.3f GHz  sort %.3f reduce %.3f | steps %s exact %s" % (
        v, med([x[0] for x in r]), med([x[1] for x in r]), med([x[2] for x in r]) or 0, med([x[3] for x in r]) or 0, med([x[4] for x in r]) or 0,
        med([x[5] for x in r]) or 0, [x[0] for x in r], all(x[6] for x in r)), flush=True)
