#!/usr/bin/env python3
"""Summarise tools/pmc_valu.sh: per-launch SQ counters of k_accumulate and the VALU-busy fractions derived from them.
Writes <dir>/summary.json (copy to profiles/accumulate_valu_pmc.json: bench.py then reports it inside roofline_valu)."""
import csv, glob, json, sys, collections
d = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k in ("k_accumulate_pieces", "k_pair_level8", "k_combine_pieces", "k_fine_sort", "k_coarse_scatter"):
    if k not in vals: continue
    c = {n: sum(v) / len(v) for n, v in vals[k].items()}
    us = sum(dur[k]) / max(1, len(dur[k]))
    e = {"counters_per_launch": c, "avg_us_profiled": round(us, 1), "launches": len(dur[k])}
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        e["valu_issue_share_of_wave_lifetime"] = round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4)
    if "SQ_ACTIVE_INST_ANY" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        e["any_issue_share_of_wave_lifetime"] = round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
        e["wait_inst_any_share"] = round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 4)
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        e["valu_busy_frac"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), 4)
        e["effective_clock_ghz"] = round(cyc / (us * 1e3), 3) if us else None
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        e["valu_util_frac"] = round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64), 4)
    out[k] = e
res = {"kernels": out, "source": "tools/pmc_valu.sh: rocprofv3 --pmc (SQ_* / GRBM_GUI_ACTIVE), --kernel-trace only, python3 bench.py --steps 3 --warmup 1 directly after `--`; "
                                   "SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md)"}
a = out.get("k_accumulate_pieces", {})
for k in ("valu_busy_frac", "valu_util_frac"):
    if k in a: res[k] = a[k]
res["build"] = sys.argv[2] if len(sys.argv) > 2 else "?"
try:  # vector instructions per mixed addition and lane: wave instructions x 64 lanes / additions of the launch (from the bench line)
    line = [l for l in open(f"{d}/sq.log") if l.startswith("{")][-1]
    adds = json.loads(line)["roofline_valu"]["mixed_additions_per_launch"]
    res["mixed_additions_per_launch"] = adds
    res["valu_insts_per_mixed_addition"] = round(a["counters_per_launch"]["SQ_INSTS_VALU"] * 64 / adds, 1)
    res["effective_clock_ghz"] = a.get("effective_clock_ghz")
except Exception:
    pass
json.dump(res, open(f"{d}/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
