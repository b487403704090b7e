#!/usr/bin/env python3
"""Summarise tools/pmc_accumulate.sh output: per-dispatch FETCH_SIZE / WRITE_SIZE (KiB) of k_accumulate, the
calibration factors measured with tools/calib_gather, and the corrected HBM bytes per launch.
Writes <dir>/summary.json; copy it to profiles/accumulate_pmc_2p<log_n>.json to have bench.py report `roofline.traffic`."""
import csv, glob, json, sys, collections
d = sys.argv[1]
def per_kernel(pattern):
    out = collections.defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]].append(float(r["Counter_Value"]))  # (runs with --no-host-legs: only the whole-MSM launch k_accumulate<false, false> exists)
    return out
res = {}
known = {"k_stream": 1 << 30, "k_gather": (1 << 22) * 64, "k_store36": (1 << 21) * 144}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    b = per_kernel(f"{d}/bench_{c}/**/*counter_collection.csv")
    k = per_kernel(f"{d}/calib_{c}/**/*counter_collection.csv")
    res[c] = {"k_accumulate_kib_per_launch": sum(b.get("k_accumulate_pieces", [0])) / max(1, len(b.get("k_accumulate_pieces", [1]))),
              "launches": len(b.get("k_accumulate_pieces", [])),
              "calib_counter_kib": {n: sum(v) / len(v) for n, v in k.items()},
              "calib_known_bytes": known}
f, w = res["FETCH_SIZE"], res["WRITE_SIZE"]
# correction factors = known bytes / counted bytes for the matching access pattern
fg = known["k_gather"] / (f["calib_counter_kib"].get("k_gather", 0) * 1024 or 1)
fs = known["k_stream"] / (f["calib_counter_kib"].get("k_stream", 0) * 1024 or 1)
ws = known["k_store36"] / (w["calib_counter_kib"].get("k_store36", 0) * 1024 or 1)
res["correction"] = {"fetch_gather_64B_records": fg, "fetch_stream_16B_per_lane": fs, "write_144B_records": ws}
res["hbm_bytes_per_launch"] = int(f["k_accumulate_kib_per_launch"] * 1024 * fg + w["k_accumulate_kib_per_launch"] * 1024 * ws)
try:  # the bench line printed under the profiler names the workload the counters belong to
    line = [l for l in open(f"{d}/bench_FETCH_SIZE.log") if l.startswith("{")][-1]
    res["source_hash"] = json.loads(line).get("source_hash")  # of the build the counters belong to: bench.py flags a later build's line traffic_stale
    cfg = json.loads(line)["config"]
    res["n_local"], res["window_bits"] = cfg["n_per_gpu"], cfg["window_bits"]
    res["num_windows"], res["glv_split"] = cfg.get("num_windows"), cfg.get("glv_split")
    res["algorithmic_bytes_per_launch"] = json.loads(line)["roofline"]["algorithmic_bytes_per_launch"]
    res["traffic_over_algorithmic"] = round(res["hbm_bytes_per_launch"] / res["algorithmic_bytes_per_launch"], 3)
except Exception as e:
    res["n_local"], res["window_bits"] = None, None
res["build"] = sys.argv[2] if len(sys.argv) > 2 else "?"
res["how"] = ("tools/pmc_accumulate.sh on an MI355X gpurun box: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
              "(with --kernel-trace only) around `python3 bench.py --steps 3 --warmup 1`; counters corrected with the factors measured "
              "by tools/calib_gather on the same box (64-B record gathers count exactly, 16 B/lane streams count 1/2, 144-B record "
              "stores over-count by ~5.6 %)")
json.dump(res, open(f"{d}/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
