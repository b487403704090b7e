#!/usr/bin/env python3
"""HBM traffic of the decomposition / sort / plan kernels from the counter passes tools/pmc_accumulate.sh already makes (FETCH_SIZE and
WRITE_SIZE per dispatch): usage tools/pmc_sort_summarize.py gpurun_out/pmc_acc_20 [build]  ->  <dir>/sort_summary.json
(copy to profiles/sort_pmc_2p<log_n>.json; bench.py reports it as roofline_sort.traffic for the matching shape).
Only the NEWEST counter file of each pass is read (gpurun_out accumulates earlier runs).  Fetches of the streaming kernels are corrected with
the factor tools/calib_gather measures for 16-byte-per-lane streams on the same box (the counter sees half of them); writes count as they are."""
import csv, glob, json, os, sys, collections
d = sys.argv[1]
def newest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1:] if fs else []
def per_kernel(files):
    out = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]].append(float(r["Counter_Value"]))
    return out
fetch = per_kernel(newest(f"{d}/bench_FETCH_SIZE/**/*counter_collection.csv"))
write = per_kernel(newest(f"{d}/bench_WRITE_SIZE/**/*counter_collection.csv"))
cal = per_kernel(newest(f"{d}/calib_FETCH_SIZE/**/*counter_collection.csv"))
stream_factor = (1 << 30) / (sum(cal["k_stream"]) / len(cal["k_stream"]) * 1024) if cal.get("k_stream") else 2.0
SORT = ["k_coarse_hist", "k_coarse_prefix", "k_coarse_starts", "k_coarse_scatter", "k_fine_sort"]  # (round 6: k_big_place became part of k_place_count, the plan stage)
OTHER = ["k_decompose_glv", "k_decompose", "k_convert_bases", "k_phi_records", "k_place_count", "k_piece_count", "k_piece_scatter", "k_combine_pieces", "k_pair_level8", "k_pair_tail", "k_reduce_bits_wide"]
avg = lambda v: sum(v) / len(v) if v else 0.0
rows = {}
for k in SORT + OTHER:
    if k in fetch or k in write:
        rows[k] = {"fetch_bytes": int(avg(fetch.get(k, [])) * 1024 * stream_factor), "write_bytes": int(avg(write.get(k, [])) * 1024), "dispatches": len(fetch.get(k, []))}
res = {"kernels": rows, "stream_fetch_correction": round(stream_factor, 3),
       "sort_hbm_bytes": sum(rows[k]["fetch_bytes"] + rows[k]["write_bytes"] for k in SORT if k in rows),
       "build": sys.argv[2] if len(sys.argv) > 2 else "?",
       "how": "tools/pmc_accumulate.sh passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, --kernel-trace only, python3 bench.py --steps 3 --warmup 1 directly after `--`), "
              "newest counter file of each pass; per-dispatch averages; fetches x the 16-B-per-lane stream factor of tools/calib_gather"}
try:
    line = [l for l in open(f"{d}/bench_FETCH_SIZE.log") if l.startswith("{")][-1]
    j = json.loads(line)
    res["source_hash"] = j.get("source_hash")  # of the build the counters belong to (bench.py: traffic_stale)
    res["n_local"], res["window_bits"], res["glv_split"] = j["config"]["n_per_gpu"], j["config"]["window_bits"], j["config"].get("glv_split")
    res["algorithmic_bytes"] = j["roofline_sort"]["algorithmic_bytes"]
    res["traffic_over_algorithmic"] = round(res["sort_hbm_bytes"] / res["algorithmic_bytes"], 3)
except Exception:
    pass
json.dump(res, open(f"{d}/sort_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
