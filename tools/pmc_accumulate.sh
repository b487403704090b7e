#!/bin/bash
# Collect HBM traffic counters for k_accumulate (separate --pmc passes, kernel-trace only: see the guide's
# rocprofv3 rules) and the counter calibration of tools/calib_gather.   usage: tools/pmc_accumulate.sh [log_n ...]  (default 20)
# Output under gpurun_out/pmc_acc_<log_n>/; summary.json -> profiles/accumulate_pmc_2p<log_n>.json (bench.py reports it as
# roofline.traffic for the matching shape, labelled "source": "file").
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"  # default: the checkout this script lives in
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
[ -x tools/calib_gather ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/calib_gather tools/calib_gather.hip
for n in ${*:-20}; do
  O=gpurun_out/pmc_acc_$n
  mkdir -p $O
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/bench_$c -- python3 bench.py --log-n $n --steps 3 --warmup 1 --pre-warm-ms 0 --no-cpu-baseline --no-host-legs > $O/bench_$c.log 2>&1
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -- ./tools/calib_gather > $O/calib_$c.log 2>&1
  done
  python3 tools/pmc_summarize.py $O "${BUILD:-round 5}" > $O/summary.txt
  cp $O/summary.json gpurun_out/accumulate_pmc_2p$n.json
  python3 tools/pmc_sort_summarize.py $O "${BUILD:-round 5}" > $O/sort_summary.txt && cp $O/sort_summary.json gpurun_out/sort_pmc_2p$n.json
  python3 -c "import json; j=json.load(open('$O/summary.json')); print('2^$n: c', j['window_bits'], 'W', j.get('num_windows'), 'glv', j.get('glv_split'), 'HBM bytes per launch', j['hbm_bytes_per_launch'], 'algorithmic', j.get('algorithmic_bytes_per_launch'), 'ratio', j.get('traffic_over_algorithmic'))"
done
