#!/bin/bash
# Collect HBM traffic counters for k_accumulate (separate --pmc passes, kernel-trace only: see the guide's
# rocprofv3 rules) and the counter calibration of tools/calib_gather.  Output under gpurun_out/pmc_acc/.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"  # default: the checkout this script lives in
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_acc
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/bench_$c -- python3 bench.py --steps 3 --warmup 1 --pre-warm-ms 0 --no-cpu-baseline --no-host-legs > $O/bench_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -- ./tools/calib_gather > $O/calib_$c.log 2>&1
done
python3 tools/pmc_summarize.py $O
