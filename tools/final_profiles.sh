#!/bin/bash
# Regenerates the judged evidence of a round on the GPU box (run through gpurun): bench line, rocprofv3 kernel stats of
# the same command, size sweep, PMC traffic of k_accumulate.  Outputs under gpurun_out/final/ (copy into profiles/).
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"  # default: the checkout this script lives in
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
cp $O/stats/p_kernel_stats.csv $O/kernel_stats.csv
cd $R
python3 tools/sweep.py 10 12 14 16 17 18 19 20 21 22 23 24 > $O/sweep.txt 2>&1
bash tools/pmc_accumulate.sh > $O/pmc.log 2>&1
cp gpurun_out/pmc_acc/*.json $O/ 2>/dev/null
cp profiles/accumulate_pmc.json $O/accumulate_pmc_written.json 2>/dev/null
ls $O
