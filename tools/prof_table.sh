#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof_table.sh <log_n> <config> [more configs]  -> per-kernel averages of tools/table_sweep.py
# (rocprofv3 --kernel-trace --stats) for the resident path with / without the window table
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
n=$1; shift
for cfg in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/proftab_${n}_$cfg
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/table_sweep.py --logn $n --configs $cfg --batch 8 --reps 3 > $out/sweep.log 2>/dev/null
  echo "== N=2^$n $cfg: $(tail -1 $out/sweep.log | cut -c1-230)"
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $out | grep -v "^#" | head -${TOPK:-18}
done
