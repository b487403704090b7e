#!/bin/bash
# usage: tools/prof_c.sh <log_n> <c> [<c> ...]  -> k_accumulate / total per window size (rocprofv3 kernel stats)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"  # default: the checkout this script lives in
cd /tmp && export TMPDIR=/tmp
n=$1; shift
for c in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/profc_${n}_$c
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --log-n $n --steps 5 --warmup 1 --no-cpu-baseline --window-bits $c > $out/bench.json 2>/dev/null
  echo "== N=2^$n c=$c: $(python3 -c "import json,sys; j=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(j['value'],'ms W',j['config']['num_windows'],'nb',j['config']['buckets_per_window'])")"
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $out | grep -v "^#" | grep -v gen_ | head -${TOPK:-6}
done
