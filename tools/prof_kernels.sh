#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof_kernels.sh <log_n> [more log_n...]   -> per-kernel averages (rocprofv3 --kernel-trace --stats)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"  # default: the checkout this script lives in
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof$n
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --log-n $n --no-cpu-baseline --no-host-legs > $out/bench.json 2>/dev/null
  echo "== N=2^$n: $(python3 -c "import json,sys; j=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(j['value'],'ms exact',j['bit_exact'])")"
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $out | grep -v "^#" | head -${TOPK:-14}
done
