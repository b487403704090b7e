#!/bin/bash
# per-kernel timeline of one device call at the given sizes (WB=<window bits> forces the width) (rocprofv3 --kernel-trace; summary by tools/device_call_trace.py)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  out=$R/gpurun_out/trace$n; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $R/tools/device_call_trace.py $n $WB > $out/run.txt 2>/dev/null
  echo "== 2^$n: $(cat $out/run.txt | tail -1)"
  python3 $R/tools/device_call_trace.py --summarise $out
done
