#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/host_timeline; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- python3 $R/tools/host_call_timeline.py ${1:-20} > $out/run.txt 2>/dev/null
tail -1 $out/run.txt
python3 $R/tools/host_call_timeline.py --summarise $out
head -2 $(find $out -name "*memory_copy_trace.csv" | head -1)
