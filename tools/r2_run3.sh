#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_run3; mkdir -p $O
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "stream or arkworks or multi or golden_default" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
python tools/host_path_sweep.py 18 19 20 22 > $O/host_sweep.txt 2>&1; grep -v "chunk 2^17\|tail" $O/host_sweep.txt
cd /tmp && export TMPDIR=/tmp
for m in pinned-copy pinned-pull; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$m -o t -- python3 $GRAFT_REPO_ROOT/tools/host_call_trace.py $m > $GRAFT_REPO_ROOT/$O/trace_$m.log 2>&1
  echo "== $m"; tail -1 $GRAFT_REPO_ROOT/$O/trace_$m.log
done
cd "$GRAFT_REPO_ROOT"
python tools/sweep.py 14 16 17 18 19 20 > $O/sweep.txt 2>&1; cat $O/sweep.txt
