#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_run2; mkdir -p $O
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "stream or arkworks or multi" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
python tools/host_path_sweep.py 18 19 20 22 > $O/host_sweep.txt 2>&1; cat $O/host_sweep.txt
python tools/chunk_len_sweep.py 16,17,18,19,20 0,8,10,11,12,14,16,20,22,24,28,29,30,32,40,48,58,64 > $O/chunk_len.txt 2>&1; cat $O/chunk_len.txt
cd /tmp && export TMPDIR=/tmp
for m in pageable pinned-copy pinned-pull; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$m -o t -- python3 $GRAFT_REPO_ROOT/tools/host_call_trace.py $m > $GRAFT_REPO_ROOT/$O/trace_$m.log 2>&1
  echo "== $m"; tail -1 $GRAFT_REPO_ROOT/$O/trace_$m.log; python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/$O/trace_$m | head -12
done
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_valu.sh > $O/pmc_valu.log 2>&1; tail -60 $O/pmc_valu.log
