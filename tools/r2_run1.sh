#!/bin/bash
# round 2, first GPU call: full -m gpu suite, default bench line, host-path sweep
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_run1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
cat $O/bench.json
python tools/host_path_sweep.py 16 18 19 20 22 > $O/host_sweep.txt 2>&1
cat $O/host_sweep.txt
