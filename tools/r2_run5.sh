#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_run5; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; grep -E "passed|failed|rc " $O/pytest.log | tail -3
python tools/sweep.py 10 14 16 17 18 19 20 21 > $O/sweep.txt 2>&1; cat $O/sweep.txt
tools/mfma_probe > $O/mfma_probe.txt 2>&1; cat $O/mfma_probe.txt
tools/f4_batch_affine > $O/f4_batch_affine.txt 2>&1; cat $O/f4_batch_affine.txt
python tools/f4_tables.py 20 22 > $O/f4_tables.txt 2>&1; cat $O/f4_tables.txt
