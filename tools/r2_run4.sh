#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_run4; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
python tools/split_sweep.py > $O/split.txt 2>&1; cat $O/split.txt
python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python - <<'PY'
import json
j=json.loads(open("gpurun_out/r2_run4/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["bit_exact"], j["roofline"], j.get("host_pointer_legs"))
PY
