#!/bin/bash
# round 5: full GPU suite on the build with the hooks-only knobs / new tests, kernel timelines, a default bench line
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export GRAFT_REPO_ROOT=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_third_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r5_third_pytest.txt
tail -8 gpurun_out/r5_third_pytest.txt
python3 bench.py > gpurun_out/r5_third_bench.json 2> gpurun_out/r5_third_bench.err; echo "bench rc $?"
bash tools/trace_device_call.sh 20 17 > gpurun_out/r5_third_timeline.txt 2>&1
cat gpurun_out/r5_third_timeline.txt | head -80
