#!/bin/bash
# round 4: host-pointer call with the copy stream carrying copies only: timeline, then the sweep of chunk sizes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/host2
bash tools/r4_host_timeline.sh 20 > gpurun_out/host2/timeline.txt 2>&1; head -1 gpurun_out/host2/timeline.txt; grep -c COPY gpurun_out/host2/timeline.txt
timeout 1200 python3 tools/host_path_sweep.py 19 20 22 > gpurun_out/host2/sweep.txt 2>&1; cat gpurun_out/host2/sweep.txt
for k in 16 17; do MSM_HIP_STREAM_CHUNK_LOG2=$k timeout 300 python3 tools/host_call_timeline.py 20 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_3_configs.py tests/test_gpu_1_parity.py -x -q -m gpu -k "stream or host or ark or chunk" 2>&1 | tail -2
