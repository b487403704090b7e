#!/bin/bash
# round 4: the top window of UNSPLIT plans spread as well -- stage + parity tests, then window widths 17..20 at the sizes that run unsplit
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/wide
timeout 1500 python -m pytest tests/test_gpu_2_stages.py tests/test_gpu_1_parity.py -x -q -m gpu > gpurun_out/wide/tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed" gpurun_out/wide/tests.log | tail -1
CS=0,18,19,20 timeout 2400 python tools/sweep_c.py 21 22 23 24 > gpurun_out/wide/sweep.txt 2>&1; cat gpurun_out/wide/sweep.txt
timeout 600 python tools/ab_libs.py --rounds 2 base r4head -- --log-n 20 --no-glv > gpurun_out/wide/noglv20.txt 2>&1; cat gpurun_out/wide/noglv20.txt
