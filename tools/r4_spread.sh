#!/bin/bash
# round 4: top window of split plans spread over all of its buckets (msmplan::glv_top_digit_bits) -- parity, then whole-process A/B against
# the build without it (tools/_ab/libmsm_hip_r4head.so) at 2^20, 2^17, 2^14, 2^12
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/spread; O=gpurun_out/spread
timeout 1500 python -m pytest tests/test_gpu_2_stages.py tests/test_gpu_1_parity.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -3 $O/tests.log
for L in 20 17 14 12; do
  timeout 900 python tools/ab_libs.py --rounds 3 base r4head -- --log-n $L > $O/ab_$L.txt 2>&1
  cat $O/ab_$L.txt
done
