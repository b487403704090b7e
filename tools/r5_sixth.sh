#!/bin/bash
# round 5: phi records written by the decomposition (default) against the separate k_phi_records launch (-DMSM_AB_PHI_KERNEL), whole processes
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out/r5_sixth; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_1_parity.py tests/test_gpu_2_stages.py -m gpu -x -q 2>&1 | tail -3
for lg in 20 19 18 17 16; do
  echo "== 2^$lg" >> $O/phi_fused_ab.txt
  timeout 900 python tools/ab_libs.py --rounds 4 phik base -- --log-n $lg >> $O/phi_fused_ab.txt 2>&1
done
cat $O/phi_fused_ab.txt
bash tools/trace_device_call.sh 20 17 > $O/timeline.txt 2>&1
grep -E "==|k_decompose|k_phi|k_coarse_hist|span" $O/timeline.txt
