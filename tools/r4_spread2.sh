#!/bin/bash
# round 4: with the top window spread, where should whole buckets end?  MSM_HIP_PIECE_LEN sweep (runs of L + rest), whole-process A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/spread; O=gpurun_out/spread
timeout 1500 python tools/ab_libs.py --rounds 2 base r4head base,MSM_HIP_PIECE_LEN=72 base,MSM_HIP_PIECE_LEN=80 base,MSM_HIP_PIECE_LEN=88 base,MSM_HIP_PIECE_LEN=96 base,MSM_HIP_PIECE_LEN=112 -- --log-n 20 > $O/len_20.txt 2>&1
cat $O/len_20.txt
timeout 900 python tools/ab_libs.py --rounds 2 base r4head base,MSM_HIP_PIECE_LEN=36 base,MSM_HIP_PIECE_LEN=40 base,MSM_HIP_PIECE_LEN=48 -- --log-n 19 > $O/len_19.txt 2>&1
cat $O/len_19.txt
timeout 900 python tools/ab_libs.py --rounds 2 base r4head base,MSM_HIP_PIECE_LEN=10 base,MSM_HIP_PIECE_LEN=12 base,MSM_HIP_PIECE_LEN=24 -- --log-n 17 > $O/len_17.txt 2>&1
cat $O/len_17.txt
