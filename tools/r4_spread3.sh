#!/bin/bash
# round 4: whole-bucket cap (MSM_HIP_PIECE_LEN = runs of L + rest) at the other sizes, whole-process A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/spread; O=gpurun_out/spread
run() { L=$1; shift; timeout 1200 python tools/ab_libs.py --rounds 2 "$@" -- --log-n $L > $O/len_$L.txt 2>&1; cat $O/len_$L.txt; }
run 18 base base,MSM_HIP_PIECE_LEN=20 base,MSM_HIP_PIECE_LEN=24 base,MSM_HIP_PIECE_LEN=28
run 21 base base,MSM_HIP_PIECE_LEN=36 base,MSM_HIP_PIECE_LEN=43 base,MSM_HIP_PIECE_LEN=48
run 22 base base,MSM_HIP_PIECE_LEN=72 base,MSM_HIP_PIECE_LEN=80 base,MSM_HIP_PIECE_LEN=96
run 24 base base,MSM_HIP_PIECE_LEN=288 base,MSM_HIP_PIECE_LEN=320 base,MSM_HIP_PIECE_LEN=384
run 16 base base,MSM_HIP_PIECE_LEN=8 base,MSM_HIP_PIECE_LEN=12
