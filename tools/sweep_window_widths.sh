#!/bin/bash
# round 4: window widths of split plans re-measured with the top window spread (narrow top windows are no longer degenerate)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/spread
CS=${CS:-0,10,11,12,13,14,15,16} timeout 2400 python tools/sweep_c.py ${SIZES:-12 13 14 15 16 17 18 19} > gpurun_out/spread/sweep_c${TAG}.txt 2>&1
cat gpurun_out/spread/sweep_c${TAG}.txt
