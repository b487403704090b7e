#!/bin/bash
# round 4: spread top window + cap at mean + 2 sigma: full GPU suite, then whole-process A/B against the build before both
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/spread; O=gpurun_out/spread
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_full.log 2>&1; echo "tests rc $?" >> $O/tests_full.log
tail -3 $O/tests_full.log
for L in 20 19 18 17 16 14 12 21; do
  timeout 900 python tools/ab_libs.py --rounds 3 base r4head -- --log-n $L > $O/ab2_$L.txt 2>&1
  cat $O/ab2_$L.txt
done
