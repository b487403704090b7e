#!/bin/bash
# round 5, first GPU call: parity suite on the new digit codes / direct gathers, then whole-process A/B against the round-4 behaviour
cd "$(dirname "${BASH_SOURCE[0]}")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5_first_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r5_first_pytest.txt
tail -5 gpurun_out/r5_first_pytest.txt
for lg in 20 17 19; do
  echo "== 2^$lg" >> gpurun_out/r5_first_ab.txt
  timeout 900 python tools/ab_libs.py --rounds 4 r4 conv d32 base -- --log-n $lg >> gpurun_out/r5_first_ab.txt 2>&1
done
cat gpurun_out/r5_first_ab.txt
