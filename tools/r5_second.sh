#!/bin/bash
# round 5, second GPU call: parity suite (paired products / DPP exchanges in the latency-bound kernels), whole-process A/B
cd "$(dirname "${BASH_SOURCE[0]}")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5_second_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r5_second_pytest.txt
tail -5 gpurun_out/r5_second_pytest.txt
rm -f gpurun_out/r5_second_ab.txt
for lg in 20 17 16; do
  echo "== 2^$lg" >> gpurun_out/r5_second_ab.txt
  timeout 900 python tools/ab_libs.py --rounds 4 r4 noilp conv base -- --log-n $lg >> gpurun_out/r5_second_ab.txt 2>&1
done
echo "== 2^22" >> gpurun_out/r5_second_ab.txt
timeout 900 python tools/ab_libs.py --rounds 3 r4 conv base -- --log-n 22 >> gpurun_out/r5_second_ab.txt 2>&1
cat gpurun_out/r5_second_ab.txt
