#!/bin/bash
# round 5: (1) what a deterministic PLACEMENT costs (every staged bucket's run sorted by point index, -DMSM_AB_SORTED_BUCKETS) against the default
# build, whole processes interleaved; (2) the sustained probe (200 back-to-back calls: Mcycles and sclk) on the default build and on the round-4
# behaviour (conversion pass, 32-bit digits)
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out/r5_fourth; mkdir -p $O
for lg in 20 17; do
  echo "== 2^$lg" >> $O/sorted_buckets_ab.txt
  timeout 900 python tools/ab_libs.py --rounds 3 base sortedb -- --log-n $lg >> $O/sorted_buckets_ab.txt 2>&1
done
MSM_HIP_LIB=$PWD/tools/_ab/libmsm_hip_sortedb.so python3 bench.py --steps 20 --no-cpu-baseline --no-host-legs 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sortedb: value', j['value'], 'bit_exact', j['bit_exact'], j['bit_exact_steps'], 'sort_ms', j['stage_ms_untimed_diagnostic_step']['sort_ms'])" >> $O/sorted_buckets_ab.txt
python3 bench.py --steps 20 --no-cpu-baseline --no-host-legs 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base: value', j['value'], 'bit_exact', j['bit_exact'], j['bit_exact_steps'], 'sort_ms', j['stage_ms_untimed_diagnostic_step']['sort_ms'])" >> $O/sorted_buckets_ab.txt
cat $O/sorted_buckets_ab.txt
echo "== default build" > $O/sustained.txt
python3 tools/sustained_probe.py >> $O/sustained.txt 2>&1
echo "== round-4 behaviour (libmsm_hip_r4.so: k_convert_bases pass, 32-bit digit codes)" >> $O/sustained.txt
MSM_HIP_LIB=$PWD/tools/_ab/libmsm_hip_r4.so python3 tools/sustained_probe.py >> $O/sustained.txt 2>&1
grep -v amdgpu.ids $O/sustained.txt
