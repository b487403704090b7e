#!/bin/bash
# round 5: GLV split against the unsplit pipeline now that arkworks-form bases are gathered as they are (unsplit: no coordinate pass at all;
# split: the phi records) -- whole bench processes interleaved; then the sustained probe on the round-4 behaviour
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export GRAFT_REPO_ROOT=$PWD
O=gpurun_out/r5_fifth; mkdir -p $O
timeout 1500 python3 tools/ab_glv.py 17 18 19 20 21 > $O/glv_ab.txt 2>&1
cat $O/glv_ab.txt
echo "== round-4 behaviour (libmsm_hip_r4.so: k_convert_bases pass, 32-bit digit codes)" > $O/sustained_r4.txt
MSM_HIP_LIB=$PWD/tools/_ab/libmsm_hip_r4.so python3 tools/sustained_probe.py 2>&1 | grep -v amdgpu.ids | head -9 >> $O/sustained_r4.txt
echo "== default build, same box" >> $O/sustained_r4.txt
python3 tools/sustained_probe.py 2>&1 | grep -v amdgpu.ids | head -9 >> $O/sustained_r4.txt
cat $O/sustained_r4.txt
