#!/bin/bash
# builds the PRODUCT library with extra compiler flags as tools/_ab/libmsm_hip_<name>.so (git-ignored, travels with gpurun): A/B of kernel
# variants by MSM_HIP_LIB on one box (tools/ab_libs.py).   usage: tools/build_variant.sh name "-DFLAG ..."
cd "$(dirname "${BASH_SOURCE[0]}")/.."
mkdir -p tools/_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $2 -shared -o tools/_ab/libmsm_hip_$1.so gpu-acceleration_amd/csrc/msm_hip.hip -Wl,-Bsymbolic -ldl ${3:+-Rpass-analysis=kernel-resource-usage} 2>&1 | grep -A12 "k_accumulate_piecesILb0ELb0E" | grep -E "VGPRs:|Occupancy|Scratch" 
echo "built tools/_ab/libmsm_hip_$1.so ($2)"
