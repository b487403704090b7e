#!/bin/bash
# A/B on ONE box: the library as built (partial sums of fp_mul pinned in source order) vs rebuilt with -DFP_NO_MAD_PIN.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'logN', '$2', 'ms', d['value'], 'acc', d['roofline']['avg_kernel_ms'], 'fp_mul peak', d['roofline_valu']['peak'], 'exact', d['bit_exact'])"; }
run() { for n in 16 17 18 20 20 22; do python3 bench.py --log-n $n --no-cpu-baseline --no-host-legs 2>/dev/null | line "$1" $n; done; }
run "pin   "
cp gpu-acceleration_amd/libmsm_hip.so /tmp/pin.so; cp gpu-acceleration_amd/libmsm_hip_hooks.so /tmp/pin_hooks.so
make -C gpu-acceleration_amd/csrc clean >/dev/null; make -j2 -C gpu-acceleration_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFP_NO_MAD_PIN" >/dev/null 2>&1
run "no pin"
cp /tmp/pin.so gpu-acceleration_amd/libmsm_hip.so; cp /tmp/pin_hooks.so gpu-acceleration_amd/libmsm_hip_hooks.so
run "pin   "
