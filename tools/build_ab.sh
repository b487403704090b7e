#!/bin/bash
# A/B on ONE box: the library as built vs rebuilt there with extra compiler flags.   usage: build_ab.sh "-DFLAG ..." [log_n ...]
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT"
FLAGS="$1"; shift
SIZES="${*:-17 20 20 22}"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'logN', '$2', 'ms', d['value'], 'acc', d['roofline']['avg_kernel_ms'], 'fp_mul peak', d['roofline_valu']['peak'], 'finish', d['stage_ms_untimed_diagnostic_step'].get('finish_ms'), 'reduce', d['stage_ms_untimed_diagnostic_step'].get('reduce_ms'), 'exact', d['bit_exact'])"; }
run() { for n in $SIZES; do python3 bench.py --log-n $n --no-cpu-baseline --no-host-legs 2>/dev/null | line "$1" $n; done; }
run "as built      "
cp gpu-acceleration_amd/libmsm_hip.so /tmp/a.so; cp gpu-acceleration_amd/libmsm_hip_hooks.so /tmp/a_hooks.so
make -C gpu-acceleration_amd/csrc clean >/dev/null; make -j2 -C gpu-acceleration_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS" >/dev/null 2>&1
run "with $FLAGS"
cp /tmp/a.so gpu-acceleration_amd/libmsm_hip.so; cp /tmp/a_hooks.so gpu-acceleration_amd/libmsm_hip_hooks.so
run "as built      "
