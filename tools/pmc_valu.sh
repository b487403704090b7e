#!/bin/bash
# Hardware VALU counters of k_accumulate (north_star: "VALU occupancy on the point-add stage"):
# one rocprofv3 --pmc pass with --kernel-trace only, the program directly after `--`.  Output: gpurun_out/pmc_valu/summary.json
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_valu
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $O/sq -- python3 bench.py --steps 3 --warmup 1 --pre-warm-ms 0 --no-cpu-baseline --no-host-legs > $O/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD \
  --kernel-trace --output-format csv -d $O/grbm -- python3 bench.py --steps 3 --warmup 1 --pre-warm-ms 0 --no-cpu-baseline --no-host-legs > $O/grbm.log 2>&1
python3 tools/pmc_valu_summarize.py $O "${BUILD:-round 5}"
