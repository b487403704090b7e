#!/bin/bash
# Regenerates the judged evidence of a round on the GPU box (run through gpurun): bench line, rocprofv3 kernel stats of
# the same command, size sweep, host-pointer sweep, PMC traffic + VALU counters of k_accumulate, the bench's other modes, the
# window-table sweep and its per-kernel tables, per-shard device-call times.
# Outputs under gpurun_out/final/ (copy into profiles/ with the round's prefix).
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"  # default: the checkout this script lives in
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/final"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export BUILD="${BUILD:-round 6 final}"
# the counter passes FIRST: bench.py quotes profiles/*_pmc_*.json for the line's traffic figures and marks them stale when their
# source hash is not the tree's -- so the files the judged line reads are regenerated, put where it looks, and only then is the line made
bash tools/pmc_accumulate.sh 17 20 21 > $O/pmc.log 2>&1
cp gpurun_out/accumulate_pmc_2p*.json gpurun_out/sort_pmc_2p*.json $O/ 2>/dev/null
cp gpurun_out/accumulate_pmc_2p*.json gpurun_out/sort_pmc_2p*.json profiles/ 2>/dev/null
bash tools/pmc_valu.sh > $O/pmc_valu.log 2>&1
cp gpurun_out/pmc_valu/summary.json $O/accumulate_valu_pmc.json 2>/dev/null
cp gpurun_out/pmc_valu/summary.json profiles/accumulate_valu_pmc.json 2>/dev/null
python3 tools/shard_times.py --build "$BUILD" > $O/shard_times.txt 2>&1   # (the line's projected_strong_scaling reads profiles/shard_times.json)
cp gpurun_out/shard_times.json $O/shard_times.json
cp gpurun_out/shard_times.json profiles/shard_times.json
python3 bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --no-host-legs > $O/bench_under_rocprof.json 2>/dev/null  # (no host legs: k_accumulate<false,false> then only has the timed shape)
cp $O/stats/p_kernel_stats.csv $O/kernel_stats.csv
cd "$R"
python3 tools/sweep.py 10 12 13 14 15 16 > $O/sweep_small.txt 2>&1
python3 tools/host_path_sweep.py 16 18 19 20 22 > $O/host_sweep.txt 2>&1
TOPK=18 bash tools/prof_kernels.sh 17 19 21 > $O/kernel_stats_2p17_2p19_2p21.txt 2>&1
python3 tools/table_sweep.py --logn 14,16,17,18,19,20,21,22 --configs plain,0 > $O/table_sweep.txt 2>&1
TOPK=18 bash tools/prof_table.sh 20 plain 20 > $O/table_kernel_stats_2p20.txt 2>&1
TOPK=18 bash tools/prof_table.sh 17 plain 16 > $O/table_kernel_stats_2p17.txt 2>&1
python3 bench.py --gpus 2 --in-process --debug-same-device --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_in_process_2x_same_device.json 2>> $O/bench.err; echo "in-process rc $?"
python3 bench.py --log-n 22 --streamed --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_streamed_2p22.json 2>> $O/bench.err; echo "streamed rc $?"
for N in 2 4; do  # started PLAINLY: bench.py launches its own torchrun child
  python3 bench.py --gpus $N --debug-same-device --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_torchrun_${N}x_same_device.json 2>> $O/bench.err; echo "self-launched $N ranks rc $?"
done
python3 tools/sustained_probe.py > $O/sustained_probe.txt 2>&1
python3 tools/adversarial_timing.py 2>&1 | grep -v amdgpu.ids > $O/skewed_scalars.txt
python3 tools/sweep.py 17 18 19 20 21 22 24 > $O/sweep_big.txt 2>&1
bash tools/trace_device_call.sh 20 17 > $O/call_timeline_2p20_2p17.txt 2>&1
python3 tools/wide_level_probe.py 2>&1 | grep -v amdgpu.ids > $O/wide_level_breakdown.txt
ls $O
