#!/bin/bash
# copies what tools/final_profiles.sh left under gpurun_out/final/ into profiles/ under the round's prefix (run HERE after the gpurun call has merged its outputs)
# usage: tools/copy_final_set.sh r6_final
cd "$(dirname "${BASH_SOURCE[0]}")/.."
P="${1:-r6_final}"; F=gpurun_out/final
cp $F/bench_line.json profiles/${P}_bench_line.json
cp $F/bench_under_rocprof.json profiles/${P}_bench_under_rocprof.json
cp $F/kernel_stats.csv profiles/${P}_kernel_stats.csv
cp $F/accumulate_pmc_2p*.json $F/sort_pmc_2p*.json profiles/
cp $F/accumulate_valu_pmc.json profiles/accumulate_valu_pmc.json
cp $F/shard_times.json profiles/shard_times.json
for f in sweep_small sweep_big host_sweep skewed_scalars call_timeline_2p20_2p17 kernel_stats_2p17_2p19_2p21 table_sweep table_kernel_stats_2p20 table_kernel_stats_2p17 sustained_probe; do cp $F/$f.txt profiles/${P}_$f.txt; done
for f in bench_in_process_2x_same_device bench_streamed_2p22 bench_torchrun_2x_same_device bench_torchrun_4x_same_device; do cp $F/$f.json profiles/${P}_$f.json; done
cp $F/wide_level_breakdown.txt profiles/r6_wide_level_breakdown.txt
cp gpurun_out/r6_final_pytest.txt profiles/${P}_pytest.txt
cp gpurun_out/r6_final_race_hunt.txt profiles/${P}_race_hunt.txt
