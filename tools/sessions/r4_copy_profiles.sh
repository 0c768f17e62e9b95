#!/bin/bash
# build container: gpurun_out/r04_bench_all + gpurun_out/r04_profiles (tools/sessions/gpu_r4_prof.sh on the GPU box) -> the tracked profiles/r04_* files
cd "$(dirname "$0")/.."
for f in gpurun_out/r04_bench_all/*.json; do cp $f profiles/r04_bench_$(basename $f); done
P=gpurun_out/r04_profiles
for cfg in large_b1_f32 large_b8_f32; do
  cp $P/bench_$cfg.json profiles/r04_bench_$cfg.json
  cp $P/classes_$cfg.txt profiles/r04_kernel_classes_$cfg.txt
  cp $P/kernel_stats_$cfg.txt profiles/r04_kernel_stats_$cfg.txt
  cp $P/pmc_$cfg.json profiles/r04_pmc_$cfg.json; cp $P/pmc_$cfg.txt profiles/r04_pmc_$cfg.txt
  cp $P/layers_$cfg.txt profiles/r04_layers_$cfg.txt
done
cp $P/instance_norm_bench.txt profiles/r04_instance_norm_bench.txt
ls profiles | grep -c r04
