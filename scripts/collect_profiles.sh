#!/bin/bash
# gpurun_out/prof_r<R>/ (what scripts/profile_round.sh wrote on the GPU box) -> profiles/r<R>_* (tracked) + the two live files bench.py reads
#   bash scripts/collect_profiles.sh 6
set -e
R=${1:-6}
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r$R
cpif() { if [ -f "$1" ]; then cp "$1" "$2"; echo "  $2"; fi; }
cpif $S/bench_kernel_stats.csv profiles/r${R}_bench_default_B32768_kernel_stats.csv
cpif $S/bench_under_rocprof.json profiles/r${R}_bench_under_rocprof_B32768.json
cpif $S/traffic.json profiles/r${R}_kernels_B32768_pmc_traffic.json
cpif $S/traffic.json profiles/traffic.json
cpif $S/k1k2_padded_input_pmc_traffic.json profiles/r${R}_k1k2_padded_input_pmc_traffic.json
cpif $S/step_traffic.json profiles/r${R}_step_hbm_traffic.json
cpif $S/step_traffic.json profiles/step_traffic.json
cpif $S/kernels_pmc_sq.json profiles/r${R}_kernels_B32768_pmc_sq.json
cpif $S/kmeans_cfg5_kernel_stats.csv profiles/r${R}_kmeans_cfg5_kernel_stats.csv
cpif $S/kmeans_cfg5_pmc_traffic.json profiles/r${R}_kmeans_cfg5_pmc_traffic.json
cpif $S/cfg4_kernel_stats.csv profiles/r${R}_cfg4_kernel_stats.csv
cpif $S/cfg4_pmc_traffic.json profiles/r${R}_cfg4_pmc_traffic.json
cpif $S/cfg4_step_kernel_stats.csv profiles/r${R}_cfg4_step_kernel_stats.csv
cpif $S/step_f32x3_B32768_kernel_stats.csv profiles/r${R}_step_f32x3_B32768_kernel_stats.csv
cpif $S/step_f32x3_pmc_traffic.json profiles/r${R}_step_f32x3_pmc_traffic.json
cpif $S/step_f32x3_pmc_traffic.json profiles/x3_traffic.json
cpif $S/bench_f32x3_under_rocprof.json profiles/r${R}_bench_f32x3_under_rocprof_B32768.json
cpif $S/p2_sweep_kernel_stats.csv profiles/r${R}_p2_sweep_final_kernel_stats.csv
