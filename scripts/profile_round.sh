#!/bin/bash
# Everything the judged profiles/ files come from, in one GPU lease (run from the repo root on the GPU box):
#   bash scripts/profile_round.sh 6            (every step)      bash scripts/profile_round.sh 6 "1 2 3"   (some: a gpurun call is limited to 20 minutes)
# Counters are collected in passes of their own (--pmc with --kernel-trace only), as gpurun requires.
set -e
R=${1:-6}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${2:-"1 2 3 4 5 6 7 8 9"}
step1() {
  # 1. kernel statistics of the bench command itself
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o b -- python3 $ROOT/bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 9 --kernel-iters 1 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
  cp /tmp/p_stats/b_kernel_stats.csv $OUT/bench_kernel_stats.csv
  echo "[profile] kernel stats done"
}
step2() {
  # 2. HBM traffic per kernel (micro table at the bench batch)
  # (k1 / k2 on the ragged encounter store, the input path of the timed step; the padded-input launches of the same kernels in a pass of their own)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -o p -- python3 $ROOT/scripts/kbench.py 32768 3 4 store > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -o p -- python3 $ROOT/scripts/kbench.py 32768 3 4 store > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_traffic.py /tmp/p_fetch/p_counter_collection.csv /tmp/p_write/p_counter_collection.csv 32768 $R > $OUT/traffic.txt
  cp $ROOT/profiles/traffic.json $OUT/traffic.json
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/d_fetch -o p -- python3 $ROOT/scripts/kbench.py 32768 3 4 nolstm dense > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/d_write -o p -- python3 $ROOT/scripts/kbench.py 32768 3 4 nolstm dense > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_generic.py /tmp/d_fetch/p_counter_collection.csv /tmp/d_write/p_counter_collection.csv 'sci_cci|rbf_' > $OUT/k1k2_padded_input_pmc_traffic.json
  echo "[profile] kernel traffic done"
}
step3() {
  # 3. HBM traffic of the whole step
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/s_fetch -o p -- python3 $ROOT/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 3 --kernel-iters 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/s_write -o p -- python3 $ROOT/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 3 --kernel-iters 1 > /dev/null 2>&1
  python3 $ROOT/scripts/step_traffic.py /tmp/s_fetch/p_counter_collection.csv /tmp/s_write/p_counter_collection.csv $R > $OUT/step_traffic.txt
  cp $ROOT/profiles/step_traffic.json $OUT/step_traffic.json
  echo "[profile] step traffic done"
}
step4() {
  # 4. SQ counters of the hand-written kernels at the bench batch
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/sq1 -o p -- python3 $ROOT/scripts/kbench.py 32768 3 4 store > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sq2 -o p -- python3 $ROOT/scripts/kbench.py 32768 3 4 store > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_sq.py $OUT/kernels_pmc_sq.json /tmp/sq1/p_counter_collection.csv /tmp/sq2/p_counter_collection.csv > $OUT/kernels_pmc_sq.txt
  echo "[profile] SQ counters done"
}
step5() {
  # 5. the k-means kernels on cfg5's own data: kernel statistics + HBM traffic (the Lloyd record of bench.py's cfg5)
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/km_stats -o k -- python3 $ROOT/scripts/kmeans_prof.py 5 > $OUT/kmeans_prof.txt 2>&1
  cp /tmp/km_stats/k_kernel_stats.csv $OUT/kmeans_cfg5_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/km_fetch -o p -- python3 $ROOT/scripts/kmeans_prof.py 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/km_write -o p -- python3 $ROOT/scripts/kmeans_prof.py 3 > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_generic.py /tmp/km_fetch/p_counter_collection.csv /tmp/km_write/p_counter_collection.csv kmeans > $OUT/kmeans_cfg5_pmc_traffic.json
  echo "[profile] cfg5 k-means done"
}
step6() {
  # 6. cfg4 (C=12, T=288, ~200 obs/channel, K=16): k1 / k2 / k3 kernel statistics + HBM traffic at its batch of 8192
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4_stats -o k -- python3 $ROOT/scripts/kbench.py 8192 5 16 12 288 200 nolstm > $OUT/cfg4_kbench.txt 2>&1
  cp /tmp/c4_stats/k_kernel_stats.csv $OUT/cfg4_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/c4_fetch -o p -- python3 $ROOT/scripts/kbench.py 8192 3 16 12 288 200 nolstm > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/c4_write -o p -- python3 $ROOT/scripts/kbench.py 8192 3 16 12 288 200 nolstm > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_generic.py /tmp/c4_fetch/p_counter_collection.csv /tmp/c4_write/p_counter_collection.csv 'sci_cci|rbf_|masked_sse|dec_' > $OUT/cfg4_pmc_traffic.json
  echo "[profile] cfg4 done"
}
step7() {
  # 7. the joint step at cfg4's shape (C=12 -> 64-wide packed encoder rows): kernel statistics
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4s_stats -o k -- python3 $ROOT/scripts/cfg4_step.py 20 > $OUT/cfg4_step.txt 2>&1
  cp /tmp/c4s_stats/k_kernel_stats.csv $OUT/cfg4_step_kernel_stats.csv
  echo "[profile] cfg4 step done"
}
step8() {
  # 8. the f32 step with every dense product as a three-term bf16 split (--dtype f32x3) at the headline batch: kernel statistics + HBM traffic of its
  #    recurrence / product kernels (65 536 encounters = two FULL batches per epoch: every launch of the profiled run moves 32 768 encounters)
  X3="--dtype f32x3 --encounters 65536 --no-secondary --no-cpu-baseline --kernel-iters 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/x3_stats -o k -- python3 $ROOT/bench.py $X3 --steps 10 --warmup 4 > $OUT/bench_f32x3_under_rocprof.json 2> /dev/null
  cp /tmp/x3_stats/k_kernel_stats.csv $OUT/step_f32x3_B32768_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/x3_fetch -o p -- python3 $ROOT/bench.py $X3 --steps 4 --warmup 2 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/x3_write -o p -- python3 $ROOT/bench.py $X3 --steps 4 --warmup 2 > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_generic.py /tmp/x3_fetch/p_counter_collection.csv /tmp/x3_write/p_counter_collection.csv 'lstm_rec_|lstm_dwx3|dx_tile_x3|gemm_|x3_row_proj|bnhead|bn_colstats' 32768 > $OUT/step_f32x3_pmc_traffic.json
  cp $OUT/step_f32x3_pmc_traffic.json $ROOT/profiles/x3_traffic.json
  echo "[profile] f32x3 done"
}
step9() {
  # 9. configs[4]: p2's K sweep (K = 2..20, 10 reference sets, three indices per K) on 75 000 x 256 latents: kernel statistics
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2_stats -o p2 -- python3 $ROOT/scripts/p2_sweep_75k.py > $OUT/p2_sweep.txt 2>&1
  cp /tmp/p2_stats/p2_kernel_stats.csv $OUT/p2_sweep_kernel_stats.csv
  echo "[profile] p2 sweep done"
}
for s in $STEPS; do step$s; done
ls -la $OUT
