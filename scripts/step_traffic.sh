#!/bin/bash
# HBM traffic of the whole step (two PMC passes), per group and per kernel.  usage (GPU box, repo root): bash scripts/step_traffic.sh [round]
set -e
R=${1:-2}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/s_fetch -o p -- python3 $ROOT/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2 --kernel-iters 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/s_write -o p -- python3 $ROOT/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2 --kernel-iters 1 > /dev/null 2>&1
python3 $ROOT/scripts/step_traffic.py /tmp/s_fetch/p_counter_collection.csv /tmp/s_write/p_counter_collection.csv $R
