#!/bin/bash
# VERDICT r2 item 5, measured: what would the encoder gain if its backward RECOMPUTED the gates instead of reading the saved ones?
# Timing-only build (-DDIC_LSTM_EXP_RECOMPUTE, wrong results by design): the fused-projection forward stops saving the gates (1 024 of its
# 1 568 B per unit); the backward reads one 2-B-per-unit plane (the bytes of the h_prev row it would read instead), re-evaluates the four
# activations (8 transcendentals per unit) and issues 80 more MFMAs per wave and step -- optimistic: operands already resident.
# usage (GPU box):  bash scripts/lstm_recompute_ab.sh
set -e
cd "$(dirname "$0")/.."
build() {
  rm -f deep_interpolation_clustering_amd/csrc/dic_lstm.o
  make -s -C deep_interpolation_clustering_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$PWD/include -Wall -Wno-unused-function $1" > /dev/null
}
# whatever happens below (a failed compile, an interrupted run), the DEFAULT library is rebuilt on the way out: a timing-only variant is wrong by design
trap 'build ""' EXIT
for flags in "" "-DDIC_LSTM_EXP_RECOMPUTE" "" "-DDIC_LSTM_EXP_RECOMPUTE"; do
  build "$flags"; echo "== flags: [$flags]"
  python scripts/lstm_ab.py 2>&1 | grep -v amdgpu.ids | tail -1
done
build ""
