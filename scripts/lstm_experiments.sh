#!/bin/bash
# Where does lstm_fwd's time go?  Rebuild the library with experiment switches and time the kernels (scripts/kbench.py).
set -e
cd "$(dirname "$0")/.."
build() {
  rm -f deep_interpolation_clustering_amd/csrc/dic_lstm.o
  make -s -C deep_interpolation_clustering_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$PWD/include -Wall -Wno-unused-function $1" > /dev/null
}
# whatever happens below (a failed compile, an interrupted run), the DEFAULT library is rebuilt on the way out: a timing-only variant is wrong by design
trap 'build ""' EXIT
if [ "$1" == "timing" ]; then
  for flags in "-DDIC_LSTM_EXP_TIMING" "-DDIC_LSTM_EXP_TIMING -DDIC_LSTM_EXP_NOMATH -DDIC_LSTM_EXP_NOSTORE"; do
    build "$flags"; echo "== flags: [$flags]"
    python scripts/lstm_timing.py 2>&1 | grep -v amdgpu.ids; python scripts/lstm_timing.py proj 2>&1 | grep -v amdgpu.ids
  done
else
  for flags in "" "-DDIC_LSTM_EXP_NOMATH" "-DDIC_LSTM_EXP_NOSTORE" "-DDIC_LSTM_EXP_NOMATH -DDIC_LSTM_EXP_NOSTORE"; do
    build "$flags"; echo "== flags: [$flags]"
    python scripts/kbench.py 32768 10 | grep lstm
  done
fi
build ""
