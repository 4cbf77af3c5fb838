#!/bin/bash
# Rebuild dic_lstmgrad.hip with each flag set given as an argument and time the decoder dW kernel alone.  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
build() {
  rm -f deep_interpolation_clustering_amd/csrc/dic_lstmgrad.o
  make -s -C deep_interpolation_clustering_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$PWD/include -Wall -Wno-unused-function $1" > /dev/null
}
# whatever happens below (a failed compile, an interrupted run), the DEFAULT library is rebuilt on the way out: a timing-only variant is wrong by design
trap 'build ""' EXIT
if [ $# -eq 0 ]; then set -- "" "-DDIC_DWW_EXP_NOMMA"; fi
for flags in "$@"; do
  build "$flags"; echo "== flags: [$flags]"
  python scripts/dww_timing.py 2>/dev/null
  python scripts/kbench.py 32768 20 2>/dev/null | grep "lstm_dw "
done
build ""
