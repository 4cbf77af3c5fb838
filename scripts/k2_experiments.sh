#!/bin/bash
# Where does rbf_bwd's time go?  Rebuild dic_rbf.hip with each flag set given as an argument (default: the phase-removal
# experiments) and time it with scripts/kbench.py.  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
build() {
  rm -f deep_interpolation_clustering_amd/csrc/dic_rbf.o
  make -s -C deep_interpolation_clustering_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$PWD/include -Wall -Wno-unused-function -fno-slp-vectorize $1" > /dev/null
}
# whatever happens below (a failed compile, an interrupted run), the DEFAULT library is rebuilt on the way out: a timing-only variant is wrong by design
trap 'build ""' EXIT
if [ $# -eq 0 ]; then set -- "" "-DDIC_K2_EXP_NOLOOP" "-DDIC_K2_EXP_NOSTAGE" "-DDIC_K2_EXP_NOLOOP -DDIC_K2_EXP_NOSTAGE"; fi
for flags in "$@"; do
  build "$flags"; echo "== flags: [$flags]"
  python scripts/kbench.py 32768 10 2>/dev/null | grep "rbf_bwd"
done
build ""
