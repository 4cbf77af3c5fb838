#!/bin/bash
# Where does dic_lstm_dx_tile's slab period go?  Timing-only rebuilds of csrc/dic_dxproj.hip (wrong results by design) timed by scripts/dx_ab.py.
# usage (GPU box): bash scripts/dx_experiments.sh
set -e
cd "$(dirname "$0")/.."
build() {
  rm -f deep_interpolation_clustering_amd/csrc/dic_dxproj.o
  make -s -C deep_interpolation_clustering_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$PWD/include -Wall -Wno-unused-function $1" > /dev/null
}
# whatever happens below (a failed compile, an interrupted run), the DEFAULT library is rebuilt on the way out: a timing-only variant is wrong by design
trap 'build ""' EXIT
if [ $# -eq 0 ]; then set -- "" "-DDIC_DXT_NA=2 -DDIC_DXT_NB=2" "-DDIC_DXT_EXP_NOB" "-DDIC_DXT_EXP_NOA" "-DDIC_DXT_EXP_NOMMA" "-DDIC_DXT_EXP_NOSTORE" "-DDIC_DXT_EXP_NOA -DDIC_DXT_EXP_NOB" ""; fi
for flags in "$@"; do
  build "$flags"; echo "== flags: [$flags]"
  python3 scripts/dx_ab.py ${DX_B:-32768} 2>/dev/null | grep "dx_tile  \|library" | tail -2
done
build ""
