#!/bin/bash
# Where does sci_cci_fwd's time go?  Rebuild dic_interp.hip with each flag set given as an argument (default: with and without the
# streaming passes) and time it with scripts/kbench.py; "timing" = per-phase cycle stamps of the generic tile kernel.  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
build() {
  rm -f deep_interpolation_clustering_amd/csrc/dic_interp.o
  make -s -C deep_interpolation_clustering_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$PWD/include -Wall -Wno-unused-function -fno-slp-vectorize $1" > /dev/null
}
# whatever happens below (a failed compile, an interrupted run), the DEFAULT library is rebuilt on the way out: a timing-only variant is wrong by design
trap 'build ""' EXIT
if [ "$1" == "timing" ]; then
  build "-DDIC_K1_EXP_TIMING"; python scripts/k1_timing.py 2>&1 | grep -v amdgpu.ids
  build ""; exit 0
fi
if [ $# -eq 0 ]; then set -- "" "-DDIC_K1_EXP_NOLOOP"; fi
for flags in "$@"; do
  build "$flags"; echo "== flags: [$flags]"
  python scripts/kbench.py 32768 10 2>/dev/null | grep "sci_cci_fwd"
done
build ""
