#!/bin/bash
# Where does lstm_fwdx8's step go?  Timing-only rebuilds of csrc/dic_lstm32.hip (wrong results by design) timed by scripts/fwdx_ab.py
# at B = 32768, with and without the saved-state stores.  usage (GPU box): bash scripts/fwdx_experiments.sh
set -e
cd "$(dirname "$0")/.."
build() {
  rm -f deep_interpolation_clustering_amd/csrc/dic_lstm32.o
  make -s -C deep_interpolation_clustering_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$PWD/include -Wall -Wno-unused-function $1" > /dev/null
}
# whatever happens below (a failed compile, an interrupted run), the DEFAULT library is rebuilt on the way out: a timing-only variant is wrong by design
trap 'build ""' EXIT
if [ "$1" == "timing" ]; then
  build "-DDIC_FWDX_EXP_TIMING"; python3 scripts/fwdx_timing.py 2>/dev/null; python3 scripts/fwdx_timing.py nosave 2>/dev/null
  build ""; exit 0
fi
if [ $# -eq 0 ]; then set -- "" "-DDIC_FWDX_EXP_NOTRANS" "-DDIC_FWDX_EXP_NOGATE" "-DDIC_FWDX_EXP_NOPROJ" "-DDIC_FWDX_EXP_NOXLOAD" "-DDIC_FWDX_EXP_NOPROJ -DDIC_FWDX_EXP_NOGATE" "-fno-slp-vectorize" ""; fi
for flags in "$@"; do
  build "$flags"; echo "== flags: [$flags]"
  FWDX_NOSAVE=1 python3 scripts/fwdx_ab.py 32768 2>/dev/null | grep "us"
done
build ""
