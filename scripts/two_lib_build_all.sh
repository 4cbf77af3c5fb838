#!/bin/bash
# Build a SECOND libdic_hip.so with extra flags on EVERY source (objects in a scratch directory) for scripts/two_lib_ab.py:
#   bash scripts/two_lib_build_all.sh "<flags>" <out.so>
set -e
cd "$(dirname "$0")/../deep_interpolation_clustering_amd/csrc"
flags=$1; out=$2
d=$(mktemp -d)
for f in *.hip; do
  extra=""; case $f in dic_rbf.hip|dic_interp.hip) extra="-fno-slp-vectorize";; esac
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function $extra $flags -c $f -o $d/${f%.hip}.o 2>/dev/null &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 $d/*.o -o "$out"
rm -rf $d
