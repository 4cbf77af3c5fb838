#!/bin/bash
# Build a SECOND libdic_hip.so from a variant of one csrc file (and / or extra flags) for scripts/two_lib_ab.py:
#   bash scripts/two_lib_build.sh <file.hip> <variant source | -> "<flags>" <out.so>
set -e
cd "$(dirname "$0")/../deep_interpolation_clustering_amd/csrc"
f=$1; v=$2; flags=$3; out=$4
cp $f $f.work
if [ "$v" != "-" ]; then cp "$v" $f; fi
o=${f%.hip}.o; cp $o $o.keep
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function $flags -c $f -o $o 2>/dev/null
hipcc -shared -fPIC --offload-arch=gfx950 *.o -o "$out"
cp $f.work $f; mv $o.keep $o; rm -f $f.work; touch $o
