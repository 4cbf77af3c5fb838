#!/bin/bash
# Build a SECOND libdic_hip.so from a variant of one csrc file (and / or extra flags) for scripts/two_lib_ab.py:
#   bash scripts/two_lib_build.sh <file.hip> <variant source | -> "<flags>" <out.so>
# Nothing in the tree is touched: the variant is compiled from where it lies into a scratch directory and linked with the tree's other objects
# (run `make -C deep_interpolation_clustering_amd/csrc` first so that those are current).  A failed compile prints the compiler's messages and
# leaves the tree and the default library as they were.
set -euo pipefail
csrc="$(cd "$(dirname "$0")/../deep_interpolation_clustering_amd/csrc" && pwd)"
f=$1; v=$2; flags=$3; out=$4
src="$csrc/$f"; if [ "$v" != "-" ]; then src="$(realpath "$v")"; fi
d=$(mktemp -d); trap 'rm -rf "$d"' EXIT
extra=""; case $f in dic_rbf.hip|dic_interp.hip) extra="-fno-slp-vectorize";; esac
cp "$src" "$d/$f"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$csrc" -I"$csrc/../../include" -Wall -Wno-unused-function $extra $flags -c "$d/$f" -o "$d/${f%.hip}.o"
objs=""
for o in "$csrc"/*.o; do
  if [ "$(basename "$o")" = "${f%.hip}.o" ]; then objs="$objs $d/${f%.hip}.o"; else objs="$objs $o"; fi
done
hipcc -shared -fPIC --offload-arch=gfx950 $objs -o "$out"
echo "built $out ($f from $src, flags: $flags)"
