#!/bin/bash
# A/B a csrc file on one box: committed version (git show HEAD:<file> saved beforehand as <file>.base) vs working copy.
# usage (here): git show HEAD:deep_interpolation_clustering_amd/csrc/dic_lstm.hip > deep_interpolation_clustering_amd/csrc/dic_lstm.hip.base
#        (GPU):  bash scripts/ab_kernel.sh dic_lstm.hip lstm
set -e
cd "$(dirname "$0")/.."
f=deep_interpolation_clustering_amd/csrc/$1
cp $f $f.work
for v in base work base work; do
  cp $f.$v $f; touch $f
  make -s -C deep_interpolation_clustering_amd/csrc > /dev/null 2>&1
  echo "== $v"; python scripts/kbench.py 32768 10 2>/dev/null | grep "$2"
done
cp $f.work $f; touch $f; make -s -C deep_interpolation_clustering_amd/csrc > /dev/null 2>&1
