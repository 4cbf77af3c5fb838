#!/bin/bash
# A/B a csrc file on one box: committed version (saved beforehand as <file>.base: git show HEAD:<path> > <path>.base) vs the
# working copy.   usage (GPU box):  bash scripts/ab_kernel.sh dic_lstm.hip "python scripts/kbench.py 32768 10" lstm
set -e
cd "$(dirname "$0")/.."
f=deep_interpolation_clustering_amd/csrc/$1
cp $f $f.work
for v in base work base work; do
  cp $f.$v $f; touch $f
  make -s -C deep_interpolation_clustering_amd/csrc > /dev/null 2>&1
  echo "== $v"; $2 2>/dev/null | grep "$3"
done
cp $f.work $f; touch $f; make -s -C deep_interpolation_clustering_amd/csrc > /dev/null 2>&1
