#!/bin/bash
# usage: ab_run.sh <script.py> "<tags (product = in-tree lib)>" rounds   -- interleaved runs of one timing script, one line per library
for r in $(seq 1 ${3:-2}); do
  for tag in $2; do
    if [ "$tag" = product ]; then TAG=$tag python $1
    else TAG=$tag CHADAVIT_HIP_LIB=$PWD/scratch/sidebuild/$tag/libchadavit_hip_$tag.so CHADAVIT_ALLOW_FOREIGN_LIB=1 python $1; fi
  done
done
