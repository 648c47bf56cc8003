#!/bin/bash
# usage: ab_tags.sh "<tags (product = in-tree lib)>" rounds [grep]   -- interleaved runs of scratch/r5/attn_ab_lib.py, one line per library and shape
for r in $(seq 1 ${2:-2}); do
  for tag in $1; do
    if [ "$tag" = product ]; then TAG=$tag python scratch/r5/attn_ab_lib.py | grep "${3:-.}"
    else TAG=$tag CHADAVIT_HIP_LIB=$PWD/scratch/sidebuild/$tag/libchadavit_hip_$tag.so CHADAVIT_ALLOW_FOREIGN_LIB=1 python scratch/r5/attn_ab_lib.py | grep "${3:-.}"; fi
  done
done
