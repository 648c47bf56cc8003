#!/bin/bash
cd $GRAFT_REPO_ROOT
S=scratch/sidebuild
echo "== product"; timeout 300 python scratch/r6/p32_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-105
for tag in m32_abl16 m32_abl32 m32_abl48; do
  echo "== $tag"
  CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/$tag/libchadavit_hip_$tag.so timeout 300 python scratch/r6/p32_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-105
done
