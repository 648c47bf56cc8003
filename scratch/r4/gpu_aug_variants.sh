#!/bin/bash
for v in p384_b13 p384_b20 p192_b9 p288_b12 p512_b20; do
  echo "== $v"
  CHADAVIT_HIP_LIB=$GRAFT_REPO_ROOT/chadavit_amd/build_$v/libchadavit_hip_$v.so python scratch/r4/aug_kernels_ab.py 2>&1 | grep "worst\|ms per"
done
