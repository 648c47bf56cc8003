#!/bin/bash
for v in blur4096 blur3072 blur9216; do
  echo "== $v"
  CHADAVIT_HIP_LIB=$GRAFT_REPO_ROOT/chadavit_amd/build_$v/libchadavit_hip_$v.so python scratch/r4/aug_kernels_ab.py 2>&1 | grep "worst\|ms per"
done
