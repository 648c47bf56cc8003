#!/bin/bash
cd $GRAFT_REPO_ROOT
S=scratch/sidebuild
for i in 1 2; do
echo "== product"; python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids | head -3
echo "== delta folded into dP's accumulator"; CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/dqfold/libchadavit_hip_dqfold.so python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids | head -3
done
CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/dqfold/libchadavit_hip_dqfold.so python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "attention and not identical and not row_major" 2>&1 | tail -2
