#!/bin/bash
cd $GRAFT_REPO_ROOT
S=scratch/sidebuild
echo "== product"; python scratch/r3/mx8_bench.py 2>&1 | grep -v amdgpu.ids
echo "== ping-pong"; CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/mxpp/libchadavit_hip_mxpp.so python scratch/r3/mx8_bench.py 2>&1 | grep -v amdgpu.ids
echo "== product"; python scratch/r3/mx8_bench.py 2>&1 | grep -v amdgpu.ids
echo "== ping-pong"; CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/mxpp/libchadavit_hip_mxpp.so python scratch/r3/mx8_bench.py 2>&1 | grep -v amdgpu.ids
echo "== tests with the ping-pong library"
CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/mxpp/libchadavit_hip_mxpp.so python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "mx8" 2>&1 | tail -3
CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/mxpp/libchadavit_hip_mxpp.so python -m pytest tests/test_model_gpu.py -q -x -m gpu -k "fp8" 2>&1 | tail -3
