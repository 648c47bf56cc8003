#!/bin/bash
O=gpurun_out/r4f; mkdir -p $O
echo "=== product (64-key tiles, 3 waves/SIMD)"; timeout 600 python scratch/r3/attn_m32.py 2>&1 | grep -v amdgpu.ids | tee $O/fwd_default.log | grep -E "FAIL|us " 
echo "=== 32-key tiles, 4 waves/SIMD"; CHADAVIT_HIP_LIB=$PWD/chadavit_amd/build_kv32/libchadavit_hip_kv32.so timeout 600 python scratch/r3/attn_m32.py 2>&1 | grep -v amdgpu.ids | tee $O/fwd_kv32.log | grep -E "FAIL|us "
