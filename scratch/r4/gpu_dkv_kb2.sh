#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
export CHADAVIT_ATTN_BWD_M32=1
echo "== m32, 32 keys per wave (2 waves/SIMD)"; timeout 600 python scratch/r4/attn_bwd_m32.py no-child time-only 2>&1 | grep "us (" | tee $O/kb1.log
export CHADAVIT_ATTN_DKV_M32_KB2=1
echo "== m32, 64 keys per wave (1 wave/SIMD), fences"; timeout 600 python scratch/r4/attn_bwd_m32.py no-child 2>&1 | grep -E "us \(|rel" | tee $O/kb2.log
echo "== same, inner fences off"; CHADAVIT_HIP_LIB=$PWD/chadavit_amd/build_nofence/libchadavit_hip_nofence.so timeout 600 python scratch/r4/attn_bwd_m32.py no-child time-only 2>&1 | grep "us (" | tee $O/kb2_nofence.log
