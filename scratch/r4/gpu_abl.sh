#!/bin/bash
# compile-time ablations of the default dK/dV kernel (wrong results, timing only): 1 = no exp2 / fma in the softmax, 2 = LDS fragment reads
# only once per tile, 4 = no LDS-DMA, 8 = dV / dK MFMAs only for the first head-dim block, 16 = S / dP MFMAs only for the first k-step
O=gpurun_out/r4m; mkdir -p $O
for a in 0 1 2 4 8 16 24 27 31; do
  if [ $a = 0 ]; then unset CHADAVIT_HIP_LIB; else export CHADAVIT_HIP_LIB=$PWD/chadavit_amd/build_abl$a/lib_abl$a.so; fi
  echo -n "ABL=$a  "; timeout 300 python scratch/r4/attn_bwd_m32.py no-child time-only 2>&1 | grep "us (" | head -2 | tail -1 | sed 's/.*dK\/dV *\([0-9.]*\)).*/dK\/dV \1 us/'
done | tee $O/ablations.log
