#!/bin/bash
# same-box A/B of two library builds on the attention timing script (alternating)
mkdir -p gpurun_out/r3
for rep in 1 2; do
  for v in old new; do
    echo "== $v run $rep"
    CHADAVIT_HIP_LIB=$PWD/chadavit_amd/build_ab/lib$v.so python scratch/attn_time.py 2>&1 | grep -v amdgpu.ids
  done
done
