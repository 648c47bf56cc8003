#!/bin/bash
# gemm_mx8: product build (and, when present, -DMX_ABL=<mask> builds under chadavit_amd/build_abl<mask>/) at the cfg5 shapes on one box
mkdir -p gpurun_out/r3
for v in 0 ${MX_ABLS}; do
  if [ $v = 0 ]; then unset CHADAVIT_HIP_LIB; else export CHADAVIT_HIP_LIB=$PWD/chadavit_amd/build_abl$v/libchadavit_hip_abl$v.so; fi
  echo "== MX_ABL=$v"; timeout 200 python scratch/r3/mx8_bench.py
done > gpurun_out/r3/mx8_ablate.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3/mx8_ablate.txt
