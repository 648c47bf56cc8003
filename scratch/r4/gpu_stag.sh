#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
for v in base stag24 stag48 stag100; do
  if [ $v = base ]; then unset CHADAVIT_HIP_LIB; else export CHADAVIT_HIP_LIB=$PWD/chadavit_amd/build_$v/lib_$v.so; fi
  echo "== $v"; timeout 600 python scratch/r4/attn_bwd_m32.py no-child time-only 2>&1 | grep "us (" | sed -n '2p;6p' | tee $O/bwd_$v.log
  timeout 600 python scratch/r3/attn_m32.py 2>&1 | grep "variant 0" | grep "us " | sed -n '2p;6p' | tee $O/fwd_$v.log
done
