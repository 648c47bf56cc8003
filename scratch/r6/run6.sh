#!/bin/bash
cd $GRAFT_REPO_ROOT
S=scratch/sidebuild; O=gpurun_out/r6_run6; mkdir -p $O
python scratch/r6/bwd_dump.py 2>/dev/null > $O/bwd_product.txt
CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/nopipe/libchadavit_hip_nopipe.so python scratch/r6/bwd_dump.py 2>/dev/null > $O/bwd_nopipe.txt
echo "== identity of the pipelined backward against the unpipelined one (empty diff = bit-identical)"; diff $O/bwd_product.txt $O/bwd_nopipe.txt | head; wc -l $O/bwd_product.txt
for tag in nopipe dqpipe; do
  echo "== backward $tag"
  CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/$tag/libchadavit_hip_$tag.so python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids
done
echo "== backward, product (dQ and dK/dV pipelined)"; python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids
echo "== backward nopipe again"
CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/nopipe/libchadavit_hip_nopipe.so python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids
