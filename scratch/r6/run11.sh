#!/bin/bash
cd $GRAFT_REPO_ROOT
S=scratch/sidebuild; O=gpurun_out/r6_run11; mkdir -p $O
A="CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/ab/libchadavit_hip_ab.so"
python scratch/r6/bwd_dump.py 2>/dev/null > $O/bwd_product.txt
env $A CHADAVIT_ATTN_DKV_PAIR=1 python scratch/r6/bwd_dump.py 2>/dev/null > $O/bwd_pair.txt
echo "== identity of the paired dK/dV against the product (dh 96 cases differ only if the kernel is wrong)"; diff $O/bwd_product.txt $O/bwd_pair.txt | head; wc -l $O/bwd_pair.txt
echo "== unpaired (side build, switch off)"; env $A python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids | head -4
echo "== paired dK/dV"; env $A CHADAVIT_ATTN_DKV_PAIR=1 python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids | head -4
echo "== paired dK/dV, zero operands"; env $A CHADAVIT_ATTN_DKV_PAIR=1 python scratch/r6/bwd_time.py zero 2>&1 | grep -v amdgpu.ids | head -4
echo "== unpaired, zero operands"; env $A python scratch/r6/bwd_time.py zero 2>&1 | grep -v amdgpu.ids | head -4
