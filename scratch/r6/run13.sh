#!/bin/bash
cd $GRAFT_REPO_ROOT
S=scratch/sidebuild; O=gpurun_out/r6_run13; mkdir -p $O
python scratch/r6/bwd_dump.py 2>/dev/null > $O/new.txt
CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/oldbwd/libchadavit_hip_oldbwd.so python scratch/r6/bwd_dump.py 2>/dev/null > $O/old.txt
echo "== identity (empty diff = bit-identical)"; diff $O/new.txt $O/old.txt | head -5; wc -l $O/new.txt
for i in 1 2; do
echo "== new"; python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids
echo "== old"; CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/oldbwd/libchadavit_hip_oldbwd.so python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids
done
