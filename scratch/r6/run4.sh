#!/bin/bash
# round 6, 4th GPU call: trajectory tests; row-major-stage forward (bit identity against the fragment-major product build, timing, its no-refill
# ablation); backward ablations (what pairing could hide at most)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run4; mkdir -p $O
python -m pytest tests/test_model_gpu.py -x -q -s -k trajectory > $O/traj.log 2>&1; grep "^trajectory\|passed\|failed" $O/traj.log | cut -c1-400; grep -B2 -A12 "^E " $O/traj.log | head -40
S=scratch/sidebuild
python scratch/r6/fwd_dump.py 2>/dev/null > $O/dump_product.txt
CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/m32rm/libchadavit_hip_m32rm.so python scratch/r6/fwd_dump.py 2>/dev/null > $O/dump_m32rm.txt
echo "== identity (diff of the two dumps; empty = bit-identical)"; diff $O/dump_product.txt $O/dump_m32rm.txt | head; wc -l $O/dump_product.txt
echo "== product (fragment-major)"; python scratch/r6/p32_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
for tag in m32rm m32rm_abl1; do
  echo "== $tag"
  CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/$tag/libchadavit_hip_$tag.so python scratch/r6/p32_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
done
echo "== product (fragment-major) again"; python scratch/r6/p32_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
echo "== backward, product"; python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids
for tag in bwd_abl1 bwd_abl8 bwd_abl9 bwd_abl2 bwd_abl11; do
  echo "== backward $tag"
  CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$S/$tag/libchadavit_hip_$tag.so python scratch/r6/bwd_time.py 2>&1 | grep -v amdgpu.ids
done
