#!/bin/bash
# first GPU round of the paired 32x32x16 forward: identity against the unpaired kernel, then timings of the product build and the side builds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_p32_1; mkdir -p $O
python scratch/r6/p32_time.py check > $O/check.log 2>&1; tail -30 $O/check.log
python scratch/r6/p32_time.py > $O/time_product.log 2>&1; cat $O/time_product.log
for tag in "$@"; do
  echo "== side build $tag"
  CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=scratch/sidebuild/$tag/libchadavit_hip_$tag.so python scratch/r6/p32_time.py > $O/time_$tag.log 2>&1; cat $O/time_$tag.log
done
CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=scratch/sidebuild/p32nr/libchadavit_hip_p32nr.so python scratch/r6/p32_time.py zero > $O/time_p32nr_zero.log 2>&1; echo "== zero operands (p32nr)"; cat $O/time_p32nr_zero.log
