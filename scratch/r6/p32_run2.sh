#!/bin/bash
# trajectory tests, then PMC (wave cycles, MFMA busy) of the forward at the headline shape: unpaired / paired, random / zero operands, dh 96 and dh 192
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_p32_2; mkdir -p $O
python -m pytest tests/test_model_gpu.py -x -q -s -k trajectory > $O/traj.log 2>&1; tail -8 $O/traj.log
pm() {  # tag op [env...]
  tag=$1; op=$2; shift; shift
  ( export "$@" GRAFT_DUMMY=1; ONE_OP_T=1206272 bash scratch/pmc.sh $op; python3 scratch/pmc_print.py gpurun_out/pmc_$op > $O/pmc_$tag.txt 2>&1; rm -rf gpurun_out/pmc_$op )
  echo "== $tag"; grep -A12 "attn_fwd" $O/pmc_$tag.txt | head -40
}
P=$GRAFT_REPO_ROOT/scratch/sidebuild/p32nr_prio/libchadavit_hip_p32nr_prio.so
pm fwd96_unpaired_random attn_fwd
pm fwd96_unpaired_zero attn_fwd ONE_OP_ZERO=1
pm fwd96_paired_random attn_fwd CHADAVIT_ATTN_FWD_PAIR32=1 CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$P
pm fwd96_paired_zero attn_fwd ONE_OP_ZERO=1 CHADAVIT_ATTN_FWD_PAIR32=1 CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$P
pm fwd192_unpaired_random attn_fwd_small
pm fwd192_paired_random attn_fwd_small CHADAVIT_ATTN_FWD_PAIR32=1 CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$P
pm fwd192_unpaired_zero attn_fwd_small ONE_OP_ZERO=1
pm fwd192_paired_zero attn_fwd_small ONE_OP_ZERO=1 CHADAVIT_ATTN_FWD_PAIR32=1 CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$P
