#!/bin/bash
# round 6, third GPU call: trajectory tests, the kernel tests that the A/B-switch refactor touches, PMC of the paired forward, ablations of the
# unpaired forward, a short default bench (the top entry points' in-step durations)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run3; mkdir -p $O
python -m pytest tests/test_model_gpu.py -x -q -s -k trajectory > $O/traj.log 2>&1; grep "^trajectory\|passed\|failed" $O/traj.log; grep -B2 -A12 "^E " $O/traj.log | head -40
python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "attention or gemm_tn or hot_kernels or tn" > $O/kernels.log 2>&1; tail -4 $O/kernels.log
for tag in m32_abl1 m32_abl2 m32_abl8 m32_abl9 m32_abl11; do
  echo "== unpaired ablation $tag"
  CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=scratch/sidebuild/$tag/libchadavit_hip_$tag.so python scratch/r6/p32_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-130
done
pm() {  # tag op [env...]
  tag=$1; op=$2; shift; shift
  ( export "$@" GRAFT_DUMMY=1; ONE_OP_T=1206272 bash scratch/pmc.sh $op; python3 scratch/pmc_print.py gpurun_out/pmc_$op > $O/pmc_$tag.txt 2>&1; rm -rf gpurun_out/pmc_$op )
  echo "== $tag"; head -14 $O/pmc_$tag.txt
}
P=$GRAFT_REPO_ROOT/scratch/sidebuild/p32nr_prio/libchadavit_hip_p32nr_prio.so
pm fwd96_paired_random attn_fwd CHADAVIT_ATTN_FWD_PAIR32=1 CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$P
pm fwd96_paired_zero attn_fwd ONE_OP_ZERO=1 CHADAVIT_ATTN_FWD_PAIR32=1 CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$P
pm fwd192_paired_random attn_fwd_small CHADAVIT_ATTN_FWD_PAIR32=1 CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$P
pm fwd192_paired_zero attn_fwd_small ONE_OP_ZERO=1 CHADAVIT_ATTN_FWD_PAIR32=1 CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$P
( time python bench.py --gpus 1 --steps 8 --warmup 3 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_run3/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
for r in d["launch_profile_top"]: print(r)
for k,v in d["config"]["other_workloads"].items(): print(k, {x:v.get(x) for x in ("images_per_s","dominant_kernel","frac","dominant_avg_us")})
PY
