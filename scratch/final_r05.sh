#!/bin/bash
# final measurement set of round 5: the driver's own command (default line with every leg + CPU baseline), rocprofv3 --kernel-trace --stats of
# the same command on one stream (serial: per-kernel durations are stand-alone ones), PMC passes of the dominant kernel and of the attention
# kernels (forward, backward pair) at the headline's shape, the `-m "gpu and slow"` tests
TAG=${1:-r05c}
O=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err
bash scratch/prof1.sh ${TAG}_serial --no-other-workloads --no-full-width-leg --data resident --serial
ONE_OP_T=1206272 bash scratch/pmc.sh proj_ffn;  python3 scratch/pmc_print.py gpurun_out/pmc_proj_ffn > $O/pmc_proj_ffn.txt 2>&1
ONE_OP_T=1206272 bash scratch/pmc.sh attn_bwd;  python3 scratch/pmc_print.py gpurun_out/pmc_attn_bwd > $O/pmc_attn_bwd.txt 2>&1
ONE_OP_T=1206272 bash scratch/pmc.sh attn_fwd;  python3 scratch/pmc_print.py gpurun_out/pmc_attn_fwd > $O/pmc_attn_fwd.txt 2>&1
tail -1 $O/bench_cfg2.json | cut -c1-300
tail -4 $O/bench_cfg2.err
head -30 gpurun_out/prof_${TAG}_serial/kernel_stats.csv | cut -c1-200
cat $O/pmc_attn_bwd.txt $O/pmc_attn_fwd.txt | head -70
