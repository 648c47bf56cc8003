#!/bin/bash
# final measurement set of round 3: the default bench line (with the cfg3 / cfg5 / cfg1 legs, CPU baseline) + the data-path leg,
# rocprofv3 kernel stats of the same command on one stream, PMC passes of the dominant kernel and of the new attention forward
TAG=${1:-r03c}
O=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
BENCH_TOP=16 python bench.py --steps 20 --warmup 5 --data pipeline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for w in cfg3 cfg5 cfg5-bf16 cfg2-mixed; do python bench.py --workload $w --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err; done
bash scratch/prof1.sh ${TAG}_default --no-other-workloads --no-full-width-leg
ONE_OP_T=603136 bash scratch/pmc.sh proj_ffn
ONE_OP_T=603136 bash scratch/pmc.sh attn_fwd
for f in $O/bench_*.json; do echo $f; tail -1 $f | cut -c1-200; done
