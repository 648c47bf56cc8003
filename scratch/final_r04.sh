#!/bin/bash
# final measurement set of round 4: the driver's own command (default line with every leg + CPU baseline), the stand-alone cfg3 / cfg5 /
# cfg2-mixed lines, rocprofv3 --kernel-trace --stats of the same command on one stream, PMC passes of the dominant kernel
TAG=${1:-r04d}
O=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for w in cfg3 cfg5 cfg2-mixed; do python bench.py --workload $w --no-cpu-baseline --no-other-workloads --data resident --steps 6 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err; done
bash scratch/prof1.sh ${TAG}_default --no-other-workloads --no-full-width-leg --data resident --serial
ONE_OP_T=1206272 bash scratch/pmc.sh proj_ffn
python3 scratch/pmc_print.py gpurun_out/pmc_proj_ffn > $O/pmc_proj_ffn.txt 2>&1
for f in $O/bench_*.json; do echo $f; tail -1 $f | cut -c1-220; done
tail -20 $O/pmc_proj_ffn.txt
