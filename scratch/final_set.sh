#!/bin/bash
# final measurement set of round 2 (last build): bench lines for every workload + rocprofv3 kernel stats of the default run
TAG=${1:-r02k}
O=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for w in cfg3 cfg1 cfg5 cfg5-bf16 cfg2-mixed; do python bench.py --workload $w --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err; done
bash scratch/prof1.sh ${TAG}_default
for f in $O/bench_*.json; do echo $f; tail -1 $f | cut -c1-200; done
