#!/bin/bash
# final measurement set of a round: bench lines for every workload + rocprofv3 kernel stats (default and one-stream) of cfg2
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --overlap --no-cpu-baseline > $O/bench_cfg2_overlap.json 2>> $O/bench_cfg2.err
for w in cfg3 cfg1 cfg5 cfg5-bf16; do python bench.py --workload $w --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err; done
bash scratch/prof1.sh ${TAG}_default
bash scratch/prof1.sh ${TAG}_overlap --overlap
for f in $O/bench_*.json; do echo $f; tail -1 $f | cut -c1-240; done
