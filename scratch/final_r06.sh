#!/bin/bash
# final measurement set of round 6: the full GPU test suite (+ the slow ones), the driver's own command, rocprofv3 --kernel-trace --stats of the same command
# on one stream AND of the cfg3 / cfg5 / cfg2-mixed workloads (by-grid summaries: every leg's fraction recomputable from profiles/), PMC passes of the dominant
# kernel and of the attention kernels at the headline's shape
TAG=${1:-r06a}
O=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
if [ "$2" != "notests" ]; then
  ( time python -m pytest tests -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
  ( time python -m pytest tests -q -m "gpu and slow" ) > $O/pytest_gpu_slow.log 2>&1; tail -3 $O/pytest_gpu_slow.log
fi
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err
tail -1 $O/bench_cfg2.json | cut -c1-300; tail -4 $O/bench_cfg2.err
bash scratch/prof1.sh ${TAG}_serial_cfg2 --no-other-workloads --no-full-width-leg --data resident --serial
for wl in cfg3 cfg5 cfg2-mixed; do
  bash scratch/prof1.sh ${TAG}_serial_$wl --workload $wl --no-other-workloads --no-full-width-leg --data resident --serial
done
for d in cfg2 cfg3 cfg5 cfg2-mixed; do echo "== $d"; head -12 gpurun_out/prof_${TAG}_serial_$d/kernel_stats_by_grid.csv | cut -c1-160; tail -1 gpurun_out/prof_${TAG}_serial_$d/bench.json | cut -c1-200; done
ONE_OP_T=1206272 bash scratch/pmc.sh proj_ffn;  python3 scratch/pmc_print.py gpurun_out/pmc_proj_ffn > $O/pmc_proj_ffn.txt 2>&1
ONE_OP_T=1206272 bash scratch/pmc.sh attn_bwd;  python3 scratch/pmc_print.py gpurun_out/pmc_attn_bwd > $O/pmc_attn_bwd.txt 2>&1
ONE_OP_T=1206272 bash scratch/pmc.sh attn_fwd;  python3 scratch/pmc_print.py gpurun_out/pmc_attn_fwd > $O/pmc_attn_fwd.txt 2>&1
bash scratch/pmc.sh attn_bwd_small; python3 scratch/pmc_print.py gpurun_out/pmc_attn_bwd_small > $O/pmc_attn_bwd_small.txt 2>&1
bash scratch/pmc.sh attn_bwd_base; python3 scratch/pmc_print.py gpurun_out/pmc_attn_bwd_base > $O/pmc_attn_bwd_base.txt 2>&1
cat $O/pmc_proj_ffn.txt | head -16
