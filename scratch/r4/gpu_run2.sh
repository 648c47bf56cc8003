#!/bin/bash
# round 4, second GPU call: the 32x32x16 attention backward (correctness + timing vs the 16x16x32 pair), the full GPU suite, the default bench
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python scratch/r4/attn_bwd_m32.py 2>&1 | tee $O/attn_bwd_m32.log | tail -40
python -m pytest tests -m gpu -x -q -s 2>&1 | tee $O/pytest_gpu.log | tail -15
timeout 900 python bench.py --steps 10 --warmup 3 --no-other-workloads > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -c 3000 $O/bench_cfg2.json
