#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
python -m pytest tests -m gpu -x -q -s 2>&1 | tee $O/pytest_gpu.log | tail -8
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -c 1500 $O/bench_cfg2.json
