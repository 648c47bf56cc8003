#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "hot_kernels or block or ffn or relu" 2>&1 | tail -4
for v in 1 0 1 0; do
  export CHADA_FFN_W8=$v
  echo "== CHADA_FFN_W8=$v"; timeout 600 python bench.py --steps 6 --warmup 3 --batch 512 --no-other-workloads --no-cpu-baseline --no-full-width-leg --data resident 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print('  ', e['kernel'][:60], e['avg_us']) for e in d['launch_profile_top'][:6]]"
done | tee $O/w8_ab.log
