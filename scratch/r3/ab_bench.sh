#!/bin/bash
# quick check: the kernel tests that touch the changed library + the default bench twice
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "${KSEL:-ffn or block or proj or mx8}" > gpurun_out/r3/pytest_ab.txt 2>&1; tail -3 gpurun_out/r3/pytest_ab.txt
for rep in 1 2; do
  BENCH_TOP=12 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-full-width-leg --no-other-workloads > gpurun_out/r3/ab_new_${rep}.json 2> gpurun_out/r3/ab_new_${rep}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3/ab_new_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"] if "kernel" in d["roofline"] else "", d["roofline"].get("achieved"), d["roofline"].get("frac"))
    except Exception as e: print(f, "ERR", e)
PY
