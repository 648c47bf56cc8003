#!/bin/bash
# same-box A/B: product library vs the same sources built without packed-f32 VALU ops (-target-feature -packed-fp32-ops)
mkdir -p gpurun_out/r3
timeout 120 ./scratch/r3/coissue2 > gpurun_out/r3/coissue2.txt 2>&1
for rep in 1 2; do
  for v in pk nopk; do
    if [ $v = nopk ]; then export CHADAVIT_HIP_LIB=$PWD/chadavit_amd/build_nopk/libchadavit_hip_nopk.so; else unset CHADAVIT_HIP_LIB; fi
    BENCH_TOP=24 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-width-leg > gpurun_out/r3/ab_${v}_${rep}.json 2> gpurun_out/r3/ab_${v}_${rep}.err
  done
done
timeout 900 python -m pytest tests/test_ddp_gpu.py tests/test_kernels_gpu.py -k "ddp or two_ranks or bench_contract or flat_param or head_ops" -x -q > gpurun_out/r3/pytest_a.txt 2>&1; tail -15 gpurun_out/r3/pytest_a.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3/ab_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
