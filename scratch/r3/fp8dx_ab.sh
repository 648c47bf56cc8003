#!/bin/bash
# cfg5 with the FFN dX GEMMs in bf16 (product default before this change) vs on the MX-scaled MFMA, same box, alternating
mkdir -p gpurun_out/fp8dx
python -m pytest tests/test_kernels_gpu.py -x -q -k "mx8" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py -x -q -s -k "fp8_weight_path" 2>&1 | grep -E "passed|failed|fp8 step|Error|assert" | head -20
for i in 1 2; do
  for v in 0 1; do
    CHADAVIT_FP8_DX=$v python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-full-width-leg > gpurun_out/fp8dx/cfg5_dx$v.$i.json 2> gpurun_out/fp8dx/cfg5_dx$v.$i.err
    python - <<P
import json
d=json.loads(open("gpurun_out/fp8dx/cfg5_dx$v.$i.json").read().strip().splitlines()[-1])
print("fp8_dx=$v run $i:", d["value"], "images/s", d["ms_per_step"], "ms")
for k in d.get("launch_profile_top", [])[:14]:
    if "gemm_nt" in k["kernel"] or "mx8" in k["kernel"]: print("   ", k["kernel"], k["avg_us"], "x", k["launches_per_step"])
P
  done
done
