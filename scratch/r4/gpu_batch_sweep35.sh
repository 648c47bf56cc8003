#!/bin/bash
# per-GPU batch sweep of the cfg5 / cfg3 workloads (BASELINE.json names no batch for configs[2] / [4]; configs[3] = 128 per GPU)
mkdir -p gpurun_out/r4bs
for spec in "cfg5 32" "cfg5 64" "cfg5 128" "cfg3 128" "cfg3 256" "cfg3 512"; do
  set -- $spec
  python bench.py --workload $1 --batch $2 --steps 4 --warmup 2 --no-cpu-baseline --no-other-workloads --no-launch-profile --no-full-width-leg --data resident > gpurun_out/r4bs/$1_$2.json 2> gpurun_out/r4bs/$1_$2.err
  python - "$1" "$2" <<'PY'
import json, sys
w, b = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/r4bs/{w}_{b}.json").read().strip().splitlines()[-1])
    print(w, b, d["value"], "images/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print(w, b, "failed", e)
PY
done
