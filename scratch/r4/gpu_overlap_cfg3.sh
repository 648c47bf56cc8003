#!/bin/bash
mkdir -p gpurun_out/r4ov
for rep in 1 2; do for mode in "" "--overlap"; do
  python bench.py --workload cfg3 $mode --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-launch-profile --no-full-width-leg --data resident > gpurun_out/r4ov/o.json 2> gpurun_out/r4ov/o.err
  python - "$mode" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4ov/o.json").read().strip().splitlines()[-1])
print("cfg3", sys.argv[1] or "default", d["value"], d["ms_per_step"])
PY
done; done
