#!/bin/bash
mkdir -p gpurun_out/r4fm
for w in cfg3 cfg2-mixed; do
  python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-launch-profile --no-full-width-leg > gpurun_out/r4fm/$w.json 2> gpurun_out/r4fm/$w.err
  python - $w <<'PY'
import json, sys
w = sys.argv[1]
d = json.loads(open(f"gpurun_out/r4fm/{w}.json").read().strip().splitlines()[-1])
print(w, d["value"], json.dumps(d["config"].get("data_path"), indent=0))
PY
  tail -2 gpurun_out/r4fm/$w.err
done
