#!/bin/bash
mkdir -p gpurun_out/r4u8
python -m pytest tests/test_augment_gpu.py -x -q -m gpu 2>&1 | tail -4
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-workloads --no-launch-profile --no-full-width-leg > gpurun_out/r4u8/bench.json 2> gpurun_out/r4u8/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4u8/bench.json").read().strip().splitlines()[-1])
print(d["value"], json.dumps(d["config"]["data_path"], indent=0))
PY
tail -3 gpurun_out/r4u8/bench.err
