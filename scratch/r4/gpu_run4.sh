#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tee $O/pytest_gpu.log | tail -6
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-other-workloads --no-cpu-baseline --data resident > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 900 python bench.py --workload cfg3 --steps 5 --warmup 2 --no-other-workloads --no-cpu-baseline --data resident > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<'PY'
import json
for w in ("cfg5","cfg3"):
    d=json.load(open(f"gpurun_out/r4g/bench_{w}.json"))
    print(w, d["value"], d["ms_per_step"])
    for e in d["launch_profile_top"][:14]:
        print(f"   {e['kernel'][:64]:64s} {e['ms_per_step']:7.3f} ms {e['avg_us']:8.1f} us x {e['launches_per_step']}")
PY
