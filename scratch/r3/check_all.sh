#!/bin/bash
# full GPU suite + smoke + the four workloads on one box (no profiler): regression check after kernel changes
O=gpurun_out/r3/check_${1:-a}
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
BENCH_TOP=16 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for w in cfg3 cfg5 cfg2-mixed; do BENCH_TOP=16 timeout 600 python bench.py --workload $w --no-cpu-baseline --no-other-workloads --steps 6 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), {k:(v.get("images_per_s") if isinstance(v,dict) else v) for k,v in d["config"].get("other_workloads",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
