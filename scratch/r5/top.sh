#!/bin/bash
# usage: top.sh <workload> [n]  -- the launch profile of one workload (ms per step by entry point)
BENCH_TOP=${2:-30} python bench.py --workload $1 --steps 4 --warmup 2 --no-other-workloads --no-cpu-baseline --data resident --no-full-width-leg 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], json.dumps(d.get('power')))
tot=d['roofline']['instrumented_ms_per_step']
for r in d['launch_profile_top']: print(f\"{r['ms_per_step']:8.3f} ms {100*r['ms_per_step']/tot:5.1f}%  {r['avg_us']:9.1f} us x {r['launches_per_step']:3d}  {r['kernel']}\")
print('instrumented', tot)"
