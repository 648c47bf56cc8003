#!/bin/bash
cd $GRAFT_REPO_ROOT
for mode in --serial --overlap --serial --overlap; do
  echo "== cfg2 $mode"; python bench.py --gpus 1 --steps 10 --warmup 3 --no-other-workloads --no-cpu-baseline --no-full-width-leg --no-launch-profile --data resident $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
for wl in cfg3 cfg2-mixed; do for mode in --serial --overlap; do
  echo "== $wl $mode"; python bench.py --gpus 1 --steps 10 --warmup 3 --workload $wl --no-other-workloads --no-cpu-baseline --no-full-width-leg --no-launch-profile --data resident $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
