#!/bin/bash
# same-box A/B of whole-step throughput between the product library and side builds: scratch/r5/lib_ab_bench.sh "<tags>" "<workloads>" [rounds]
TAGS=${1:-"product"}; WLS=${2:-"cfg2"}; ROUNDS=${3:-2}
O=gpurun_out/lib_ab; mkdir -p $O
for r in $(seq 1 $ROUNDS); do for t in $TAGS; do for w in $WLS; do
  if [ $t = product ]; then unset CHADAVIT_HIP_LIB CHADAVIT_ALLOW_FOREIGN_LIB; else export CHADAVIT_HIP_LIB=$PWD/scratch/sidebuild/$t/libchadavit_hip_$t.so CHADAVIT_ALLOW_FOREIGN_LIB=1; fi
  v=$(timeout 300 python bench.py --workload $w --steps 6 --warmup 2 --no-other-workloads --no-cpu-baseline --data resident --no-full-width-leg --no-launch-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'])")
  echo "round $r  $t  $w  $v" | tee -a $O/ab.log
done; done; done
