#!/bin/bash
# kernel traces of the cfg5 and cfg3 steps (one stream) -- where does the time outside the big GEMMs / attention go?
for w in cfg5 cfg3; do
  bash scratch/prof1.sh r04g_$w --workload $w --no-other-workloads --no-full-width-leg --no-launch-profile --data resident --serial
  head -45 gpurun_out/prof_r04g_$w/kernel_stats.csv | cut -c1-200
done
