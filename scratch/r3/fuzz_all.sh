#!/bin/bash
# the poisoned-allocator determinism fuzz over the step goldens / dispatches
run() { echo "== $*"; env "$@" timeout 300 python scratch/r3/poison_fuzz.py $NAME 2>&1 | grep '^run' | cut -c1-230; }
NAME=step_small_mixed run FUSED0=1
NAME=step_small_mixed run X=1
NAME=step_base_c10 run FP8=1
NAME=step_base_c10 run X=1
NAME=step_base_c10 run FP8=1 OVERLAP=1
NAME=step_tiny_multicrop run FUSED0=1
NAME=step_tiny_multicrop run OVERLAP=1
NAME=step_tiny_bn_head run X=1
NAME=step_tiny_fused_rows run OVERLAP=1
