#!/bin/bash
# data-path probe: plain run of all modes, then kernel traces of the pipeline alone and of the fed step
mkdir -p gpurun_out/r4dp
python scratch/r4/data_path_probe.py --steps 10 > gpurun_out/r4dp/modes.log 2>&1
tail -3 gpurun_out/r4dp/modes.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4dp/alone -o alone -- python3 $GRAFT_REPO_ROOT/scratch/r4/data_path_probe.py --steps 4 --modes alone > $GRAFT_REPO_ROOT/gpurun_out/r4dp/alone.log 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4dp/fed -o fed -- python3 $GRAFT_REPO_ROOT/scratch/r4/data_path_probe.py --steps 4 --modes producer > $GRAFT_REPO_ROOT/gpurun_out/r4dp/fed.log 2>&1
cd $GRAFT_REPO_ROOT/gpurun_out/r4dp
for d in alone fed; do f=$(find $d -name "*kernel_stats.csv" | head -1); echo "== $d $f"; grep -i "crop_resize\|blur_finish" $f; find $d -name "*kernel_trace.csv" -size +30M -delete; done
