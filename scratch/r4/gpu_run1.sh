#!/bin/bash
# round 4, first GPU call: (1) the whole GPU suite with the new per-pass output / gradient-spread assertions; (2) the same step tests
# against a build with round 3's QKV-postlogue wait bug put back (scratch/r4/postlogue_bug_build.py): they must go RED.
O=gpurun_out/r4a; mkdir -p $O
python -m pytest tests -m gpu -x -q -s 2>&1 | tee $O/pytest_gpu.log | tail -40
echo "=== bug build ===" 
CHADAVIT_ALLOW_FOREIGN_LIB=1 CHADAVIT_HIP_LIB=$PWD/scratch/sidebuild/postlogue_bug/libchadavit_hip_postlogue_bug.so python -m pytest tests/test_model_gpu.py -m gpu -q -s \
   -k "test_bench_scale_replicated_batch_vs_golden or test_training_step_vs_golden_and_oracle" 2>&1 | tee $O/pytest_bug_build.log | tail -60
