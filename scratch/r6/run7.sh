#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_run7; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "attention" > $O/kernels.log 2>&1; tail -4 $O/kernels.log
python -m pytest tests/test_model_gpu.py -q -x -m gpu -k "training_step_vs_golden or backbone_vs_golden or trajectory" > $O/model.log 2>&1; tail -4 $O/model.log
