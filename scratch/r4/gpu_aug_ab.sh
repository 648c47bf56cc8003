#!/bin/bash
mkdir -p gpurun_out/r4aug
python scratch/r4/aug_kernels_ab.py > gpurun_out/r4aug/ab.log 2>&1; tail -40 gpurun_out/r4aug/ab.log
python -m pytest tests/test_augment_gpu.py -x -q -m gpu > gpurun_out/r4aug/pytest.log 2>&1; tail -5 gpurun_out/r4aug/pytest.log
python scratch/r4/data_path_probe.py --steps 10 --modes resident,producer,consumer,resident > gpurun_out/r4aug/modes.log 2>&1; tail -2 gpurun_out/r4aug/modes.log
