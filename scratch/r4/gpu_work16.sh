#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O
timeout 600 python scratch/r4/attn_bwd_m32.py no-child 2>&1 | grep -E "us \(|rel" | tee $O/bwd.log
timeout 600 python scratch/r3/attn_m32.py 2>&1 | grep -E "variant 0" | tee $O/fwd.log
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention or attn" 2>&1 | tail -3
