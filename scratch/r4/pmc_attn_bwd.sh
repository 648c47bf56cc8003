#!/bin/bash
# PMC passes (scratch/pmc.sh) on the attention backward pair at the bench's shape, both MFMA formulations
export ONE_OP_T=603136
bash scratch/pmc.sh attn_bwd; mv gpurun_out/pmc_attn_bwd gpurun_out/r4d_pmc_attn_bwd_m32
export CHADAVIT_ATTN_BWD_M32=-1
bash scratch/pmc.sh attn_bwd; mv gpurun_out/pmc_attn_bwd gpurun_out/r4d_pmc_attn_bwd_m16
python3 scratch/pmc_print.py gpurun_out/r4d_pmc_attn_bwd_m32 > gpurun_out/r4d_pmc_m32.txt
python3 scratch/pmc_print.py gpurun_out/r4d_pmc_attn_bwd_m16 > gpurun_out/r4d_pmc_m16.txt
cat gpurun_out/r4d_pmc_m32.txt gpurun_out/r4d_pmc_m16.txt
