#!/bin/bash
# (FETCH_SIZE and WRITE_SIZE in ONE pass crashed rocprofv3 and hung the call for 25 minutes: one derived counter per pass, every pass under `timeout`)
# PMC passes (cache hit rate, fetch size, busy counters) of the attention kernels at the Base shape (dh 384)
export TMPDIR=/tmp
for OP in attn_fwd_base attn_bwd_base; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$OP
mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -o t -- python3 $GRAFT_REPO_ROOT/scratch/one_op.py $OP $GRAFT_REPO_ROOT > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -o t -- python3 $GRAFT_REPO_ROOT/scratch/one_op.py $OP $GRAFT_REPO_ROOT > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p3 -o t -- python3 $GRAFT_REPO_ROOT/scratch/one_op.py $OP $GRAFT_REPO_ROOT > $OUT/p3.log 2>&1
find $OUT -name "*.db" -delete
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob,collections
for p in ("p1","p2","p3"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$OUT/"+p+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn" in r["Kernel_Name"]: acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print("$OP", k[0], k[1], sum(v)/len(v), len(v))
PY
done
