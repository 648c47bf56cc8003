#!/bin/bash
# LDS bank-conflict share of every kernel a script launches:  pmc_lds.sh <tag> <script.py>
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$1
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/$2 $GRAFT_REPO_ROOT > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/t_counter_collection.csv")):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "SQ_LDS_IDX_ACTIVE" in v and sum(v["SQ_LDS_IDX_ACTIVE"]) > 0:
        c = sum(v["SQ_LDS_BANK_CONFLICT"]) / sum(v["SQ_LDS_IDX_ACTIVE"])
        m = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(v["SQ_BUSY_CYCLES"]) * 32) if sum(v.get("SQ_BUSY_CYCLES", [0])) else 0
        print(f"{k:72s} launches {len(v['SQ_LDS_IDX_ACTIVE']):3d}  LDS conflict cycles {100 * c:5.1f} %  MFMA busy {100 * m:5.1f} %")
PY
