#!/bin/bash
# PMC passes of the two augmentation kernels (separate passes; kernel-trace only)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_aug
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -o t -- python3 $GRAFT_REPO_ROOT/scratch/r4/aug_only.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -o t -- python3 $GRAFT_REPO_ROOT/scratch/r4/aug_only.py > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/p3 -o t -- python3 $GRAFT_REPO_ROOT/scratch/r4/aug_only.py > $OUT/p3.log 2>&1
find $OUT -name "*.db" -delete
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/pmc_aug"
for p in ("p1", "p2", "p3"):
    f = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "crop_resize" not in k and "blur_finish" not in k:
            continue
        name = "crop_resize" if "crop_resize" in k else "blur_finish"
        key = (name, r["Grid_Size"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(key, r["Counter_Name"])] += 1
    for key in sorted(acc):
        print(p, key, {c: round(v / cnt[(key, c)], 1) for c, v in acc[key].items()})
PY
