"""print per-kernel averaged counters from a scratch/pmc.sh output dir"""
import csv, os, sys, re
src = sys.argv[1]
agg = {}
for p in ("p1", "p2", "p3"):
    fn = os.path.join(src, p, "t_counter_collection.csv")
    if not os.path.exists(fn): continue
    with open(fn) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "at::native" in n or "elementwise" in n: continue
            n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][-60:]
            d = agg.setdefault(n, {}).setdefault(r["Counter_Name"], [0, 0.0])
            d[0] += 1; d[1] += float(r["Counter_Value"])
            agg[n]["_vgpr"] = [1, float(r["VGPR_Count"])]; agg[n]["_grid"] = [1, float(r["Grid_Size"])]
for n, c in agg.items():
    v = {k: s / cnt for k, (cnt, s) in c.items()}
    print(n)
    busy = v.get("SQ_BUSY_CYCLES", 0)
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print("  grid %d vgpr %d | busy_cycles/SE? %.0f wave_cycles %.3g" % (v["_grid"], v["_vgpr"], busy, wc))
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM"):
        if k in v: print("  %-22s %.3g  (%.1f%% of wave cycles)" % (k, v[k], 100 * v[k] / wc))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v: print("  MFMA busy frac %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (busy * 32)))
    if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"): print("  LDS bank conflict frac %.3f  (conflict %.3g / active %.3g)" % (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], v["SQ_LDS_BANK_CONFLICT"], v["SQ_LDS_IDX_ACTIVE"]))
    if "FETCH_SIZE" in v: print("  FETCH %.1f MB (x2 corr %.1f)  WRITE %.1f MB" % (v["FETCH_SIZE"] / 1024, v["FETCH_SIZE"] / 512, v.get("WRITE_SIZE", 0) / 1024))
    for k in ("SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_BF16"):
        if k in v: print("  %-22s %.4g" % (k, v[k]))
