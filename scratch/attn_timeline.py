"""Per-block anatomy of the attention forward (debug lib built with -DCHADA_ATTN_TIMELINE): s_memtime stamps at
0 entry, 1 work item + cu_seqlens decoded, 2 Q rows + K/V tile 0 landed (first barrier), 3 second barrier (tile 0 done),
4 barrier in front of the last tile, 5 last tile done, 6 output stores drained; slot 7 = HW_ID | XCC_ID << 32."""
import os, sys, ctypes, collections, numpy as np, torch
os.environ["CHADAVIT_HIP_LIB"] = "scratch/ab/lib_attn_tl.so"
sys.path.insert(0, ".")
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
from chadavit_amd._lib import lib
dev = torch.device("cuda:0"); bf = torch.bfloat16
D, H = 192, 2
buf = np.zeros(16384 * 8, dtype=np.uint64)
for C, B, p, dyn in ((3, 1024, 196, 0), (3, 1024, 196, 16384), (3, 1024, 196, 65536), (3, 4096, 36, 0)):
    os.environ["CHADA_ATTN_DYNLDS"] = str(dyn)
    print("extra dynamic LDS", dyn)
    rb = RaggedBatch([C] * B, p, dev)
    n = 1 + C * p
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    for _ in range(3): ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=o, lse=lse)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=o, lse=lse)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1)
    rc = lib().chadavit_debug_attn_timeline(ctypes.c_void_p(buf.ctypes.data))
    t = buf.reshape(16384, 8).astype(np.int64)
    nblk = min(16384, B * H * ((n + 127) // 128))
    t = t[:nblk]
    ok = t[:, 6] > 0
    t = t[ok]
    xcc = t[:, 7] >> 32
    # the counters of the 8 XCDs are not aligned with each other: ticks per us from the per-XCD spans
    clk = float(np.median([(t[xcc == x, 6].max() - t[xcc == x, 0].min()) for x in np.unique(xcc)])) / us
    nkt = (n + 63) // 64
    d = lambda a, b_: (t[:, a] - t[:, b_]) / clk
    print(f"N={n} B={B}: launch {us:.1f} us, rc={rc}, {len(t)} blocks stamped, s_memtime {clk:.1f} ticks/us, key tiles {nkt}")
    names = ["decode (work, cu)", "Q + tile 0 landed", "tile 0", f"tiles 1..{nkt - 2} (each)", "last tile (masked)", "normalise + stores drained"]
    vals = [d(1, 0), d(2, 1), d(3, 2), d(4, 3) / max(1, nkt - 2), d(5, 4), d(6, 5)]
    for nm, v in zip(names, vals):
        print(f"   {nm:28s} mean {v.mean():7.2f} us  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}")
    tot = d(6, 0)
    print(f"   block total {tot.mean():.2f} us; sum over blocks / launch time = {tot.sum() / us:.1f} blocks in flight (of {256 * (3 if dyn == 0 else 2 if dyn < 60000 else 1)} slots)")
    # per-CU slot gaps: blocks that ran on the same CU, ordered by entry
    cu = ((t[:, 7] >> 32) << 16) | ((t[:, 7] & 0xFFFF) >> 8)
    gaps = []; conc = []
    for k in np.unique(cu):
        sel = t[cu == k]
        sel = sel[np.argsort(sel[:, 0])]
        ends = np.sort(sel[:, 6])
        # time between a block ending on this CU and the next block entering (the k-th end frees a slot for entry k+resident)
        res = 3 if dyn == 0 else 2 if dyn < 60000 else 1
        if len(sel) > res:
            g_ = (sel[res:, 0] - ends[:len(sel) - res]) / clk
            gaps.append(g_)
    if gaps:
        g_ = np.concatenate(gaps)
        print(f"   {len(np.unique(cu))} CUs seen; slot turnover (block end -> next block's first instruction) mean {g_.mean():.2f} us, p10 {np.percentile(g_, 10):.2f}, p90 {np.percentile(g_, 90):.2f}")
