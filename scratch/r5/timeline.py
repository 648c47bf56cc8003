"""Per-block phase sums of the persistent dK/dV kernel (side build with -DCHADA_PERS_TIMELINE, selected by CHADAVIT_HIP_LIB)."""
import os, sys, ctypes, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
from ctypes import c_int
import numpy as np
dev = torch.device('cuda:0'); bf = torch.bfloat16
D, H, p = 192, 2, 196
C = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = 1200000 // (1 + C * p)
rb = RaggedBatch([C] * B, p, dev)
qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf); do = torch.randn((rb.T, D), device=dev).to(bf)
o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
dq = torch.empty_like(qkv); dl = torch.empty((H, rb.T), device=dev)
def parts(pp):
    T, D3 = qkv.shape
    rc = ops.lib().chadavit_attn_bwd_parts(ops._ptr(qkv), ops._ptr(o), ops._ptr(do), ops._ptr(lse), ops._ptr(dq), ops._ptr(dl), ops._ptr(rb.cu_seqlens), ops._ptr(rb.work),
                                           c_int(rb.work.shape[0]), c_int(T), c_int(D3 // 3), c_int(H), c_int(pp), ops._stream())
    assert rc == 0
parts(3)
for _ in range(3): parts(4)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); parts(4); e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1)
raw = ctypes.CDLL(os.environ['CHADAVIT_HIP_LIB'])
buf = (ctypes.c_ulonglong * (512 * 8))()
raw.chadavit_debug_read_pers_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = raw.chadavit_debug_read_pers_timeline(buf, 512 * 8)
a = np.array(buf, dtype=np.float64).reshape(512, 8)
items, tiles, life = a[:, 4], a[:, 5], a[:, 6]
clk = life.mean() / us  # ticks per us
print(f"C={C} len={1 + C * p} kernel {us:.1f} us; per block: items {items.mean():.1f} (min {items.min():.0f} max {items.max():.0f}), tiles/item {tiles.sum() / items.sum():.2f}, life {life.mean():.0f} ticks (min {life.min():.0f} max {life.max():.0f}) -> {clk:.0f} ticks/us")
names = ["item top (cold start / wait / barrier)", "tiles before the last (+ hand-over)", "last tile", "prefetch + epilogue issue"]
tot = a[:, :4].sum()
for k in range(4):
    print(f"  {names[k]:42s} {a[:, k].sum() / items.sum() / clk:7.2f} us per item  ({100 * a[:, k].sum() / tot:5.1f} %)")
print(f"  sum of phases / life = {tot / life.sum():.3f}")
