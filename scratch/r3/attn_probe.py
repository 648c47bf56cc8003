"""Timeline of the pipelined forward (attention_m32.hip built with -DCHADA_ATTN_PROBE): where a block's cycles go."""
import ctypes, sys, os, numpy as np, torch
sys.path.insert(0, '.')
from chadavit_amd._lib import lib
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
L = lib()
def fwd(qkv, rb, H, variant, out, lse):
    T, D3 = qkv.shape; D = D3 // 3
    rc = L.chadavit_attn_fwd_m32(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(lse.data_ptr()),
                                 ctypes.c_void_p(rb.cu_seqlens.data_ptr()), ctypes.c_void_p(rb.work.data_ptr()), ctypes.c_int(rb.n_work),
                                 ctypes.c_int(T), ctypes.c_int(D), ctypes.c_int(H), ctypes.c_int(variant),
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
for name, nch, p, D, H in (("tiny global 1024x589", [3] * 1024, 196, 192, 2), ("tiny local 4096x109", [3] * 4096, 36, 192, 2),
                           ("25x589: one block per CU at most", [3] * 25, 196, 192, 2), ("51x589: two blocks per CU at most", [3] * 51, 196, 192, 2),
                           ("12x1961 (10 channels), <= 1 block per CU", [10] * 8, 196, 192, 2)):
    rb = RaggedBatch(nch, p, dev)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o = torch.empty((rb.T, D), device=dev, dtype=bf); lse = torch.empty((H, rb.T), device=dev)
    for _ in range(3): fwd(qkv, rb, H, 4, o, lse)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fwd(qkv, rb, H, 4, o, lse); e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1)
    nb = min(16384, rb.n_work * H)
    buf = np.zeros((nb, 8), dtype=np.uint64)
    rc = L.chadavit_debug_attn_probe(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nb))
    assert rc == 0, rc
    b = buf[buf[:, 4] > 0].astype(np.float64)
    tot, wait, pro, epi, nkt = b[:, 0], b[:, 1], b[:, 2], b[:, 3], b[:, 4]
    span = (buf[:, 6].max() - buf[buf[:, 5] > 0][:, 5].min())
    print(f"{name}: launch {us:.1f} us (with probes); blocks {len(b)}; tiles per block {nkt.mean():.1f}")
    print(f"   per block (shader cycles): total {tot.mean():.0f}  = prologue (decode, Q, tile 0 landed, S'(0), max, first half exp) {pro.mean():.0f}"
          f" + steady state {(tot - pro - epi).mean():.0f} [{((tot - pro - epi) / np.maximum(nkt - 1, 1)).mean():.0f} per iteration, of which waiting at the barrier {(wait / np.maximum(nkt - 1, 1)).mean():.0f}]"
          f" + epilogue (last tile, range check, stores) {epi.mean():.0f}")
    print(f"   whole launch in shader cycles (first start -> last end, counters of all XCDs): {span:.0f}  -> {span / us / 1e3:.2f} GHz if comparable")
    for q in (10, 50, 90):
        print(f"   percentile {q}: total {np.percentile(tot, q):.0f} wait/iter {np.percentile(wait / np.maximum(nkt - 1, 1), q):.0f}")
