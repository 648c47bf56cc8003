"""dK/dV (and dQ, forward) time against sequence length at (nearly) constant tokens, both scheduling modes: per-item fixed cost vs per-tile cost."""
import os, sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
from ctypes import c_int
dev = torch.device('cuda:0'); bf = torch.bfloat16
def t(fn, reps=20, rounds=3):
    for _ in range(3): fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]
def parts(qkv, o, do, lse, cu, work, H, p, dq, dl):
    T, D3 = qkv.shape
    rc = ops.lib().chadavit_attn_bwd_parts(ops._ptr(qkv), ops._ptr(o), ops._ptr(do), ops._ptr(lse), ops._ptr(dq), ops._ptr(dl), ops._ptr(cu), ops._ptr(work),
                                           c_int(work.shape[0]), c_int(T), c_int(D3 // 3), c_int(H), c_int(p), ops._stream())
    assert rc == 0, rc
D, H = 192, 2
TT = 1200000
for p in (196,):
    for C in (1, 2, 3, 4, 6, 10):
        n = 1 + C * p
        B = TT // n
        rb = RaggedBatch([C] * B, p, dev)
        qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
        do = torch.randn((rb.T, D), device=dev).to(bf)
        os.environ['CHADAVIT_ATTN_PERSISTENT'] = '0'
        o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
        dq = torch.empty_like(qkv); dl = torch.empty((H, rb.T), device=dev)
        items = sum((n + 127) // 128 for _ in range(B)) * H
        line = f"len {n:5d} x {B:5d} seqs, items {items:6d}, q-tiles/item {(n + 63) // 64:3d}:"
        for m in (0, 1):
            os.environ['CHADAVIT_ATTN_PERSISTENT'] = str(m)
            tf = t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=o, lse=lse))
            tq = t(lambda: parts(qkv, o, do, lse, rb.cu_seqlens, rb.work, H, 3, dq, dl))
            tk = t(lambda: parts(qkv, o, do, lse, rb.cu_seqlens, rb.work, H, 4, dq, dl))
            line += f" | mode {m}: fwd {tf:7.1f} dq {tq:7.1f} dkv {tk:7.1f} us; dkv per item-slot {tk * 512 / items:6.2f} us"
        print(line, flush=True)
