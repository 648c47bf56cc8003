"""Persistent vs one-item-per-block attention kernels on one box: bit-equality of the outputs and HIP-event timing, alternating.
usage: python scratch/r5/attn_pers_ab.py [quick]"""
import os, sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
quick = 'quick' in sys.argv
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
def mode(m): os.environ['CHADAVIT_ATTN_PERSISTENT'] = str(m)
from ctypes import c_int
def parts(qkv, o, do, lse, cu, work, H, p, dq, dl):
    T, D3 = qkv.shape
    rc = ops.lib().chadavit_attn_bwd_parts(ops._ptr(qkv), ops._ptr(o), ops._ptr(do), ops._ptr(lse), ops._ptr(dq), ops._ptr(dl), ops._ptr(cu), ops._ptr(work),
                                           c_int(work.shape[0]), c_int(T), c_int(D3 // 3), c_int(H), c_int(p), ops._stream())
    assert rc == 0, rc
import random
random.seed(1)
cases = [("tiny global B=1024", [3] * 2048, 196, 192, 2), ("tiny local B=1024", [3] * 8192, 36, 192, 2),
         ("tiny mixed 256", [random.randint(1, 10) for _ in range(512)], 196, 192, 2),
         ("small mixed 128", [random.randint(1, 10) for _ in range(256)], 196, 384, 2),
         ("base 10ch", [10] * 64, 196, 768, 2), ("tiny 1ch short", [1] * 640 + [2] * 3, 36, 192, 2)]
if quick: cases = [("tiny global B=64", [3] * 128, 196, 192, 2), ("tiny mixed 32", [random.randint(1, 10) for _ in range(64)], 196, 192, 2),
                   ("tiny 1ch short", [1] * 64 + [2] * 3, 36, 192, 2), ("small mixed 16", [random.randint(1, 10) for _ in range(32)], 196, 384, 2)]
for name, nch, p, D, H in cases:
    rb = RaggedBatch(nch, p, dev)
    torch.manual_seed(0)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    do = torch.randn((rb.T, D), device=dev).to(bf)
    res = {}
    for m in (0, 1):
        mode(m)
        o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
        dq = torch.full_like(qkv, float('nan')); dl = torch.empty((H, rb.T), device=dev)
        for rep in range(3):  # repeated launches: the queue slot resets itself
            dq.fill_(float('nan'))
            ops.attn_bwd(qkv, o, do, lse, rb.cu_seqlens, rb.work, H, dqkv=dq, delta=dl)
            torch.cuda.synchronize()
            res[(m, rep)] = (o.clone(), lse.clone(), dq.clone(), dl.clone())
    ok = True
    for rep in range(3):
        for k, nm in enumerate(("out", "lse", "dqkv", "delta")):
            a, b = res[(0, 0)][k], res[(1, rep)][k]
            same = torch.equal(a.view(torch.int16) if a.dtype == bf else a.view(torch.int32), b.view(torch.int16) if b.dtype == bf else b.view(torch.int32))
            if not same:
                ok = False
                bad = (a.float() != b.float()) | (a.float().isnan() != b.float().isnan())
                print(f"  MISMATCH {name} rep {rep} {nm}: {int(bad.sum())} of {bad.numel()} elements, nan in new {int(b.float().isnan().sum())}", flush=True)
    line = f"{name}: T={rb.T} bit-identical={ok}"
    if not quick or True:
        o, lse = res[(0, 0)][0], res[(0, 0)][1]
        dq = torch.empty_like(qkv); dl = torch.empty((H, rb.T), device=dev)
        tm = {}
        for rnd in range(2):
            for m in (0, 1):
                mode(m)
                tf = t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=o, lse=lse))
                tq = t(lambda: parts(qkv, o, do, lse, rb.cu_seqlens, rb.work, H, 3, dq, dl))
                tk = t(lambda: parts(qkv, o, do, lse, rb.cu_seqlens, rb.work, H, 4, dq, dl))
                tb = t(lambda: ops.attn_bwd(qkv, o, do, lse, rb.cu_seqlens, rb.work, H, dqkv=dq, delta=dl))
                tm.setdefault(m, []).append((tf, tq, tk, tb))
        for m in (0, 1):
            best = [min(x[i] for x in tm[m]) for i in range(4)]
            line += f" | mode {m}: fwd {best[0]:.1f} dq {best[1]:.1f} dkv {best[2]:.1f} bwd {best[3]:.1f} us"
    print(line, flush=True)
