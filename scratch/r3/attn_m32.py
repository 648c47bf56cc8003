"""attention_m32.hip forward variants vs fp32 torch and vs the 16x16x32 kernel: correctness (incl. the out-of-range re-run) + timing."""
import ctypes, sys, os, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd._lib import lib
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
L = lib()

def fwd(qkv, rb, H, variant, out=None, lse=None):
    T, D3 = qkv.shape; D = D3 // 3
    out = torch.empty((T, D), device=dev, dtype=bf) if out is None else out
    lse = torch.empty((H, T), device=dev, dtype=torch.float32) if lse is None else lse
    if variant == 0:   # whatever chadavit_attn_fwd dispatches to (CHADAVIT_ATTN_FWD_M32=-1: round 2's 16x16x32 kernels)
        return ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=out, lse=lse)
    rc = L.chadavit_attn_fwd_m32(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(lse.data_ptr()),
                                 ctypes.c_void_p(rb.cu_seqlens.data_ptr()), ctypes.c_void_p(rb.work.data_ptr()), ctypes.c_int(rb.n_work),
                                 ctypes.c_int(T), ctypes.c_int(D), ctypes.c_int(H), ctypes.c_int(variant),
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return out, lse

def ref(qkv, cu, H):
    T, D3 = qkv.shape; D = D3 // 3; dh = D // H
    q, k, v = qkv.float().split(D, dim=1)
    out = torch.empty((T, D), device=dev); lse = torch.empty((H, T), device=dev)
    for i in range(len(cu) - 1):
        a, b = cu[i], cu[i + 1]
        for h in range(H):
            s = (q[a:b, h*dh:(h+1)*dh] @ k[a:b, h*dh:(h+1)*dh].T) / dh ** 0.5
            lse[h, a:b] = torch.logsumexp(s, dim=1)
            out[a:b, h*dh:(h+1)*dh] = torch.softmax(s, dim=1) @ v[a:b, h*dh:(h+1)*dh]
    return out, lse

def check(name, nch, p, D, H, variants, spike=None):
    rb = RaggedBatch(nch, p, dev)
    torch.manual_seed(1)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    qkv[min(rb.T - 1, 70), :D] *= 6.0
    if spike is not None:
        qrow, krow, val = spike
        dh = D // H
        qkv[qrow, :dh] = val; qkv[krow, D:D + dh] = val
    o_ref, l_ref = ref(qkv, rb.host_cu_seqlens, H)
    for v in variants:
        o, l = fwd(qkv, rb, H, v)
        torch.cuda.synchronize()
        eo = (o.float() - o_ref).abs().max().item(); el = (l - l_ref).abs().max().item()
        ok = eo < 2e-2 * max(1.0, o_ref.abs().max().item()) and el < 2e-2 and torch.isfinite(o.float()).all().item()
        print(f"{name:28s} variant {v}: max|dO| {eo:.2e} max|dLSE| {el:.2e} {'OK' if ok else 'FAIL'}", flush=True)

def t(fn, reps=20, rounds=3):
    for _ in range(3): fn()
    res = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(1e3 * e0.elapsed_time(e1) / reps)
    return sorted(res)[len(res) // 2]

import os
vs96 = [0] if os.environ.get("CHADAVIT_ATTN_FWD_M32") == "-1" else [0, 8]
check("tiny 3ch x4 (589)", [3] * 4, 196, 192, 2, vs96)
check("tiny mixed", [1, 2, 5, 10, 3], 196, 192, 2, vs96)
check("tiny local (109)", [3] * 6, 36, 192, 2, vs96)
check("boundaries", [1, 14, 15, 16, 30, 31, 32, 62, 63, 64, 95, 96, 126, 127, 128, 191, 192, 256], 1, 192, 2, vs96)
check("out-of-range re-run", [3] * 3, 196, 192, 2, vs96, spike=(300, 500, 2.5))
check("out-of-range, last tile", [3] * 3, 196, 192, 2, vs96, spike=(10, 588, 2.5))
check("small mixed (dh 192)", [1, 2, 5, 10, 3], 196, 384, 2, [0])
check("small boundaries", [1, 14, 15, 16, 30, 31, 32, 62, 63, 64, 95, 96, 126, 127, 128, 191, 192, 256], 1, 384, 2, [0])
check("small re-run", [3] * 3, 196, 384, 2, [0, 2, 5], spike=(300, 500, 2.0))
for name, nch, p, D, H, vs in (("tiny global 1024x589", [3] * 1024, 196, 192, 2, vs96), ("tiny local 4096x109", [3] * 4096, 36, 192, 2, vs96),
                               ("tiny mixed 512", [1,2,3,4,5,6,7,8,9,10] * 51, 196, 192, 2, vs96),
                               ("small mixed 120", [1,2,3,4,5,6,7,8,9,10] * 24, 196, 384, 2, [0])):
    rb = RaggedBatch(nch, p, dev)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o = torch.empty((rb.T, D), device=dev, dtype=bf); lse = torch.empty((H, rb.T), device=dev)
    fl = 4.0 * sum(n * n for n in rb.lens) * D
    for rep in range(2):
        for v in vs:
            us = t(lambda: fwd(qkv, rb, H, v, out=o, lse=lse))
            print(f"{name:24s} T={rb.T} variant {v}: {us:8.1f} us  {fl/us/1e6:7.0f} TF/s  ({fl/us/1e6/2500:.3f} of 2.5 PF)", flush=True)
