"""attention_bwd_m32.hip (32x32x16 backward pair) vs fp32 torch autograd and vs the 16x16x32 pair (CHADAVIT_ATTN_BWD_M32=-1 in a child
process): correctness at tile boundaries + timing at the bench's shapes.   python scratch/r4/attn_bwd_m32.py [time-only]"""
import os, subprocess, sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
tag = "32x32x16" if os.environ.get("CHADAVIT_ATTN_BWD_M32", "-1") != "-1" else "16x16x32"

def ref(qkv, dout, cu, H):
    T, D3 = qkv.shape; D = D3 // 3; dh = D // H
    x = qkv.float().clone().requires_grad_(True)
    q, k, v = x.split(D, dim=1)
    outs = []
    for i in range(len(cu) - 1):
        a, b = cu[i], cu[i + 1]
        oh = []
        for h in range(H):
            s = (q[a:b, h*dh:(h+1)*dh] @ k[a:b, h*dh:(h+1)*dh].T) / dh ** 0.5
            oh.append(torch.softmax(s, dim=1) @ v[a:b, h*dh:(h+1)*dh])
        outs.append(torch.cat(oh, dim=1))
    out = torch.cat(outs)
    (out * dout.float()).sum().backward()
    return x.grad

def check(name, nch, p, D, H):
    rb = RaggedBatch(nch, p, dev)
    torch.manual_seed(1)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    dout = torch.randn((rb.T, D), device=dev).to(bf)
    out, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    g_ref = ref(qkv, dout, rb.host_cu_seqlens, H)
    worst = 0
    for side in (None, torch.cuda.Stream()):
        dqkv = ops.attn_bwd(qkv, out, dout, lse, rb.cu_seqlens, rb.work, H, side=side)
        torch.cuda.synchronize()
        for nm, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
            a, b = dqkv[:, sl].float(), g_ref[:, sl]
            rel = ((a - b).norm() / b.norm()).item(); mx = (a - b).abs().max().item() / b.abs().max().item()
            ok = rel < 1.5e-2 and mx < 3e-2 and torch.isfinite(a).all().item()
            worst = max(worst, rel)
            if not ok:
                print(f"{tag} {name:24s} {nm} side={side is not None}: rel {rel:.2e} max {mx:.2e} FAIL", flush=True)
    print(f"{tag} {name:24s} worst rel {worst:.2e}", flush=True)

def t(fn, reps=10, rounds=3):
    for _ in range(3): fn()
    res = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(1e3 * e0.elapsed_time(e1) / reps)
    return sorted(res)[len(res) // 2]

if "time-only" not in sys.argv:
    check("tiny 3ch x4 (589)", [3] * 4, 196, 192, 2)
    check("tiny mixed", [1, 2, 5, 10, 3], 196, 192, 2)
    check("tiny local (109)", [3] * 6, 36, 192, 2)
    check("boundaries", [1, 14, 15, 16, 30, 31, 32, 33, 62, 63, 64, 65, 95, 96, 97, 126, 127, 128, 129, 191, 192, 256, 257], 1, 192, 2)
    check("small mixed (dh 192)", [1, 2, 5, 10, 3], 196, 384, 2)
    check("small boundaries", [1, 14, 15, 16, 30, 31, 32, 33, 62, 63, 64, 65, 95, 96, 97, 126, 127, 128, 129, 191, 192, 256], 1, 384, 2)
for name, nch, p, D, H in (("tiny global 1024x589", [3] * 1024, 196, 192, 2), ("tiny local 4096x109", [3] * 4096, 36, 192, 2),
                           ("tiny mixed 512", [1,2,3,4,5,6,7,8,9,10] * 51, 196, 192, 2), ("small mixed 250", [1,2,3,4,5,6,7,8,9,10] * 25, 196, 384, 2)):
    rb = RaggedBatch(nch, p, dev)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf); dout = torch.randn((rb.T, D), device=dev).to(bf)
    out, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    dqkv = torch.empty_like(qkv); delta = torch.empty((H, rb.T), device=dev)
    fl = 10.0 * sum(n * n for n in rb.lens) * D
    L = ops.lib()
    import ctypes
    def parts(pp):
        rc = L.chadavit_attn_bwd_parts(*[ctypes.c_void_p(x.data_ptr()) for x in (qkv, out, dout, lse, dqkv, delta, rb.cu_seqlens, rb.work)],
                                       ctypes.c_int(rb.n_work), ctypes.c_int(rb.T), ctypes.c_int(D), ctypes.c_int(H), ctypes.c_int(pp),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    for rep in range(2):
        us = t(lambda: parts(7)); udq = t(lambda: parts(3)); udkv = t(lambda: parts(4))
        print(f"{tag} {name:22s} T={rb.T}: pair {us:8.1f} us (dQ+delta {udq:7.1f}, dK/dV {udkv:7.1f})  {fl/us/1e6:6.0f} TF/s = {fl/us/1e6/2500:.3f} of 2.5 PF", flush=True)
if tag == "16x16x32" and "no-child" not in sys.argv:   # the default dispatch first, then the 32x32x16 pair in a child process
    subprocess.run([sys.executable] + sys.argv + ["no-child"], env=dict(os.environ, CHADAVIT_ATTN_BWD_M32="1"))
