"""Run ONE hot entry point a few times (for rocprofv3 --pmc passes)."""
import sys, torch
sys.path.insert(0, sys.argv[2] if len(sys.argv) > 2 else '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
import os
T = int(os.environ.get("ONE_OP_T", "150784"))
name = sys.argv[1]
if os.environ.get("ONE_OP_ZERO"):   # all-zero operands: the same instructions at full clock (what the power cap costs)
    _randn = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **k)
shapes = {"qkv": (T, 576, 192, 0), "outproj": (T, 192, 192, 3), "ffn1": (T, 2048, 192, 1), "ffn2": (T, 192, 2048, 3),
          "dH": (T, 2048, 192, 4), "dx1": (T, 192, 2048, 3), "dh": (T, 192, 576, 0)}
if name in shapes:
    M, N, K, epi = shapes[name]
    x = torch.randn((M, K), device=dev).to(bf); w = (torch.randn((N, K), device=dev) / K ** .5).to(bf)
    bias = torch.zeros(N, device=dev); aux = torch.randn((M, N), device=dev).to(bf) if epi in (3, 4) else None
    o = torch.empty((M, N), device=dev, dtype=bf)
    fn = lambda: ops.gemm_nt(x, w, out=o, bias=bias, epilogue=epi, aux=aux)
elif name in ("dW1", "dW2"):
    I, J = (2048, 192) if name == "dW1" else (192, 2048)
    a = torch.randn((T, I), device=dev).to(bf); b = torch.randn((T, J), device=dev).to(bf)
    c = torch.empty((I, J), device=dev); cs = torch.empty(I, device=dev); ws = torch.empty(24 << 20, device=dev)
    fn = lambda: ops.gemm_tn(a, b, c, colsum=cs, workspace=ws)
elif name == "proj_ffn":  # the whole-block kernel, training instance (H, z, y saved; both LayerNorm tails)
    D, FF = 192, 2048
    a = torch.randn((T, D), device=dev).to(bf); xr = torch.randn((T, D), device=dev).to(bf)
    w1 = (torch.randn((FF, D), device=dev) / D ** .5).to(bf); w2 = (torch.randn((D, FF), device=dev) / FF ** .5).to(bf)
    wo = (torch.randn((D, D), device=dev) / D ** .5).to(bf)
    slab = torch.cat([w1.reshape(-1), w2.reshape(-1), wo.reshape(-1)])
    pkp = torch.empty(ops.ffn_proj_packed_bytes(D, FF) // 2, device=dev, dtype=bf)
    ops.ffn_pack_proj_batched(slab, pkp, torch.tensor([0, w1.numel(), w1.numel() + w2.numel(), -1, 0], device=dev), 1, D, FF)
    z0, f0 = torch.zeros(D, device=dev), torch.zeros(FF, device=dev)
    ln = (torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-5)
    y = torch.empty((T, D), device=dev, dtype=bf); x1 = torch.empty((T, D), device=dev, dtype=bf); z = torch.empty((T, D), device=dev, dtype=bf)
    h = torch.empty((T, FF), device=dev, dtype=bf); st = (torch.empty(T, device=dev), torch.empty(T, device=dev))
    wq = (torch.randn((3 * D, D), device=dev) / D ** .5).to(bf)
    slab = torch.cat([slab, wq.reshape(-1)])
    ops.ffn_pack_proj_batched(slab, pkp, torch.tensor([0, w1.numel(), w1.numel() + w2.numel(), w1.numel() + w2.numel() + wo.numel(), 0], device=dev), 1, D, FF)
    bq = torch.zeros(3 * D, device=dev); qkv = torch.empty((T, 3 * D), device=dev, dtype=bf)
    rb_ = ops.relu_bits_buffer(T, FF, dev)   # round 2: the training instance also records the ReLU bit pattern
    fn = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=h, ln_b=ln, stats_a=st, stats_b=st,
                                     qkv_bias=bq, qkv=qkv, relu_bits=rb_)
elif name == "ffn_bwd_dx":  # dx1 = dz + ((dz W2) * [H > 0]) W1 in one launch, dpre written for the dW1 GEMM
    D, FF = 192, 2048
    dz = torch.randn((T, D), device=dev).to(bf)
    w1t = (torch.randn((D, FF), device=dev) / D ** .5).to(bf); w2t = (torch.randn((FF, D), device=dev) / FF ** .5).to(bf)
    pkb = ops.ffn_pack(w2t, w1t)
    rb_ = torch.randint(0, 256, (int(ops.relu_bits_buffer(T, FF, dev).numel()),), device=dev, dtype=torch.uint8)
    dx = torch.empty((T, D), device=dev, dtype=bf); dp = torch.empty((T, FF), device=dev, dtype=bf)
    fn = lambda: ops.ffn_bwd_dx(dz, pkb, rb_, dx1=dx, dpre=dp)
elif name in ("attn_fwd", "attn_bwd"):
    rb = RaggedBatch([3] * (T // 589), 196, dev)
    qkv = torch.randn((rb.T, 576), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
    do = torch.randn((rb.T, 192), device=dev).to(bf); dq = torch.empty_like(qkv); dl = torch.empty((2, rb.T), device=dev)
    fn = (lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2, out=o, lse=lse)) if name == "attn_fwd" else \
         (lambda: ops.attn_bwd(qkv, o, do, lse, rb.cu_seqlens, rb.work, 2, dqkv=dq, delta=dl))
elif name in ("attn_fwd_base", "attn_bwd_base", "attn_fwd_small", "attn_bwd_small"):
    # Base: 64 images x 10 channels (1961 tokens), 2 heads of 384; Small: 120 images x 10 channels, 2 heads of 192
    base = name.endswith("base")
    D = 768 if base else 384
    rb = RaggedBatch([10] * (64 if base else 120), 196, dev)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
    do = torch.randn((rb.T, D), device=dev).to(bf); dq = torch.empty_like(qkv); dl = torch.empty((2, rb.T), device=dev)
    fn = (lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2, out=o, lse=lse)) if "fwd" in name else \
         (lambda: ops.attn_bwd(qkv, o, do, lse, rb.cu_seqlens, rb.work, 2, dqkv=dq, delta=dl))
for _ in range(4): fn()
torch.cuda.synchronize()
