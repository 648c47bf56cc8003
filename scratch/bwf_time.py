"""Time the fused block backward dX kernel against the four launches it replaces (and its debug variants)."""
import os, sys, math, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
T = int(os.environ.get("ONE_OP_T", "301568")); D, FF = 192, 2048
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(s, device=dev, generator=g)
y = r(T, D).to(bf); dx2 = r(T, D).to(bf); z = r(T, D).to(bf)
w1 = (r(FF, D) / D ** .5).to(bf); w2 = (r(D, FF) / FF ** .5).to(bf); wo = (r(D, D) / D ** .5).to(bf)
g1 = torch.ones(D, device=dev); g2 = torch.ones(D, device=dev)
st = torch.empty((4, T), device=dev)
ops.layernorm_fwd(y, g1, torch.zeros_like(g1), 1e-5, mean=st[0], rstd=st[1])
ops.layernorm_fwd(z, g2, torch.zeros_like(g2), 1e-5, mean=st[2], rstd=st[3])
bits = torch.randint(0, 256, (int(ops.relu_bits_buffer(T, FF, dev).numel()),), device=dev, dtype=torch.uint8)
ln_ws = ops.layernorm_bwd_workspace(D, dev)
gr = [torch.zeros(D, device=dev) for _ in range(4)]
pkb = ops.ffn_pack(w2.t().contiguous(), w1.t().contiguous())
dpre = torch.empty((T, FF), device=dev, dtype=bf)
wot = wo.t().contiguous()
slab = torch.cat([w2.t().contiguous().view(-1), w1.t().contiguous().view(-1), wot.view(-1)])
desc = torch.tensor([0, FF * D, 2 * FF * D, -1, 0], device=dev, dtype=torch.int64)
pkp = torch.zeros(ops.ffn_proj_packed_bytes(D, FF) // 2, device=dev, dtype=bf)
ops.ffn_pack_proj_batched(slab, pkp, desc, 1, D, FF)
ws = ops.block_bwd_dx_workspace(T, D, dev)
bufs = [torch.empty((T, D), device=dev, dtype=bf) for _ in range(4)]
def chain():
    dz = ops.layernorm_bwd(dx2, z, st[2], st[3], g2, gr[0], gr[1], ln_ws, dx=bufs[0])
    dx1 = ops.ffn_bwd_dx(dz, pkb, bits, dx1=bufs[1], dpre=dpre)
    dy = ops.layernorm_bwd(dx1, y, st[0], st[1], g1, gr[2], gr[3], ln_ws, dx=bufs[2])
    ops.gemm_nt(dy, wot, out=bufs[3])
def fused():
    ops.block_bwd_dx(dx2, z, st[2], st[3], g2, pkp, bits, dpre, y, st[0], st[1], g1, gr[0], gr[1], gr[2], gr[3], ws, dz=bufs[0], dy=bufs[2], da=bufs[3])
def plain():
    ops.ffn_bwd_dx(dx2, pkb, bits, dx1=bufs[1], dpre=dpre)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"T={T}  chain {t(chain):.1f} us   plain ffn_bwd_dx {t(plain):.1f} us")
for dbg in os.environ.get("BWF_DBGS", "0,1,2,4,6,7,3,5").split(","):
    os.environ["CHADA_BWF_DBG"] = dbg
    print(f"  fused dbg={dbg}: {t(fused):.1f} us")
