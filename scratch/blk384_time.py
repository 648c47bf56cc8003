"""Whole-block kernel at D = 384 (Small): inference and training instance timings (CHADAVIT_HIP_LIB selects the build)."""
import os, sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
T, D, FF = 254664, 384, 2048
a = torch.randn((T, D), device=dev).to(bf); xr = torch.randn((T, D), device=dev).to(bf)
w1 = (torch.randn((FF, D), device=dev) / D ** .5).to(bf); w2 = (torch.randn((D, FF), device=dev) / FF ** .5).to(bf)
wo = (torch.randn((D, D), device=dev) / D ** .5).to(bf); wq = (torch.randn((3 * D, D), device=dev) / D ** .5).to(bf)
slab = torch.cat([w1.reshape(-1), w2.reshape(-1), wo.reshape(-1), wq.reshape(-1)])
pkp = torch.empty(ops.ffn_proj_packed_bytes(D, FF) // 2, device=dev, dtype=bf)
o3 = w1.numel() + w2.numel() + wo.numel()
ops.ffn_pack_proj_batched(slab, pkp, torch.tensor([0, w1.numel(), w1.numel() + w2.numel(), o3, 0], device=dev), 1, D, FF)
z0, f0 = torch.zeros(D, device=dev), torch.zeros(FF, device=dev)
ln = (torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-5)
bq = torch.zeros(3 * D, device=dev); qkv = torch.empty((T, 3 * D), device=dev, dtype=bf)
y = torch.empty((T, D), device=dev, dtype=bf); x1 = torch.empty_like(y); z = torch.empty_like(y)
h = torch.empty((T, FF), device=dev, dtype=bf); st = (torch.empty(T, device=dev), torch.empty(T, device=dev))
rb_ = ops.relu_bits_buffer(T, FF, dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
infer = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, ln_b=ln, qkv_bias=bq, qkv=qkv, want_x1=False, want_hn=False)
train = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=h, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv, relu_bits=rb_)
fl = T * (4.0 * D * FF + 2.0 * D * D * 4)
ti = t(infer); print(f"D=384 inference: {ti:.1f} us  {fl / ti / 1e6:.0f} TFLOP/s   lib {os.environ.get('CHADAVIT_HIP_LIB', 'default')}")
if os.environ.get("SKIP_TRAIN") is None:
    tt = t(train); print(f"D=384 training:  {tt:.1f} us  {fl / tt / 1e6:.0f} TFLOP/s")
