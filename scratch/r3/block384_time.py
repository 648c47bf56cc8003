"""The whole-block kernel at D = 384 (Small, cfg3's row count) and D = 192: no-grad and training instances, HIP-event timing."""
import sys, torch
sys.path.insert(0, ".")
from chadavit_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
def setup(T, D, FF=2048):
    a = torch.randn((T, D), device=dev).to(bf); xr = torch.randn((T, D), device=dev).to(bf)
    w1 = (torch.randn((FF, D), device=dev) / D ** .5).to(bf); w2 = (torch.randn((D, FF), device=dev) / FF ** .5).to(bf)
    wo = (torch.randn((D, D), device=dev) / D ** .5).to(bf); wq = (torch.randn((3 * D, D), device=dev) / D ** .5).to(bf)
    slab = torch.cat([w1.reshape(-1), w2.reshape(-1), wo.reshape(-1), wq.reshape(-1)])
    pkp = torch.empty(ops.ffn_proj_packed_bytes(D, FF) // 2, device=dev, dtype=bf)
    ops.ffn_pack_proj_batched(slab, pkp, torch.tensor([0, w1.numel(), w1.numel() + w2.numel(), w1.numel() + w2.numel() + wo.numel(), 0], device=dev), 1, D, FF)
    z0, f0 = torch.zeros(D, device=dev), torch.zeros(FF, device=dev)
    ln = (torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-5)
    bq = torch.zeros(3 * D, device=dev)
    y = torch.empty((T, D), device=dev, dtype=bf); z = torch.empty((T, D), device=dev, dtype=bf); h = torch.empty((T, FF), device=dev, dtype=bf)
    st = (torch.empty(T, device=dev), torch.empty(T, device=dev)); rb = ops.relu_bits_buffer(T, FF, dev)
    nograd = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, ln_b=ln, qkv_bias=bq, want_hn=False)
    train = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, stats1=st, z=z, h=h, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, relu_bits=rb)
    return nograd, train
for T, D in ((254664, 384), (603136, 192)):
    for nm, fn in zip(("no-grad", "training"), setup(T, D)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
        print(f"D {D} rows {T} {nm}: {e0.elapsed_time(e1) * 100:.1f} us", flush=True)
