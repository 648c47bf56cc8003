"""Library attention (torch SDPA -> the ROCm flash / memory-efficient kernels torch ships) on the step's attention shapes beside the
build's kernels.  cfg2: 2048 sequences x 589 tokens, 2 heads x 96; local crops: 8192 x 109; cfg5-like: 64 x 1961, 2 x 384."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

from chadavit_amd import ops
from chadavit_amd.ragged import ragged_batch

dev = torch.device("cuda:0")


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, C, p, H, dh) in ((2048, 3, 196, 2, 96), (8192, 3, 36, 2, 96), (64, 10, 196, 2, 384), (256, 5, 196, 2, 192)):
    N = 1 + C * p
    D = H * dh
    g = torch.Generator(device=dev).manual_seed(0)
    rb = ragged_batch([C] * B, p, dev)
    qkv = torch.randn(B * N, 3 * D, device=dev, dtype=torch.bfloat16, generator=g)
    dout = torch.randn(B * N, D, device=dev, dtype=torch.bfloat16, generator=g)
    out, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    t_f = timeit(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=out, lse=lse))
    dqkv = torch.empty_like(qkv)
    t_b = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, rb.cu_seqlens, rb.work, H, dqkv=dqkv))
    print(f"B={B} N={N} heads={H} dh={dh}:  chadavit fwd {t_f:8.1f} us  bwd {t_b:8.1f} us", flush=True)
    # the library: (B, H, N, dh) operands (already split and transposed: its best case, no packing cost counted)
    q, k, v = (qkv.view(B, N, 3, H, dh)[:, :, i].transpose(1, 2).contiguous().requires_grad_(True) for i in range(3))
    do = dout.view(B, N, H, dh).transpose(1, 2).contiguous()
    from torch.nn.attention import SDPBackend, sdpa_kernel
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("mem-efficient", SDPBackend.EFFICIENT_ATTENTION)):
        try:
            with sdpa_kernel(be):
                o = F.scaled_dot_product_attention(q, k, v)
                tf = timeit(lambda: F.scaled_dot_product_attention(q, k, v))

                def fb():
                    o_ = F.scaled_dot_product_attention(q, k, v)
                    o_.backward(do)
                    q.grad = k.grad = v.grad = None
                tfb = timeit(fb)
            ref = o.transpose(1, 2).reshape(B * N, D)
            err = float((ref.float() - out.float()).abs().max())
            print(f"    torch SDPA {name:14s} fwd {tf:8.1f} us  fwd+bwd {tfb:8.1f} us  (bwd ~ {tfb - tf:8.1f})   max |diff| vs chadavit {err:.3e}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"    torch SDPA {name}: not available for this shape: {repr(e)[:120]}", flush=True)
    del q, k, v, do, qkv, dout, out, lse, dqkv
    torch.cuda.empty_cache()
