"""attention fwd / bwd time vs sequence length at constant total tokens (per-block vs per-tile cost)."""
import sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
D, H = int(sys.argv[1]) if len(sys.argv) > 1 else 192, 2
def timeit(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
for C in (1, 3, 6, 10):
    n = 1 + 196 * C
    B = max(1, 301568 // n)
    rb = RaggedBatch([C] * B, 196, dev)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    do = torch.randn((rb.T, D), device=dev).to(bf); dq = torch.empty_like(qkv); dl = torch.empty((H, rb.T), device=dev)
    tf = timeit(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=o, lse=lse))
    tb = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, rb.cu_seqlens, rb.work, H, dqkv=dq, delta=dl))
    fl = 4.0 * B * n * n * D
    print(f"D={D} N={n:5d} B={B:5d} T={rb.T}: fwd {tf:7.1f} us = {fl / tf / 1e6:6.0f} TF/s   bwd {tb:7.1f} us = {2.5 * fl / tb / 1e6:6.0f} TF/s")
