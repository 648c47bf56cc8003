import sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
rb = RaggedBatch([10] * 64, 196, dev)   # cfg5: 64 global images x 10 channels -> 1961 tokens each
qkv = torch.randn((rb.T, 3 * 768), device=dev).to(bf)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
us = t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2, out=o, lse=lse))
fl = 4.0 * 64 * 1961 ** 2 * 768
print(f"attn_fwd dh=384 T={rb.T}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s")
