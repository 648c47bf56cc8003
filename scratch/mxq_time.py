import sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, K in ((125504, 768), (125504, 3072), (89600, 768)):
    x = torch.randn((M, K), device=dev).to(bf)
    q, s = ops.mx8_quantize(x)
    us = t(lambda: ops.mx8_quantize(x, q=q, scales=s))
    nb = M * K * 3 + M * K / 32
    print(f"mx8_quantize {M}x{K}: {us:.1f} us  {nb / us / 1e6:.2f} TB/s")
