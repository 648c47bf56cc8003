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
T = 301568
ws = torch.empty(24 << 20, device=dev)
for I, J in ((576, 192), (192, 192), (2048, 192), (192, 2048)):
    a = torch.randn((T, I), device=dev).to(bf); b = torch.randn((T, J), device=dev).to(bf)
    c = torch.empty((I, J), device=dev); cs = torch.empty(I, device=dev)
    us = t(lambda: ops.gemm_tn(a, b, c, colsum=cs, workspace=ws))
    ref = a[:4096].float().t() @ b[:4096].float()
    print(f"gemm_tn T={T} {I}x{J}: {us:.1f} us  {2.0 * T * (I + J) / us / 1e6:.2f} TB/s")
