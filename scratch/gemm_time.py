import sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def t(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
M = int(sys.argv[1]) if len(sys.argv) > 1 else 301568
for name, N, K, epi in (("qkv", 576, 192, 0), ("dH", 2048, 192, 4), ("ffn1", 2048, 192, 1), ("dx1", 192, 2048, 3), ("outproj", 192, 192, 3), ("dh", 192, 576, 0)):
    x = torch.randn((M, K), device=dev).to(bf); w = (torch.randn((N, K), device=dev) / K ** .5).to(bf)
    bias = torch.zeros(N, device=dev); aux = torch.randn((M, N), device=dev).to(bf) if epi in (3, 4) else None
    o = torch.empty((M, N), device=dev, dtype=bf)
    us = t(lambda: ops.gemm_nt(x, w, out=o, bias=bias, epilogue=epi, aux=aux))
    nb = 2.0 * (M * K + N * K + M * N * (2 if epi in (3, 4) else 1))
    print(f"{name}: {us:.1f} us  {2.0*M*N*K/us/1e6:.0f} TF/s  {nb/us/1e6:.2f} TB/s", flush=True)
for name, I, J in (("dW1", 2048, 192), ("dW2", 192, 2048)):
    a = torch.randn((M, I), device=dev).to(bf); b = torch.randn((M, J), device=dev).to(bf)
    c = torch.empty((I, J), device=dev); cs = torch.empty(I, device=dev); ws = torch.empty(24 << 20, device=dev)
    us = t(lambda: ops.gemm_tn(a, b, c, colsum=cs, workspace=ws))
    print(f"{name}: {us:.1f} us  {2.0*M*(I+J)/us/1e6:.2f} TB/s", flush=True)
M = 254664
for name, N, K, epi in (("S ffn2", 384, 2048, 3), ("S ffn1", 2048, 384, 1), ("S dH", 2048, 384, 4), ("S qkv", 1152, 384, 0), ("S outproj", 384, 384, 3), ("S dh", 384, 1152, 0), ("S da", 384, 384, 0)):
    x = torch.randn((M, K), device=dev).to(bf); w = (torch.randn((N, K), device=dev) / K ** .5).to(bf)
    bias = torch.zeros(N, device=dev); aux = torch.randn((M, N), device=dev).to(bf) if epi in (3, 4) else None
    o = torch.empty((M, N), device=dev, dtype=bf)
    us = t(lambda: ops.gemm_nt(x, w, out=o, bias=bias, epilogue=epi, aux=aux))
    nb = 2.0 * (M * K + N * K + M * N * (2 if epi in (3, 4) else 1))
    print(f"{name}: {us:.1f} us  {2.0*M*N*K/us/1e6:.0f} TF/s  {nb/us/1e6:.2f} TB/s", flush=True)
