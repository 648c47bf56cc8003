"""bf16 gemm_nt at the Base / Small long-K shapes (dX GEMMs of cfg5 / cfg3 and the bf16 forward of cfg5-bf16)."""
import sys, torch
sys.path.insert(0, ".")
from chadavit_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
def run(M, N, K, epi):
    x = torch.randn((M, K), device=dev).to(bf); w = (torch.randn((N, K), device=dev) / K ** .5).to(bf)
    bias = torch.zeros(N, device=dev); aux = torch.randn((M, N), device=dev).to(bf) if epi in (3, 4) else None
    o = torch.empty((M, N), device=dev, dtype=bf)
    fn = lambda: ops.gemm_nt(x, w, out=o, bias=bias, epilogue=epi, aux=aux)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"M {M} N {N:5d} K {K:5d} epi {epi}: {us:8.1f} us  {2.0*M*N*K/us*1e-6:6.0f} TF/s", flush=True)
for M, N, K, epi in ((125504, 768, 2304, 0), (125504, 768, 2048, 3), (125504, 768, 768, 0), (125504, 768, 768, 3), (125504, 2304, 768, 0), (125504, 2048, 768, 4), (254664, 384, 2048, 3), (254664, 384, 1152, 0)):
    run(M, N, K, epi)
