"""gemm_nt_mx8 at the cfg5 shapes (Base, 64 images x 10 channels: 125 504 rows): time per launch, TFLOP/s, fraction of the 5 PF MX-fp8 roof."""
import sys, torch
sys.path.insert(0, ".")
from chadavit_amd import ops
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 125504
torch.manual_seed(0)
def run(N, K, epi, emit_q, reps=20):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    xq, xs = ops.mx8_quantize(x); wq, ws = ops.mx8_quantize(w)
    aux = torch.randn(M, N, device=dev, dtype=torch.bfloat16) if epi == 3 else None
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm_nt_mx8(xq, xs, wq, ws, bias=bias, epilogue=epi, aux=aux, out=out, emit_q=emit_q)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * M * N * K
    print(f"M {M} N {N:5d} K {K:5d} epi {epi} q {int(emit_q)}: {us:8.1f} us  {fl / us * 1e-6:7.0f} TF/s  ({fl / us * 1e-6 / 5000:.3f} of 5 PF)", flush=True)
for rep in range(1):
    run(2304, 768, 0, False)
    run(768, 768, 3, False)
    run(2048, 768, 1, True)
    run(768, 2048, 3, False)
