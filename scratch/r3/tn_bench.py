"""gemm_tn at the cfg5 / cfg3 weight-gradient shapes."""
import sys, os, torch
sys.path.insert(0, ".")
from chadavit_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
def run(T, I, J):
    a = torch.randn((T, I), device=dev).to(bf); b = torch.randn((T, J), device=dev).to(bf)
    c = torch.empty((I, J), device=dev); cs = torch.empty(I, device=dev); ws = torch.empty(64 << 20, device=dev)
    fn = lambda: ops.gemm_tn(a, b, c, colsum=cs, workspace=ws)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"T {T} I {I} J {J}: {us:8.1f} us  {2.0*T*I*J/us*1e-6:6.0f} TF/s  {2.0*T*(I+J)/us*1e-6:5.2f} TB/s", flush=True)
for T, I, J in ((125504, 2304, 768), (125504, 768, 768), (125504, 2048, 768), (125504, 768, 2048), (254664, 1152, 384), (254664, 384, 384), (254664, 2048, 384), (254664, 384, 2048), (603136, 576, 192), (603136, 2048, 192), (603136, 192, 2048), (603136, 192, 192), (446464, 576, 192), (5120, 2048, 2048), (5120, 65536, 256)):
    run(T, I, J)
