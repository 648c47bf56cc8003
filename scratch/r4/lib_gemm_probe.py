"""Library GEMMs (torch.matmul -> hipBLASLt / rocBLAS) on the step's plain GEMM shapes, beside the hand-written kernels' times.
Only a probe: where would a library call beat the build's own kernel?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from chadavit_amd import ops

dev = torch.device("cuda:0")
T = 1206272


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device=dev).manual_seed(0)
H = torch.randn(T, 2048, device=dev, dtype=torch.bfloat16, generator=g)
dz = torch.randn(T, 192, device=dev, dtype=torch.bfloat16, generator=g)
qkv = torch.randn(T, 576, device=dev, dtype=torch.bfloat16, generator=g)
W = torch.randn(576, 192, device=dev, dtype=torch.bfloat16, generator=g)
Wt = W.t().contiguous()
out32 = torch.empty(2048, 192, device=dev, dtype=torch.float32)
print("TN dW2 = H^T dz  [2048 x 192], K = T:")
print("  torch.matmul bf16 out      ", round(timeit(lambda: torch.matmul(H.t(), dz)), 1), "us")
try:
    print("  torch.mm out fp32 (addmm)  ", round(timeit(lambda: torch.mm(H.t(), dz, out_dtype=torch.float32)), 1), "us")
except Exception as e:  # noqa: BLE001
    print("  torch.mm out_dtype: n/a", repr(e)[:80])
ws = torch.empty(24 * 1024 * 1024, device=dev, dtype=torch.float32)
gw = torch.zeros(2048, 192, device=dev, dtype=torch.float32)
print("  chadavit gemm_tn           ", round(timeit(lambda: ops.gemm_tn(H, dz, gw, workspace=ws)), 1), "us")
gw2 = torch.zeros(192, 2048, device=dev, dtype=torch.float32)
print("TN dW1 = dz^T H  [192 x 2048]:")
print("  torch.matmul               ", round(timeit(lambda: torch.matmul(dz.t(), H)), 1), "us")
print("  chadavit gemm_tn           ", round(timeit(lambda: ops.gemm_tn(dz, H, gw2, workspace=ws)), 1), "us")
print("QKV dX = dqkv W  [T x 576] x [576 x 192]:")
print("  torch.matmul               ", round(timeit(lambda: torch.matmul(qkv, W)), 1), "us")
print("  chadavit gemm_nt           ", round(timeit(lambda: ops.gemm_nt(qkv, Wt)), 1), "us")
x = torch.randn(T, 192, device=dev, dtype=torch.bfloat16, generator=g)
Wq = torch.randn(576, 192, device=dev, dtype=torch.bfloat16, generator=g)
print("QKV = x Wqkv^T  [T x 192] x [192 x 576]:")
print("  torch.matmul               ", round(timeit(lambda: torch.matmul(x, Wq.t())), 1), "us")
print("  chadavit gemm_nt           ", round(timeit(lambda: ops.gemm_nt(x, Wq)), 1), "us")
