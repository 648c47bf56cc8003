"""dW2 TN GEMM with / without the ReLU record, and the block kernel's training instance with / without its own record (same box)."""
import sys, math, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
M, D, FF = 603136, 192, 2048
dz = torch.randn((M, D), device=dev).to(bf)
h = torch.relu(torch.randn((M, FF), device=dev)).to(bf)
c, cs = torch.zeros((D, FF), device=dev), torch.zeros(D, device=dev)
ws = torch.empty(24 * 1024 * 1024, device=dev)
bits = ops.relu_bits_buffer(M, FF, dev)
for _ in range(2):
    print(f"TN dW2 plain {timeit(lambda: ops.gemm_tn(dz, h, c, colsum=cs, workspace=ws)):7.1f} us   with record {timeit(lambda: ops.gemm_tn(dz, h, c, colsum=cs, workspace=ws, relu_bits=bits)):7.1f} us")
