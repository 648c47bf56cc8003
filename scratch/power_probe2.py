"""Random vs zero operands for the other hot kernels (attention forward / backward, dW2 TN GEMM)."""
import sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rb = RaggedBatch([3] * 512, 196, dev)
for zero in (False, True):
    qkv = (torch.zeros if zero else torch.randn)((rb.T, 576), device=dev).to(bf)
    do = (torch.zeros if zero else torch.randn)((rb.T, 192), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
    dq = torch.empty_like(qkv); dl = torch.empty((2, rb.T), device=dev)
    f = t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2, out=o, lse=lse))
    b = t(lambda: ops.attn_bwd(qkv, o, do, lse, rb.cu_seqlens, rb.work, 2, dqkv=dq, delta=dl))
    a = (torch.zeros if zero else torch.randn)((rb.T, 192), device=dev).to(bf); h = (torch.zeros if zero else torch.randn)((rb.T, 2048), device=dev).to(bf)
    c = torch.empty((192, 2048), device=dev); cs = torch.empty(192, device=dev); ws = torch.empty(24 << 20, device=dev)
    g = t(lambda: ops.gemm_tn(a, h, c, colsum=cs, workspace=ws))
    print(f"{'zero  ' if zero else 'random'} operands: attn_fwd {f:.1f} us  attn_bwd {b:.1f} us  gemm_tn dW2 {g:.1f} us")
