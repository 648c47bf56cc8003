"""same-box A/B of two builds of the library on the attention kernels (cfg2 / cfg3 / cfg5 shapes): run once per lib"""
import os, sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
import random
for D, chans, p in ((192, [3] * 1024, 196), (192, [3] * 4096, 36), (384, None, 196), (768, [10] * 64, 196)):
    if chans is None:
        random.seed(0); chans = [random.randint(1, 10) for _ in range(256)]
    rb = RaggedBatch(chans, p, dev)
    torch.manual_seed(0)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
    do = torch.randn((rb.T, D), device=dev).to(bf); dq = torch.empty_like(qkv); dl = torch.empty((2, rb.T), device=dev)
    tf = timeit(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2, out=o, lse=lse))
    tb = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, rb.cu_seqlens, rb.work, 2, dqkv=dq, delta=dl))
    print(f"{os.environ.get('CHADAVIT_HIP_LIB', 'tree')}: D={D} seqs={len(chans)} T={rb.T}: fwd {tf:7.1f} us  bwd {tb:7.1f} us  checks {o.float().abs().sum().item():.6e} {dq.float().abs().sum().item():.6e}")
