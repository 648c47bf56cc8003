"""CLS-row attention kernels at the bench shapes (cfg2 global / local, cfg3-like mixed, cfg5)."""
import sys, random, torch
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
random.seed(0)
for name, D, chans, p in (("cfg2 global", 192, [3] * 1024, 196), ("cfg2 local", 192, [3] * 4096, 36), ("cfg3 global", 384, [random.randint(1, 10) for _ in range(256)], 196),
                          ("cfg5", 768, [10] * 64, 196)):
    rb = RaggedBatch(chans, p, dev)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o, lse = ops.attn_cls_fwd(qkv, rb.cu_seqlens, 2)
    do = torch.randn((len(chans), D), device=dev).to(bf)
    dq = torch.empty_like(qkv)
    tf = timeit(lambda: ops.attn_cls_fwd(qkv, rb.cu_seqlens, 2))
    tb = timeit(lambda: ops.attn_cls_bwd(qkv, rb.cu_seqlens, o, do, lse, 2, dqkv=dq))
    print(f"{name:12s} T={rb.T}: cls fwd {tf:7.1f} us ({rb.T * D * 4 / tf / 1e6:.2f} TB/s)  cls bwd {tb:7.1f} us ({rb.T * D * 10 / tb / 1e6:.2f} TB/s)")
