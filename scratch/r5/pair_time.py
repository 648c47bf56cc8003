import os, sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
def t(fn, reps=10, rounds=5):
    for _ in range(3): fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]
rb = RaggedBatch([10] * 64, 196, dev); torch.manual_seed(0)
qkv = torch.randn((rb.T, 3 * 768), device=dev).to(bf)
o, l = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
print(f"{os.environ.get('TAG', '?'):8s} pair={os.environ.get('CHADAVIT_ATTN_FWD_PAIR', '0')}  cfg5 global {t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2, out=o, lse=l)):8.1f} us", flush=True)
