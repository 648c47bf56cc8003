"""dh 384 forward of the loaded library (CHADAVIT_ATTN_FWD_RM picks the stage layout) at cfg5's passes and a ragged mix."""
import os, sys, torch, random
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
random.seed(3); D, H = 768, 2; res = []
for name, nch, p in (("global", [10] * 64, 196), ("ragged", [random.randint(1, 10) for _ in range(96)], 196), ("local", [10] * 256, 36)):
    rb = RaggedBatch(nch, p, dev); torch.manual_seed(0)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o0, l0 = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    res.append(f"{name} {t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=o0, lse=l0)):7.1f}")
print(f"{os.environ.get('TAG', '?'):8s} RM={os.environ.get('CHADAVIT_ATTN_FWD_RM', '0')}  " + "  ".join(res) + " us", flush=True)
