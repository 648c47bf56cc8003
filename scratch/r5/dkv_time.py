import os, sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
from ctypes import c_int
dev = torch.device('cuda:0'); bf = torch.bfloat16
D, H, p = 192, 2, 196
def t(fn, reps=20, rounds=5):
    for _ in range(3): fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]
line = os.environ.get('TAG', '?') + ':'
for C in (3, 1):
    B = 1200000 // (1 + C * p)
    rb = RaggedBatch([C] * B, p, dev)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf); do = torch.randn((rb.T, D), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    dq = torch.empty_like(qkv); dl = torch.empty((H, rb.T), device=dev)
    def parts(pp):
        T, D3 = qkv.shape
        rc = ops.lib().chadavit_attn_bwd_parts(ops._ptr(qkv), ops._ptr(o), ops._ptr(do), ops._ptr(lse), ops._ptr(dq), ops._ptr(dl), ops._ptr(rb.cu_seqlens), ops._ptr(rb.work),
                                               c_int(rb.work.shape[0]), c_int(T), c_int(D3 // 3), c_int(H), c_int(pp), ops._stream())
        assert rc == 0
    parts(3)
    for m in (0, 1):
        os.environ['CHADAVIT_ATTN_PERSISTENT'] = str(m)
        line += f"  C={C} mode {m}: dkv {t(lambda: parts(4)):7.1f} us"
print(line, flush=True)
