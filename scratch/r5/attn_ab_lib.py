"""One library's attention timings at the bench shapes (run once per library; CHADAVIT_HIP_LIB selects, TAG labels)."""
import os, sys, torch, random
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
from ctypes import c_int
dev = torch.device('cuda:0'); bf = torch.bfloat16
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
random.seed(1)
modes = [int(x) for x in os.environ.get('MODES', '0').split(',')]
for name, nch, p, D, H in (("cfg2 global 1024", [3] * 2048, 196, 192, 2), ("cfg2 local 1024", [3] * 8192, 36, 192, 2),
                           ("cfg3 global 128", [random.randint(1, 10) for _ in range(256)], 196, 384, 2),
                           ("cfg3 local 128", [random.randint(1, 10) for _ in range(1024)], 36, 384, 2),
                           ("cfg5 global 32", [10] * 64, 196, 768, 2), ("mixed tiny 256", [random.randint(1, 10) for _ in range(512)], 196, 192, 2)):
    rb = RaggedBatch(nch, p, dev)
    torch.manual_seed(0)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf); do = torch.randn((rb.T, D), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    dq = torch.empty_like(qkv); dl = torch.empty((H, rb.T), device=dev)
    def parts(pp):
        T, D3 = qkv.shape
        rc = ops.lib().chadavit_attn_bwd_parts(ops._ptr(qkv), ops._ptr(o), ops._ptr(do), ops._ptr(lse), ops._ptr(dq), ops._ptr(dl), ops._ptr(rb.cu_seqlens), ops._ptr(rb.work),
                                               c_int(rb.work.shape[0]), c_int(T), c_int(D3 // 3), c_int(H), c_int(pp), ops._stream())
        assert rc == 0
    for m in modes:
        os.environ['CHADAVIT_ATTN_PERSISTENT'] = str(m)
        tf = t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=o, lse=lse))
        tq = t(lambda: parts(3)); tk = t(lambda: parts(4))
        print(f"{os.environ.get('TAG', '?'):6s} mode {m} {name:18s} T={rb.T:8d}: fwd {tf:7.1f}  dq {tq:7.1f}  dkv {tk:7.1f}  pair {tq + tk:7.1f} us", flush=True)
