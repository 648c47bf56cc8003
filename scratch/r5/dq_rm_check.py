"""dh 384 dQ: fragment-major kernel (first run, saves dqkv + delta) against CHADAVIT_ATTN_DQ_RM=1, the row-major kernel as eight waves x 16 rows (second run: compares bit for bit, times)."""
import os, sys, torch, random
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
from ctypes import c_int
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
pair = os.environ.get("CHADAVIT_ATTN_DQ_RM", "0") != "0"
random.seed(3); D, H = 768, 2
for name, nch, p in (("global 64x1961", [10] * 64, 196), ("ragged 1-10", [random.randint(1, 10) for _ in range(96)], 196), ("local 256x361", [10] * 256, 36),
                     ("tiny ragged", [1, 10, 3, 2], 196), ("one short", [1], 36), ("five tokens", [1], 4)):
    rb = RaggedBatch(nch, p, dev); torch.manual_seed(0)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf); do = torch.randn((rb.T, D), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    dq = torch.zeros_like(qkv); dl = torch.empty((H, rb.T), device=dev)
    def parts(pp):
        rc = ops.lib().chadavit_attn_bwd_parts(ops._ptr(qkv), ops._ptr(o), ops._ptr(do), ops._ptr(lse), ops._ptr(dq), ops._ptr(dl), ops._ptr(rb.cu_seqlens), ops._ptr(rb.work),
                                               c_int(rb.work.shape[0]), c_int(rb.T), c_int(D), c_int(H), c_int(pp), ops._stream())
        assert rc == 0
    parts(3); torch.cuda.synchronize()
    f = f"/tmp/dq_ref_{name.replace(' ', '_')}.pt"; msg = ""
    if not pair: torch.save(dq.cpu(), f)
    else:
        r = torch.load(f)
        msg = f"bit-identical {bool((dq.cpu().view(torch.int16) == r.view(torch.int16)).all())}  max|d| {float((dq.cpu().float() - r.float()).abs().max()):.4g}"
    print(f"{'rm8' if pair else 'fm':7s} {name:16s} T={rb.T:7d}  dQ {t(lambda: parts(3)):8.1f} us  {msg}", flush=True)
