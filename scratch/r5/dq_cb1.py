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
for name, nch, p, D, H in (("cfg2 global 1024", [3] * 2048, 196, 192, 2), ("mixed tiny 256", [random.randint(1, 10) for _ in range(512)], 196, 192, 2), ("tiny 1ch", [1] * 4096, 196, 192, 2)):
    rb = RaggedBatch(nch, p, dev)
    torch.manual_seed(0)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf); do = torch.randn((rb.T, D), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    def parts(pp, dq, dl):
        T, D3 = qkv.shape
        rc = ops.lib().chadavit_attn_bwd_parts(ops._ptr(qkv), ops._ptr(o), ops._ptr(do), ops._ptr(lse), ops._ptr(dq), ops._ptr(dl), ops._ptr(rb.cu_seqlens), ops._ptr(rb.work),
                                               c_int(rb.work.shape[0]), c_int(T), c_int(D3 // 3), c_int(H), c_int(pp), ops._stream())
        assert rc == 0
    res = {}
    line = f"{name:18s}:"
    for m in ("0", "1", "0", "1"):
        if m == "1": os.environ['CHADA_DQ96_CB1'] = "1"
        else: os.environ.pop('CHADA_DQ96_CB1', None)
        dq = torch.full_like(qkv, float('nan')); dl = torch.empty((H, rb.T), device=dev)
        parts(3, dq, dl); torch.cuda.synchronize()
        res[m] = (dq[:, :D].clone(), dl.clone())
        line += f"  cb1={m}: dq {t(lambda: parts(3, dq, dl)):7.1f} us"
    same = torch.equal(res["0"][0].view(torch.int16), res["1"][0].view(torch.int16)) and torch.equal(res["0"][1], res["1"][1])
    print(line + f"  bit-identical={same}", flush=True)
