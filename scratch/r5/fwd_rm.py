"""ops.attn_fwd (whatever CHADAVIT_ATTN_FWD_RM / _M32 select in this process) against the 32x32x16 kernel called directly: results and time."""
import ctypes, os, sys, torch, random
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd._lib import lib
from chadavit_amd.ragged import RaggedBatch
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
tag = os.environ.get("TAG", "RM=" + os.environ.get("CHADAVIT_ATTN_FWD_RM", "0"))
for name, nch, p, D, H in (("cfg2 global 1024", [3] * 2048, 196, 192, 2), ("cfg2 local 1024", [3] * 8192, 36, 192, 2),
                           ("mixed tiny 256", [random.randint(1, 10) for _ in range(512)], 196, 192, 2),
                           ("cfg3 global 128", [random.randint(1, 10) for _ in range(256)], 196, 384, 2),
                           ("cfg3 local 128", [random.randint(1, 10) for _ in range(1024)], 36, 384, 2),
                           ("cfg5 global 32", [10] * 64, 196, 768, 2), ("cfg5 local 32", [10] * 256, 36, 768, 2)):
    rb = RaggedBatch(nch, p, dev)
    torch.manual_seed(0)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o0, l0 = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    o1 = torch.empty_like(o0); l1 = torch.empty_like(l0)
    rc = lib().chadavit_attn_fwd_m32(ops._ptr(qkv), ops._ptr(o1), ops._ptr(l1), ops._ptr(rb.cu_seqlens), ops._ptr(rb.work), ctypes.c_int(rb.n_work),
                                     ctypes.c_int(rb.T), ctypes.c_int(D), ctypes.c_int(H), ctypes.c_int(1), ops._stream())
    assert rc == 0 or D == 768, rc   # (the 32x32x16 kernel has no dh 384 instance in the library: the difference columns are meaningless there)
    torch.cuda.synchronize()
    print(f"{tag:8s} {name:18s} T={rb.T:8d}: max |o - o_m32| {float((o1.float() - o0.float()).abs().max()):.4f}  max |lse diff| {float((l1 - l0).abs().max()):.2e}   "
          f"attn_fwd {t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=o0, lse=l0)):8.1f} us", flush=True)
