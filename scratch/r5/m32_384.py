"""(needs scratch/r5/attention_m32_dh384_instance.patch applied) attn_fwd_m32_kernel<384> (32x32x16, 4 waves x 32 query rows, one wave per SIMD) against the dispatched attn_fwd_dma_kernel<384> (16x16x32, 8 x 16 rows):
results and time at cfg5's global pass (64 x 1961 tokens, 2 heads of 384) and a ragged mix."""
import ctypes, sys, torch, random
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd._lib import lib
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
D, H = 768, 2
random.seed(3)
for name, nch in (("cfg5 global 32", [10] * 64), ("ragged 1-10", [random.randint(1, 10) for _ in range(96)]), ("cfg5 local 32", [10] * 256)):
    rb = RaggedBatch(nch, 196 if "local" not in name else 36, dev)
    torch.manual_seed(0)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o0, l0 = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    o1 = torch.empty_like(o0); l1 = torch.empty_like(l0)
    def m32(variant=0):
        rc = lib().chadavit_attn_fwd_m32(ops._ptr(qkv), ops._ptr(o1), ops._ptr(l1), ops._ptr(rb.cu_seqlens), ops._ptr(rb.work), ctypes.c_int(rb.n_work),
                                         ctypes.c_int(rb.T), ctypes.c_int(D), ctypes.c_int(H), ctypes.c_int(variant), ops._stream())
        assert rc == 0, rc
    m32(); torch.cuda.synchronize()
    print(f"{name:16s} T={rb.T:7d}: max |o - o_ref| {float((o1.float() - o0.float()).abs().max()):.4f}  max |lse diff| {float((l1 - l0).abs().max()):.2e}   "
          f"dma<384> {t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=o0, lse=l0)):8.1f} us   m32<384> lean {t(m32):8.1f}  textbook {t(lambda: m32(1)):8.1f} us", flush=True)
