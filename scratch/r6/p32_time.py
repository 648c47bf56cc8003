"""Forward attention, dh 96 / 192: unpaired 32x32x16 kernel (variant 0) against the paired schedule (variant 6) -- bit identity and time.
usage: python scratch/r6/p32_time.py [check] [zero]   (CHADAVIT_HIP_LIB + CHADAVIT_ALLOW_FOREIGN_LIB=1 select a side build)"""
import sys, os, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd._lib import lib
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
check = 'check' in sys.argv; zero = 'zero' in sys.argv
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
def fwd(qkv, rb, H, variant, o, lse):
    T, D3 = qkv.shape
    rc = lib().chadavit_attn_fwd_m32(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), rb.cu_seqlens.data_ptr(), rb.work.data_ptr(), rb.work.shape[0], T, D3 // 3, H, variant,
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
g = torch.Generator().manual_seed(0)
mixed = [int(x) for x in torch.randint(1, 11, (256,), generator=g)]
shapes = [("tiny global 2048x589", [3] * 2048, 196, 192), ("tiny local 8192x109", [3] * 8192, 36, 192), ("tiny mixed 512", mixed + mixed, 196, 192),
          ("small mixed global 256", mixed, 196, 384), ("small mixed local 1024", mixed * 4, 36, 384), ("tiny c1 4096x197", [1] * 4096, 196, 192)]
if check:
    shapes += [("tiny odd", [1, 2, 3, 10, 7, 1, 1, 4, 5], 196, 192), ("small odd", [1, 2, 3, 10, 7, 1, 1, 4, 5], 196, 384), ("tiny 16px", [1, 3, 2], 1, 192), ("tiny 32px", [2, 10, 5], 4, 192)]
for name, nch, p, D in shapes:
    H = 2
    rb = RaggedBatch(nch, p, dev)
    qkv = (torch.zeros if zero else torch.randn)((rb.T, 3 * D), device=dev).to(bf)
    o0 = torch.empty((rb.T, D), device=dev, dtype=bf); l0 = torch.empty((H, rb.T), device=dev)
    o6 = torch.empty_like(o0); l6 = torch.empty_like(l0)
    fwd(qkv, rb, H, 0, o0, l0); fwd(qkv, rb, H, 6, o6, l6); torch.cuda.synchronize()
    same = bool((o0.view(torch.int16) == o6.view(torch.int16)).all()) and bool((l0 == l6).all())
    msg = f"{name}: T={rb.T} identical={same}"
    if not same:
        msg += f" max|do|={float((o0.float() - o6.float()).abs().max()):.3e} max|dlse|={float((l0 - l6).abs().max()):.3e}"
    if not check:
        t0 = t(lambda: fwd(qkv, rb, H, 0, o0, l0)); t6 = t(lambda: fwd(qkv, rb, H, 6, o6, l6))
        t0b = t(lambda: fwd(qkv, rb, H, 0, o0, l0)); t6b = t(lambda: fwd(qkv, rb, H, 6, o6, l6))
        fl = 4.0 * sum(n * n for n in rb.lens) * D
        msg += f"  unpaired {t0:.1f} / {t0b:.1f} us ({fl/t0/1e6:.0f} TF/s)  paired {t6:.1f} / {t6b:.1f} us ({fl/t6/1e6:.0f} TF/s)  ratio {t6/t0:.3f}"
    print(msg, flush=True)
if check:  # the range check's re-run: a spike far beyond the first tile's maximum (tests/test_kernels_gpu.py::test_attention_fwd_m32_variants_and_out_of_range_rerun)
    for D in (192, 384):
        dh = D // 2
        rb = RaggedBatch([3, 1, 3, 2], 196, dev)
        for spike in ((300, 500), (10, 588), (786 + 5, 786 + 588), (786 + 197 + 130, 786 + 197 + 3)):
            qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
            qkv[spike[0], :dh] = 2.5; qkv[spike[1], D:D + dh] = 2.5
            o0 = torch.empty((rb.T, D), device=dev, dtype=bf); l0 = torch.empty((2, rb.T), device=dev); o6 = torch.empty_like(o0); l6 = torch.empty_like(l0)
            fwd(qkv, rb, 2, 0, o0, l0); fwd(qkv, rb, 2, 6, o6, l6); torch.cuda.synchronize()
            same = bool((o0.view(torch.int16) == o6.view(torch.int16)).all()) and bool((l0 == l6).all())
            print(f"spike D={D} {spike}: identical={same} finite={bool(torch.isfinite(o6.float()).all())} max|do|={float((o0.float() - o6.float()).abs().max()):.3e}", flush=True)
