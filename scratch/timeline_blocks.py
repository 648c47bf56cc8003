"""Per-block phase durations and residency of the fused block kernel (debug lib with -DCHADA_FFN_TIMELINE)."""
import os, sys, ctypes, numpy as np, torch
os.environ["CHADAVIT_HIP_LIB"] = "scratch/ab/lib_timeline.so"
sys.argv = ["one_op.py", "proj_ffn", "."]
ns = {}
exec(compile(open("scratch/one_op.py").read().split("for _ in range(4): fn()")[0], "one_op", "exec"), ns)
ops = ns["ops"]; g = ns
from chadavit_amd._lib import lib
a, xr, pkp, z0, ln, f0, y, x1, z, h, st, bq, qkv, rb_ = (g[k] for k in ("a", "xr", "pkp", "z0", "ln", "f0", "y", "x1", "z", "h", "st", "bq", "qkv", "rb_"))
variants = {
    "training (H + record)": lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=h, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv, relu_bits=rb_),
    "inference": lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, ln_b=ln, qkv_bias=bq, qkv=qkv, want_x1=False, want_hn=False),
}
nblk = (a.shape[0] + 127) // 128
buf = np.zeros(4096 * 8, dtype=np.uint64)
for name, fn in variants.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    fn(); torch.cuda.synchronize()
    lib().chadavit_debug_blocks(ctypes.c_void_p(buf.ctypes.data))
    t = buf.reshape(4096, 8)[:nblk].astype(np.int64)
    t0 = t[:, 0].min()
    dur = t[:, 4] - t[:, 0]
    ph = [t[:, i + 1] - t[:, i] for i in range(4)]
    hw = buf.reshape(4096, 8)[:nblk, 7]
    cu = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64) * 1000 + ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64) * 100 + ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64) * 16 + ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64)
    ncu = len(np.unique(cu))
    conc, spans, nb = [], [], []
    for c in np.unique(cu):
        m = cu == c
        span = t[m, 4].max() - t[m, 0].min()
        conc.append(dur[m].sum() / span); spans.append(span); nb.append(m.sum())
    print(f"{name}: blocks {nblk} on {ncu} distinct CU ids; per CU: blocks {np.mean(nb):.1f} (min {np.min(nb)}, max {np.max(nb)}), busy span {np.mean(spans) / 2.3e3:.0f} us (max {np.max(spans) / 2.3e3:.0f}), mean concurrency {np.mean(conc):.2f} (min {np.min(conc):.2f})")
    print(f"   per block: total {dur.mean():.0f} cycles ({dur.mean() / 2.3e3:.1f} us) | prologue {ph[0].mean():.0f} | chunk loop {ph[1].mean():.0f} | LN tails {ph[2].mean():.0f} | QKV postlogue {ph[3].mean():.0f}")
    tl = np.zeros(8 * 80 * 8, dtype=np.uint64)
    lib().chadavit_debug_timeline(ctypes.c_void_p(tl.ctypes.data))
    tl = tl.reshape(8, 80, 8).astype(np.int64)
    pro = [(int(np.mean(tl[:, 70 + j, 1] - tl[:, 70 + j, 0])), int(np.mean(tl[:, 70 + j + 1, 0] - tl[:, 70 + j, 1])) if j < 2 else -1) for j in range(3)]
    post = [(int(np.mean(tl[:, 60 + q, 1] - tl[:, 60 + q, 0])), int(np.mean(tl[:, 60 + q + 1, 0] - tl[:, 60 + q, 1])) if q < 8 else -1) for q in range(9)]
    print(f"   prologue steps (wait, work): {pro}; start -> first step {int(np.mean(tl[:, 70, 0] - t[1200:1208, 0]))}; P loop end -> loop start (LN1, stores) {int(np.mean(t[:, 1] - t[:, 5]))}")
    print(f"   postlogue blocks (wait, work): {post}")
