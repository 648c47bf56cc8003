"""Per-phase cycle counts of the fused block kernel's chunk loop (debug lib built with -DCHADA_FFN_TIMELINE)."""
import os, sys, ctypes, numpy as np, torch
os.environ["CHADAVIT_HIP_LIB"] = "scratch/ab/lib_timeline.so"
sys.argv = ["one_op.py", "proj_ffn", "."]
ns = {}
exec(compile(open("scratch/one_op.py").read().split("for _ in range(4): fn()")[0], "one_op", "exec"), ns)
ops = ns["ops"]; g = ns
from chadavit_amd._lib import lib
a, xr, pkp, z0, ln, f0, y, x1, z, h, st, bq, qkv, rb_ = (g[k] for k in ("a", "xr", "pkp", "z0", "ln", "f0", "y", "x1", "z", "h", "st", "bq", "qkv", "rb_"))
variants = {
    "full (H + bits)": lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=h, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv, relu_bits=rb_),
    "H, no bits": lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=h, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv),
    "no H, bits": lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=None, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv, relu_bits=rb_),
    "no H, no bits": lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=None, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv),
}
buf = np.zeros(8 * 80 * 8, dtype=np.uint64)
for name, fn in variants.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    fn(); torch.cuda.synchronize()
    rc = lib().chadavit_debug_timeline(ctypes.c_void_p(buf.ctypes.data))
    t = buf.reshape(8, 80, 8).astype(np.int64)
    ks = np.arange(8, 60)
    # phases: 0 arrive at wait, 1 after barrier, 2 before ffn_core, 3 after the MFMA loop issue, 4 after post / slab write
    rows = []
    for b in range(8):
        tb = t[b]
        it = tb[ks + 1, 0] - tb[ks, 0]
        rows.append([np.mean(it), np.mean(tb[ks, 1] - tb[ks, 0]), np.mean(tb[ks, 2] - tb[ks, 1]), np.mean(tb[ks, 3] - tb[ks, 2]),
                     np.mean(tb[ks, 4] - tb[ks, 3]), np.mean(tb[ks + 1, 0] - tb[ks, 4])])
    r = np.mean(np.array(rows), 0)
    print(f"{name:16s} rc={rc} cycles/iteration {r[0]:7.0f} | wait+barrier {r[1]:6.0f} | pre (stores, masks) {r[2]:6.0f} | DMA issue + MFMA loop {r[3]:6.0f} | post + slab {r[4]:6.0f} | tail (gather) {r[5]:6.0f}")
