"""Whole-block kernel (training / no-grad) and the backward dX instance, one library (CHADAVIT_HIP_LIB selects, TAG labels); ONE_OP_T rows."""
import os, sys, torch
os.environ.setdefault("ONE_OP_T", "603136")
def setup(name):
    sys.argv = ["one_op.py", name, "."]
    ns = {}
    exec(compile(open("scratch/one_op.py").read().split("for _ in range(4): fn()")[0], "one_op", "exec"), ns)
    return ns
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
g = setup("proj_ffn"); ops = g["ops"]
a, xr, pkp, z0, ln, f0, y, x1, z, h, st, bq, qkv, rb_ = (g[k] for k in ("a", "xr", "pkp", "z0", "ln", "f0", "y", "x1", "z", "h", "st", "bq", "qkv", "rb_"))
full = g["fn"]
infer = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, ln_b=ln, qkv_bias=bq, qkv=qkv, want_x1=False, want_hn=False)
tf, ti = t(full), t(infer)
del g, a, xr, y, x1, z, h, qkv; torch.cuda.empty_cache()
g2 = setup("ffn_bwd_dx"); tb = t(g2["fn"])
print(f"{os.environ.get('TAG', '?'):8s} T={os.environ['ONE_OP_T']}: training {tf:7.1f}  no-grad {ti:7.1f}  bwd dX {tb:7.1f} us", flush=True)
