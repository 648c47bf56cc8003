"""Whole-block kernel, training instance: which outputs cost what (M = 301568)."""
import os, sys, torch
sys.argv = ["one_op.py", "proj_ffn", "."]
ns = {}
exec(compile(open("scratch/one_op.py").read().split("for _ in range(4): fn()")[0], "one_op", "exec"), ns)
ops = ns["ops"]; g = ns
a, xr, pkp, z0, ln, f0, y, x1, z, h, st, bq, qkv, rb_ = (g[k] for k in ("a", "xr", "pkp", "z0", "ln", "f0", "y", "x1", "z", "h", "st", "bq", "qkv", "rb_"))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
full = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=h, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv, relu_bits=rb_)
no_h = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=None, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv, relu_bits=rb_)
no_h_bits = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=None, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv)
h_no_bits = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=h, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv)
infer = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, ln_b=ln, qkv_bias=bq, qkv=qkv, want_x1=False, want_hn=False)
_orig_req = ops._req
ops._req = lambda t, dt, nm: None if nm == 'h' else _orig_req(t, dt, nm)   # (the aliased view is not contiguous)
h_alias = torch.empty((1, h.shape[1]), device=h.device, dtype=h.dtype).expand(h.shape[0], h.shape[1])   # row stride 0: every H row lands on the same 4 KiB
full_alias = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, y=y, x1=x1, stats1=st, z=z, h=h_alias, ln_b=ln, stats_a=st, stats_b=st, qkv_bias=bq, qkv=qkv, relu_bits=rb_)
h_small = torch.empty((4096, h.shape[1]), device=h.device, dtype=h.dtype)
for name, fn in (("full, H rows aliased (ldh = 0)", full_alias), ("full (H + bits + y x1 z x2 hn qkv)", full), ("no H", no_h), ("no H, no bits", no_h_bits), ("H, no bits", h_no_bits), ("inference (x2 qkv only)", infer)):
    print(f"{name:40s} {t(fn):8.1f} us")
