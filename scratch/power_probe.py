"""Is the block kernel running against the power budget?  Same launch on random and on all-zero operands (identical instruction
stream and HBM traffic, far less switching in the MFMA / LDS data paths)."""
import os, sys, torch
sys.argv = ["one_op.py", "proj_ffn", "."]
ns = {}
exec(compile(open("scratch/one_op.py").read().split("for _ in range(4): fn()")[0], "one_op", "exec"), ns)
fn = ns["fn"]
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"random operands: {t(fn):.1f} us")
for k in ("a", "xr", "pkp"):
    ns[k].zero_()
print(f"zero operands:   {t(fn):.1f} us")
