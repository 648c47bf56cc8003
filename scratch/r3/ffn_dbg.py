"""Phase cycle counts of the block kernel, no-grad instance (instrumented build: CHADAVIT_HIP_LIB=.../libchadavit_hip_dbg.so), wave 0 of blocks 0 (first round, cold), 2049 and 4000."""
import sys, ctypes, torch
sys.path.insert(0, ".")
from chadavit_amd import ops
from chadavit_amd._lib import lib
dev = torch.device("cuda:0"); bf = torch.bfloat16
T, D, FF = 603136, 192, 2048
a = torch.randn((T, D), device=dev).to(bf); xr = torch.randn((T, D), device=dev).to(bf)
w1 = (torch.randn((FF, D), device=dev) / D ** .5).to(bf); w2 = (torch.randn((D, FF), device=dev) / FF ** .5).to(bf)
wo = (torch.randn((D, D), device=dev) / D ** .5).to(bf); wq = (torch.randn((3 * D, D), device=dev) / D ** .5).to(bf)
slab = torch.cat([w1.reshape(-1), w2.reshape(-1), wo.reshape(-1), wq.reshape(-1)])
pkp = torch.empty(ops.ffn_proj_packed_bytes(D, FF) // 2, device=dev, dtype=bf)
ops.ffn_pack_proj_batched(slab, pkp, torch.tensor([0, w1.numel(), w1.numel() + w2.numel(), w1.numel() + w2.numel() + wo.numel(), 0], device=dev), 1, D, FF)
z0, f0 = torch.zeros(D, device=dev), torch.zeros(FF, device=dev)
ln = (torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-5)
bq = torch.zeros(3 * D, device=dev)
fn = lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, ln_b=ln, qkv_bias=bq, want_hn=False)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
print(f"no-grad block kernel, {T} rows: {e0.elapsed_time(e1) * 100:.1f} us per launch")
import sys as _s
if not hasattr(lib(), "chadavit_ffn_dbg_read"): _s.exit(0)
buf = (ctypes.c_ulonglong * 32)()
lib().chadavit_ffn_dbg_read(buf)
for nm, o in (("block 0", 0), ("block 2049", 8), ("block 4000", 16)):
    v = list(buf)[o:o + 8]
    nc = max(1, v[7])
    print(f"  {nm}: block lifetime {v[0]} cycles | prologue (out-proj, LN1) {v[1]} | FFN loop {v[2]} = {nc - 1} iterations: barrier+wait {v[3] / (nc - 1):.0f} + core {v[4] / (nc - 1):.0f} per iteration | last GEMM2 + to tail {v[5]} | tails + QKV postlogue {v[6]}")
v = list(buf)[24:31]
print(f"  block 2049 prologue: start -> first projection barrier wait begins {v[0]} | wait for stream block 0 {v[1]} | step 0 {v[2]} | wait {v[3]} | step 1 {v[4]} | wait {v[5]} | step 2 (+ residual loads issued) {v[6]}")
