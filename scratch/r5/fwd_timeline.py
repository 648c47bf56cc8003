"""Phase sums of the 16x16x32 forward (side build with -DCHADA_FWD_TIMELINE, CHADAVIT_HIP_LIB selects it): wave 0 of every block."""
import os, sys, ctypes, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
import numpy as np
dev = torch.device('cuda:0'); bf = torch.bfloat16
D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
C = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = {768: 64, 384: 128, 192: 2048}[D] if C == 10 or D != 192 else 2048
if D == 192: n = 2048 if C == 3 else 400
rb = RaggedBatch([C] * n, 196, dev)
qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
for _ in range(3): ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2, out=o, lse=lse)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2, out=o, lse=lse); e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1)
raw = ctypes.CDLL(os.environ['CHADAVIT_HIP_LIB'])
nblk = min(4096, rb.n_work * 2)
buf = (ctypes.c_ulonglong * (nblk * 64))()
raw.chadavit_debug_read_fwd_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert raw.chadavit_debug_read_fwd_timeline(buf, nblk * 64) == 0
aw = np.array(buf, dtype=np.float64).reshape(nblk, 8, 8)
NWV = 8 if D == 768 else 4
full = aw[:, :NWV, 7].min(axis=1) > 0   # blocks whose waves all have query rows
print("per wave (full blocks only), ticks per tile:  wait+barrier / DMA issue / S / softmax / O   | block life")
L_ = 1 + C * 196; nt_ = (L_ + (64 if D == 192 else 32) - 1) // (64 if D == 192 else 32)
for wv in range(NWV):
    x = aw[full][:, wv, :]
    print(f"   wave {wv}: " + " / ".join(f"{x[:, k].mean() / nt_:7.0f}" for k in range(5)) + f"   | {x[:, 7].mean():9.0f}")
a = aw[:, 0, :]
a = a[a[:, 7] > 0]
L = 1 + C * 196; KVT = 64 if D == 192 else 32; ntile = (L + KVT - 1) // KVT
life = a[:, 7]
print(f"D={D} C={C} len={L} ({ntile} key tiles of {KVT}) blocks {len(a)}  kernel {us:.1f} us; block life {life.mean():.0f} ticks (min {life.min():.0f} max {life.max():.0f}); blocks per CU {len(a) * (2 if D == 192 else 1) / 256 / (3 if D == 192 else (2 if D == 384 else 1)):.1f} rounds")
names = ["wait + barrier (tiles 1..)", "LDS-DMA issue", "S^T MFMAs + K reads", "softmax", "O^T MFMAs + V reads", "block start -> tile 0 ready", "epilogue (stores issued)"]
for k in range(7):
    per = a[:, k].mean() / (ntile if k in (0, 1, 2, 3, 4) else 1)
    print(f"  {names[k]:30s} {a[:, k].mean():10.0f} ticks per block ({100 * a[:, k].sum() / life.sum():5.1f} %)   {per:8.0f} per " + ("tile" if k < 5 else "block"))
print(f"  sum of phases / life = {a[:, :7].sum() / life.sum():.3f}")
