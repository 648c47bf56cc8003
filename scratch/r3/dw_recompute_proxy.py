"""What would a weight-gradient kernel that RECOMPUTES H and dpre cost (the previous review's item 2)?

Its dataflow is the attention dK/dV kernel's at head width 192, term by term:
   token rows = queries, hidden units = keys;   x1 = Q, dz = dO, W1 = K, W2^T = V   (all [*, 192])
   pre = x1 W1^T = S;   dH = dz W2 = dP;   H = relu(pre + b1) ~ P;   dpre = dH * [pre > 0] ~ dS
   dW2^T[hid, d] += sum_rows H dz  = dV;      dW1[hid, d] += sum_rows dpre x1 = dK
-- four GEMM units of (rows x hidden x 192) each, two row tiles needed both row-wise (the recompute GEMMs) and transposed (the
row-reducing GEMMs), key-owning waves with their W rows in registers.  `attn_bwd_dkv_dma_kernel<192, 1>` IS that kernel with exp
instead of relu (and the exponentials were shown not to matter to it: profiles/r03b_attention_fwd.md).  So: time it on a batch
with the FFN's number of (row, hidden unit) pairs -- sequences of 2048 tokens (2048 "hidden units" each), one head of 192 --
and compare with what the recompute would REPLACE per block: two TN GEMMs + the H write of the forward + the dpre write of the
backward."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd._lib import lib
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
M, FF, D = 603136, 2048, 192
nseq = round(M / FF)                      # 294.5 -> 294 sequences of 2048 tokens = 602 112 rows
rb = RaggedBatch([1] * nseq, FF - 1, dev)  # 1 + 1 * 2047 = 2048 tokens each
assert rb.T == nseq * FF
qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
out = torch.randn((rb.T, D), device=dev).to(bf); dout = torch.randn((rb.T, D), device=dev).to(bf)
lse = torch.zeros((1, rb.T), device=dev); delta = torch.zeros((1, rb.T), device=dev)
dqkv = torch.empty_like(qkv)
L = lib()
def dkv():
    rc = L.chadavit_attn_bwd_parts(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(dout.data_ptr()),
                                   ctypes.c_void_p(lse.data_ptr()), ctypes.c_void_p(dqkv.data_ptr()), ctypes.c_void_p(delta.data_ptr()),
                                   ctypes.c_void_p(rb.cu_seqlens.data_ptr()), ctypes.c_void_p(rb.work.data_ptr()), ctypes.c_int(rb.n_work),
                                   ctypes.c_int(rb.T), ctypes.c_int(D), ctypes.c_int(1), ctypes.c_int(4), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
def t(fn, reps=10, rounds=3):
    for _ in range(2): fn()
    res = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(1e3 * e0.elapsed_time(e1) / reps)
    return sorted(res)[len(res) // 2]
us = t(dkv)
pairs = nseq * FF * FF
fl = 4 * 2.0 * pairs * D
print(f"dK/dV kernel, dh 192, {nseq} x {FF} tokens = {rb.T} rows x {FF} 'hidden units': {us:.1f} us  = {fl / us / 1e6:.0f} TFLOP/s over its four GEMM units"
      f" ({fl / us / 1e6 / 2500:.3f} of 2.5 PF)  -> scaled to M = {M}: {us * M / rb.T:.1f} us per block")
# what it would replace (same box): the two TN weight-gradient GEMMs, and the extra time of the training forward / the backward dX for
# writing H / dpre
a = torch.randn((M, FF), device=dev).to(bf); b = torch.randn((M, D), device=dev).to(bf)
c1 = torch.empty((FF, D), device=dev); c2 = torch.empty((D, FF), device=dev); cs1 = torch.empty(FF, device=dev); cs2 = torch.empty(D, device=dev)
ws = torch.empty(24 << 20, device=dev)
t1 = t(lambda: ops.gemm_tn(a, b, c1, colsum=cs1, workspace=ws))
t2 = t(lambda: ops.gemm_tn(b, a, c2, colsum=cs2, workspace=ws))
print(f"dW1 TN GEMM {t1:.1f} us, dW2 TN GEMM {t2:.1f} us (incl. their split-T reductions)")
del a, c1, c2
dz = torch.randn((M, D), device=dev).to(bf)
w1t = (torch.randn((D, FF), device=dev) / D ** .5).to(bf); w2t = (torch.randn((FF, D), device=dev) / FF ** .5).to(bf)
pkb = ops.ffn_pack(w2t, w1t)
bits = torch.randint(0, 256, (int(ops.relu_bits_buffer(M, FF, dev).numel()),), device=dev, dtype=torch.uint8)
dx = torch.empty((M, D), device=dev, dtype=bf); dp = torch.empty((M, FF), device=dev, dtype=bf)
tb1 = t(lambda: ops.ffn_bwd_dx(dz, pkb, bits, dx1=dx, dpre=dp))
tb0 = t(lambda: ops.ffn_bwd_dx(dz, pkb, bits, dx1=dx, dpre=None))
print(f"FFN backward dX: {tb1:.1f} us with dpre written, {tb0:.1f} us without  (-> the dpre write costs {tb1 - tb0:.1f} us)")
