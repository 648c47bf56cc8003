import sys, torch
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
rb = RaggedBatch([10] * 64, 196, dev)   # cfg5: 64 global images x 10 channels -> 1961 tokens each
qkv = torch.randn((rb.T, 3 * 768), device=dev).to(bf)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
us = t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2, out=o, lse=lse))
fl = 4.0 * 64 * 1961 ** 2 * 768
print(f"attn_fwd dh=384 T={rb.T}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s")
do = torch.randn((rb.T, 768), device=dev).to(bf); dq_ = torch.empty_like(qkv); dl = torch.empty((2, rb.T), device=dev)
us = t(lambda: ops.attn_bwd(qkv, o, do, lse, rb.cu_seqlens, rb.work, 2, dqkv=dq_, delta=dl), n=5)
print(f"attn_bwd dh=384: {us:.1f} us  {10.0 * 64 * 1961 ** 2 * 768 / us / 1e6:.0f} TFLOP/s (algorithmic 5 GEMM units)")
from chadavit_amd._lib import lib
import ctypes
for parts, name in ((3, "delta + dQ"), (4, "dK/dV")):
    def run(parts=parts):
        rc = lib().chadavit_attn_bwd_parts(ops._ptr(qkv), ops._ptr(o), ops._ptr(do), ops._ptr(lse), ops._ptr(dq_), ops._ptr(dl), ops._ptr(rb.cu_seqlens), ops._ptr(rb.work),
                                           ctypes.c_int(rb.work.shape[0]), ctypes.c_int(rb.T), ctypes.c_int(768), ctypes.c_int(2), ctypes.c_int(parts), ops._stream())
        assert rc == 0, rc
    print(f"   {name}: {t(run, n=5):.1f} us")
