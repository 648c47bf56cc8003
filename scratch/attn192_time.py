import sys, torch, ctypes
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else '.')
from chadavit_amd import ops
from chadavit_amd._lib import lib
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
import random
random.seed(0)
nch = [random.randint(1, 10) for _ in range(256)]       # cfg3: 256 global crops with 1-10 channels
rb = RaggedBatch(nch, 196, dev)
qkv = torch.randn((rb.T, 3 * 384), device=dev).to(bf)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
do = torch.randn((rb.T, 384), device=dev).to(bf); dq_ = torch.empty_like(qkv); dl = torch.empty((2, rb.T), device=dev)
for parts, name in ((3, "delta + dQ"), (4, "dK/dV")):
    def run(parts=parts):
        rc = lib().chadavit_attn_bwd_parts(ops._ptr(qkv), ops._ptr(o), ops._ptr(do), ops._ptr(lse), ops._ptr(dq_), ops._ptr(dl), ops._ptr(rb.cu_seqlens), ops._ptr(rb.work),
                                           ctypes.c_int(rb.work.shape[0]), ctypes.c_int(rb.T), ctypes.c_int(384), ctypes.c_int(2), ctypes.c_int(parts), ops._stream())
        assert rc == 0, rc
    print(f"dh=192 T={rb.T} {name}: {t(run):.1f} us")
