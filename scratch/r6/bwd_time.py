"""Backward attention kernels one by one (dQ with fused delta: parts 3; dK/dV: parts 4) at the bench's shapes.
usage: python scratch/r6/bwd_time.py [zero]   (CHADAVIT_HIP_LIB + CHADAVIT_ALLOW_FOREIGN_LIB=1 select a side build)"""
import sys, os, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd._lib import lib
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
zero = 'zero' in sys.argv
def t(fn, reps=20, rounds=5):
    for _ in range(3): fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]
g = torch.Generator().manual_seed(0)
mixed = [int(x) for x in torch.randint(1, 11, (256,), generator=g)]
shapes = [("tiny global 2048x589", [3] * 2048, 196, 192), ("tiny mixed 512", mixed + mixed, 196, 192), ("small mixed global 256", mixed, 196, 384),
          ("tiny c1 4096x197", [1] * 4096, 196, 192), ("base 64x1961", [10] * 64, 196, 768)]
mk = torch.zeros if zero else torch.randn
for name, nch, p, D in shapes:
    H = 2
    rb = RaggedBatch(nch, p, dev)
    qkv = mk((rb.T, 3 * D), device=dev).to(bf)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    do = mk((rb.T, D), device=dev).to(bf); dq = torch.empty_like(qkv); dl = torch.empty((H, rb.T), device=dev)
    def call(parts):
        rc = lib().chadavit_attn_bwd_parts(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), dq.data_ptr(), dl.data_ptr(), rb.cu_seqlens.data_ptr(),
                                           rb.work.data_ptr(), rb.work.shape[0], rb.T, D, H, parts, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    call(7)
    tq = t(lambda: call(3)); tkv = t(lambda: call(4)); tq2 = t(lambda: call(3)); tkv2 = t(lambda: call(4))
    fl = 4.0 * sum(n * n for n in rb.lens) * D
    print(f"{name}: T={rb.T} dQ {tq:.1f} / {tq2:.1f} us ({fl / tq / 1e6:.0f} TF/s)  dK/dV {tkv:.1f} / {tkv2:.1f} us ({1.5 * fl / tkv / 1e6:.0f} TF/s)  pair {tq + tkv:.1f}", flush=True)
