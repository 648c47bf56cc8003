"""Backward attention on seeded inputs -> sha256 of (dqkv, delta) per case: run under two libraries and diff.
usage: python scratch/r6/bwd_dump.py  (CHADAVIT_HIP_LIB + CHADAVIT_ALLOW_FOREIGN_LIB=1 select a side build)"""
import sys, hashlib, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
mixed = [int(x) for x in torch.randint(1, 11, (64,), generator=g)]
edge = [1, 14, 15, 16, 30, 31, 32, 62, 63, 64, 95, 96, 126, 127, 128, 191, 192, 256]
cases = [([3] * 64, 196, 192), ([3] * 128, 36, 192), (mixed, 196, 192), (mixed, 196, 384), (mixed, 36, 384), ([1, 2, 3, 10, 7, 1, 1, 4, 5], 196, 192),
         ([1, 3, 2], 1, 192), ([2, 10, 5], 4, 192), (edge, 1, 192), (edge, 1, 384), ([10, 1, 3], 196, 768), (edge, 1, 768)]
for i, (nch, p, D) in enumerate(cases):
    rb = RaggedBatch(nch, p, dev)
    qkv = torch.randn((rb.T, 3 * D), generator=torch.Generator().manual_seed(100 + i)).to(bf).to(dev)
    do = torch.randn((rb.T, D), generator=torch.Generator().manual_seed(200 + i)).to(bf).to(dev)
    o, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
    delta = torch.empty((2, rb.T), device=dev)
    dqkv = ops.attn_bwd(qkv, o, do, lse, rb.cu_seqlens, rb.work, 2, delta=delta)
    torch.cuda.synchronize()
    h = hashlib.sha256(dqkv.cpu().view(torch.int16).numpy().tobytes() + delta.cpu().numpy().tobytes()).hexdigest()[:16]
    print(i, D, p, len(nch), h, bool(torch.isfinite(dqkv.float()).all()), flush=True)
