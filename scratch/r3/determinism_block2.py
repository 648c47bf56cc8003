import math, os, sys, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from chadavit_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
D, FF, M = 192, 2048, int(os.environ.get("M", "603136"))
a = torch.randn((M, D), device=dev).bfloat16(); x = torch.randn((M, D), device=dev).bfloat16()
wo = (torch.randn((D, D), device=dev) / math.sqrt(D)).bfloat16(); w1 = (torch.randn((FF, D), device=dev) / math.sqrt(D)).bfloat16()
w2 = (torch.randn((D, FF), device=dev) / math.sqrt(FF)).bfloat16(); wq = (torch.randn((3 * D, D), device=dev) / math.sqrt(D)).bfloat16()
bo, b1, b2, bq = [torch.randn(n, device=dev) * 0.1 for n in (D, FF, D, 3 * D)]
lns = [((1 + 0.2 * torch.randn(D, device=dev)), 0.2 * torch.randn(D, device=dev), 1e-5) for _ in range(3)]
slab = torch.cat([w1.reshape(-1), w2.reshape(-1), wo.reshape(-1), wq.reshape(-1)])
pkq = torch.empty(ops.ffn_proj_packed_bytes(D, FF) // 2, device=dev, dtype=torch.bfloat16)
o1, o2 = w1.numel(), w1.numel() + w2.numel()
ops.ffn_pack_proj_batched(slab, pkq, torch.tensor([0, o1, o2, o2 + wo.numel(), 0], device=dev, dtype=torch.int64), 1, D, FF)
# reference QKV from the separate-output instance (hn written, then a plain GEMM)
_, x2r, hnr = ops.proj_ffn_ln_fwd(a, x, pkq, bo, lns[0], b1, b2, lns[1], ln_b=lns[2], want_x1=False, want_hn=True)
qref = ops.gemm_nt(hnr, wq, bias=bq)
for it in range(6):
    _, x2, _, qkv = ops.proj_ffn_ln_fwd(a, x, pkq, bo, lns[0], b1, b2, lns[1], ln_b=lns[2], want_x1=False, want_hn=False, qkv_bias=bq)
    torch.cuda.synchronize()
    bad = (qkv.float() - qref.float()).abs() > 0.25
    rows, cols = bad.nonzero(as_tuple=True)
    print(f"run {it}: x2 equal {torch.equal(x2, x2r)}; qkv elements off by > 0.25 vs GEMM(hn): {int(bad.sum())}; rows {rows.unique().numel()}; "
          f"row%128 hist {sorted(collections.Counter((rows % 128).tolist()).items())[:20]}; col//192 {sorted(collections.Counter((cols // 192).tolist()).items())}; "
          f"col%192//32 {sorted(collections.Counter(((cols % 192) // 32).tolist()).items())}; blocks {sorted(set((rows // 128).tolist()))[:10]}", flush=True)
