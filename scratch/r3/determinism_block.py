"""Determinism of the whole-block kernel's instances: the same launch repeated, outputs compared bit for bit."""
import math, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from chadavit_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
D, FF = int(os.environ.get("D", "192")), 2048
for M in (26282, 603136, 1024):
    a = torch.randn((M, D), device=dev).bfloat16(); x = torch.randn((M, D), device=dev).bfloat16()
    wo = (torch.randn((D, D), device=dev) / math.sqrt(D)).bfloat16(); w1 = (torch.randn((FF, D), device=dev) / math.sqrt(D)).bfloat16()
    w2 = (torch.randn((D, FF), device=dev) / math.sqrt(FF)).bfloat16(); wq = (torch.randn((3 * D, D), device=dev) / math.sqrt(D)).bfloat16()
    bo, b1, b2, bq = [torch.randn(n, device=dev) * 0.1 for n in (D, FF, D, 3 * D)]
    lns = [((1 + 0.2 * torch.randn(D, device=dev)), 0.2 * torch.randn(D, device=dev), 1e-5) for _ in range(3)]
    slab = torch.cat([w1.reshape(-1), w2.reshape(-1), wo.reshape(-1), wq.reshape(-1)])
    pkq = torch.empty(ops.ffn_proj_packed_bytes(D, FF) // 2, device=dev, dtype=torch.bfloat16)
    o1, o2 = w1.numel(), w1.numel() + w2.numel()
    ops.ffn_pack_proj_batched(slab, pkq, torch.tensor([0, o1, o2, o2 + wo.numel(), 0], device=dev, dtype=torch.int64), 1, D, FF)
    for label, kw in (("no-grad + qkv (teacher)", dict(want_x1=False, want_hn=False, qkv_bias=bq)),
                      ("no-grad, no qkv", dict(want_x1=False, want_hn=True))):
        ref = None
        bad = {}
        for it in range(10):
            r = ops.proj_ffn_ln_fwd(a, x, pkq, bo, lns[0], b1, b2, lns[1], ln_b=lns[2], **kw)
            torch.cuda.synchronize()
            outs = {n: t.clone() for n, t in zip(("x1", "x2", "hn", "qkv"), r) if t is not None}
            if ref is None:
                ref = outs; continue
            for n, t in outs.items():
                if not torch.equal(t, ref[n]):
                    rows = (t != ref[n]).any(1).nonzero().flatten()
                    bad.setdefault(n, []).append((int(rows.numel()), rows[:6].tolist(), sorted(set((rows % 128).tolist()))[:12]))
        print(f"M={M} {label}: outputs {sorted(ref)}; nondeterministic: { {n: v[:3] for n, v in bad.items()} }", flush=True)
