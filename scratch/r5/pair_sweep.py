"""dh 384 attention at arbitrary sequence lengths: first run (CHADAVIT_ATTN_FWD_PAIR=0 CHADAVIT_ATTN_DQ_RM=0: the kernels of round 4) saves outputs, second run (defaults)
compares bit for bit and repeats every launch three times (determinism).  Lengths cover every tile-count parity and remainder class."""
import os, sys, torch, random
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
old = os.environ.get("CHADAVIT_ATTN_FWD_PAIR") == "0"
random.seed(11); D, H = 768, 2
# RaggedBatch takes channel counts and patches per channel: length = 1 + C * p; cover many lengths through (C, p) pairs
cases = []
for p in (1, 2, 3, 4, 7, 9, 16, 25, 31, 33, 36, 49, 64, 100, 121, 144, 169, 196):
    cases.append(([random.randint(1, 10) for _ in range(24)], p))
bad = 0
for i, (nch, p) in enumerate(cases):
    rb = RaggedBatch(nch, p, dev)
    g = torch.Generator(device="cpu").manual_seed(100 + i)
    qkv = torch.randn((rb.T, 3 * D), generator=g).bfloat16().to(dev); do = torch.randn((rb.T, D), generator=g).bfloat16().to(dev)
    outs = []
    for rep in range(1 if old else 3):
        o, l = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
        delta = torch.empty((H, rb.T), device=dev)
        dqkv = ops.attn_bwd(qkv, o, do, l, rb.cu_seqlens, rb.work, H, delta=delta)
        outs.append((o.cpu(), l.cpu(), dqkv.cpu(), delta.cpu()))
    f = f"/tmp/pair_sweep_{i}.pt"
    if old: torch.save(outs[0], f)
    else:
        ref = torch.load(f)
        for rep, t in enumerate(outs):
            ok = all(torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b) for a, b in zip(t, ref))
            if not ok: bad += 1; print("MISMATCH", i, p, nch, rep)
    print(f"case {i:2d} p={p:3d} lengths {sorted(set(1 + c * p for c in nch))[:6]}.. T={rb.T}", "saved" if old else "ok", flush=True)
print("mismatches:", bad)
