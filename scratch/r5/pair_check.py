"""dh 384 forward: default dispatch (first run, saves outputs) against CHADAVIT_ATTN_FWD_PAIR=1 (second run, compares bit for bit and times)."""
import os, sys, torch, random
sys.path.insert(0, '.')
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device('cuda:0'); bf = torch.bfloat16
def t(fn, reps=10, rounds=5):
    for _ in range(3): fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]
pair = os.environ.get("CHADAVIT_ATTN_FWD_PAIR", "0") != "0"
random.seed(3); D, H = 768, 2
for name, nch, p in (("global 64x1961", [10] * 64, 196), ("ragged 1-10", [random.randint(1, 10) for _ in range(96)], 196), ("local 256x361", [10] * 256, 36),
                     ("tiny ragged", [1, 10, 3, 2], 196), ("one short", [1], 36)):
    rb = RaggedBatch(nch, p, dev); torch.manual_seed(0)
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o, l = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    torch.cuda.synchronize()
    f = f"/tmp/pair_ref_{name.replace(' ', '_')}.pt"
    msg = ""
    if not pair: torch.save((o.cpu(), l.cpu()), f)
    else:
        o0, l0 = torch.load(f)
        msg = f"bit-identical out {bool((o.cpu().view(torch.int16) == o0.view(torch.int16)).all())} lse {bool((l.cpu() == l0).all())}  max|do| {float((o.cpu().float() - o0.float()).abs().max()):.4g}"
    print(f"{'pair' if pair else 'default':8s} {name:16s} T={rb.T:7d}  {t(lambda: ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H, out=o, lse=l)):8.1f} us  {msg}", flush=True)
