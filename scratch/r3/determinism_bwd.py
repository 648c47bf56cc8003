import math, os, sys, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from chadavit_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
D, FF = 192, 2048
for M in (26282, 26368, 1000):
    dz = torch.randn((M, D), device=dev).bfloat16()
    w1 = (torch.randn((FF, D), device=dev) / math.sqrt(D)).bfloat16(); w2 = (torch.randn((D, FF), device=dev) / math.sqrt(FF)).bfloat16()
    pkb = ops.ffn_pack(w2.t().contiguous(), w1.t().contiguous())
    n = int(ops.relu_bits_buffer(M, FF, dev).numel())
    bits = torch.randint(0, 256, (n,), device=dev, dtype=torch.uint8)
    ref = None
    for it in range(6):
        junk = torch.full((1 << 26,), float(it), device=dev)
        dpre = torch.full((M, FF), 7.0, device=dev, dtype=torch.bfloat16)
        dx = ops.ffn_bwd_dx(dz, pkb, bits, dpre=dpre)
        torch.cuda.synchronize()
        if ref is None:
            ref = (dx.clone(), dpre.clone()); continue
        bad = (dpre != ref[1]).nonzero()
        badx = (dx != ref[0]).nonzero()
        print(f"M={M} run {it}: dpre differing elements {bad.shape[0]} rows {sorted(set(bad[:,0].tolist()))[:8]} cols {sorted(set(bad[:,1].tolist()))[:12]}; dx differing {badx.shape[0]}; untouched dpre (==7) {int((dpre == 7.0).sum())}", flush=True)
        del junk
