import os, sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
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
for T in (1206272, 603136, 283868):
    for I, J in ((2048, 192), (192, 2048), (576, 192), (1152, 384), (2048, 384), (384, 2048)):
        a = torch.randn((T, I), device=dev).to(bf); b = torch.randn((T, J), device=dev).to(bf)
        ws = torch.empty(48 << 20, device=dev)
        res = {}
        line = f"T={T} {I}x{J}:"
        for m in ("0", "1", "0", "1"):
            os.environ['CHADA_TN_OCC3'] = m
            c = torch.empty((I, J), device=dev); cs = torch.empty(I, device=dev)
            us = t(lambda: ops.gemm_tn(a, b, c, colsum=cs, workspace=ws))
            res[m] = c.clone()
            gb = 2.0 * T * (I + J) / 1e9
            line += f"  occ3={m}: {us:7.1f} us ({gb / us * 1e3:5.2f} TB/s)"
        ref = (a[:65536].float().t() @ b[:65536].float())
        rel = float((res["1"] - res["0"]).norm() / res["0"].norm())
        print(line + f"  rel diff {rel:.2e}", flush=True)
        del a, b
