"""Weight-gradient TN GEMM at the headline's rows: time, and the result against fp32 torch on a slice (C and the column sums)."""
import sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def t(fn, n=10, rounds=5):
    for _ in range(3): fn()
    out = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(out)[len(out) // 2]
ws = torch.empty(24 << 20, device=dev)
for T in (1206272, 33000):
    for I, J in ((2048, 192), (192, 2048), (576, 192), (2048, 384), (384, 2048)):
        if T > 100000 and I * J > 2048 * 192: continue
        a = torch.randn((T, I), device=dev).to(bf); b = torch.randn((T, J), device=dev).to(bf)
        c = torch.empty((I, J), device=dev); cs = torch.empty(I, device=dev)
        us = t(lambda: ops.gemm_tn(a, b, c, colsum=cs, workspace=ws))
        n = min(T, 33000)
        ref = a[:n].double().t() @ b[:n].double(); refcs = a[:n].double().sum(0)
        if T == n:
            err = float((c.double() - ref).abs().max() / ref.abs().max()); ecs = float((cs.double() - refcs).abs().max() / refcs.abs().max())
        else:
            err = ecs = float('nan')
        print(f"gemm_tn T={T} {I}x{J}: {us:.1f} us  {2.0 * T * (I + J) / us / 1e6:.2f} TB/s  rel err C {err:.2e} colsum {ecs:.2e}", flush=True)
