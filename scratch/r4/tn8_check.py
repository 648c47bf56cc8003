"""gemm_tn on the 8-wave 256 x 192 / 192 x 256 tiles vs the 4-wave tiles (CHADA_TN_NO8=1 in a child process) and vs fp32 torch."""
import os, subprocess, sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
tag = "4-wave" if os.environ.get("CHADA_TN_NO8") else "8-wave"
def t(fn, reps=10, rounds=3):
    for _ in range(3): fn()
    res = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(1e3 * e0.elapsed_time(e1) / reps)
    return sorted(res)[1]
for T in (40000, 77777, 603136, 1206272):
    for I, J in ((2048, 192), (192, 2048)):
        g = torch.Generator(device=dev).manual_seed(T + I)
        a = torch.randn((T, I), device=dev, generator=g).to(bf); b = torch.randn((T, J), device=dev, generator=g).to(bf)
        c = torch.empty((I, J), device=dev); cs = torch.empty(I, device=dev); ws = torch.empty(48 << 20, device=dev)
        ops.gemm_tn(a, b, c, colsum=cs, workspace=ws)
        if T <= 77777:
            ref = a.float().T @ b.float(); rcs = a.float().sum(0)
            err = ((c - ref).abs().max() / ref.abs().max()).item(); ecs = ((cs - rcs).abs().max() / rcs.abs().max()).item()
            c2 = torch.empty_like(c); ops.gemm_tn(a, b, c2, colsum=cs, workspace=ws)
            print(f"{tag} T={T} {I}x{J}: rel err {err:.2e} colsum {ecs:.2e} deterministic {torch.equal(c, c2)}", flush=True)
        else:
            us = t(lambda: ops.gemm_tn(a, b, c, colsum=cs, workspace=ws))
            print(f"{tag} T={T} {I}x{J}: {us:8.1f} us  {2*T*(I+J)/us/1e6:6.2f} TB/s", flush=True)
if tag == "8-wave" and "no-child" not in sys.argv:
    subprocess.run([sys.executable] + sys.argv + ["no-child"], env=dict(os.environ, CHADA_TN_NO8="1"))
