import sys, torch, math
sys.path.insert(0, '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
M, D, FF = 301568, 192, 2048
x = torch.randn((M, D), device=dev).to(bf)
w1 = (torch.randn((FF, D), device=dev) / D ** .5).to(bf); w2 = (torch.randn((D, FF), device=dev) / FF ** .5).to(bf)
b1, b2 = torch.zeros(FF, device=dev), torch.zeros(D, device=dev)
pk = ops.ffn_pack(w1, w2)
o = torch.empty((M, D), device=dev, dtype=bf)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
for rpw in (32, 64, 32, 64):
    t = timeit(lambda: ops.ffn_fwd(x, pk, b1, b2, resid=x, out=o, rows_per_wave=rpw))
    print(f"rows_per_wave {rpw}: {t:.1f} us = {4.0 * M * D * FF / t / 1e6:.0f} TF/s")
