"""ffn_bwd_dx (the FFN's backward dX from the ReLU bit records) at the headline shape: time per launch."""
import sys, torch
sys.path.insert(0, '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
T, D, FF = 1206272, 192, 2048
dz = torch.randn((T, D), device=dev).to(bf)
w1t = (torch.randn((D, FF), device=dev) / D ** .5).to(bf); w2t = (torch.randn((FF, D), device=dev) / FF ** .5).to(bf)
pkb = ops.ffn_pack(w2t, w1t)
rb_ = torch.randint(0, 256, (int(ops.relu_bits_buffer(T, FF, dev).numel()),), device=dev, dtype=torch.uint8)
dx = torch.empty((T, D), device=dev, dtype=bf); dp = torch.empty((T, FF), device=dev, dtype=bf)
fn = lambda: ops.ffn_bwd_dx(dz, pkb, rb_, dx1=dx, dpre=dp)
for _ in range(3): fn()
out = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize(); out.append(1e3 * e0.elapsed_time(e1) / 10)
import hashlib
print("ffn_bwd_dx us:", " ".join(f"{x:.1f}" for x in out), "sha", hashlib.sha256(dx.cpu().view(torch.int16).numpy().tobytes() + dp[:65536].cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12])
