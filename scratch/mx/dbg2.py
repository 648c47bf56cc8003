import sys, torch, math, itertools
sys.path.insert(0, '.')
from chadavit_amd import ops
from tests.test_kernels_gpu import _mx_dequant
dev = torch.device('cuda')
torch.manual_seed(0)
M = N = K = 128
x = torch.randn(M, K).bfloat16().to(dev)
w = (torch.randn(N, K) / math.sqrt(K)).bfloat16().to(dev)
xq, xs = ops.mx8_quantize(x)
wq, ws = ops.mx8_quantize(w)
dx, dw = _mx_dequant(xq, xs).double().cpu(), _mx_dequant(wq, ws).double().cpu()
out = ops.gemm_nt_mx8(xq, xs, wq, ws).float().cpu().double()
ref = dx @ dw.t()
err = (out - ref).abs()
blk = err.view(8, 16, 8, 16).amax((1, 3))
print('max err per 16x16 block (rows m, cols n):'); print((blk > 0.05).int())
# unscaled products per 16-byte piece
vx = xq.cpu().view(torch.float8_e4m3fn).double(); vw = wq.cpu().view(torch.float8_e4m3fn).double()
for (m, n) in [(0, 0), (1, 0), (0, 1), (5, 7), (17, 3), (2, 2)]:
    parts = [(dx[m, 16*s:16*s+16] * dw[n, 16*s:16*s+16]).sum().item() for s in range(8)]
    print((m, n), 'out', round(out[m, n].item(), 4), 'ref', round(ref[m, n].item(), 4), 'slot partials', [round(p, 3) for p in parts])
    # try: out = sum_s x_slot[s] * w_slot[perm(s)]
    best = None
    for sw in range(8):
        val = sum((dx[m, 16*s:16*s+16] * dw[n, 16*(s ^ sw):16*(s ^ sw)+16]).sum().item() for s in range(8))
        if best is None or abs(val - out[m, n].item()) < best[0]:
            best = (abs(val - out[m, n].item()), sw)
    print('   best xor pairing', best)
    # scale mismatch hypothesis: unscaled per-block sums with various scale combos
    ub = [(vx[m, 32*b:32*b+32] * vw[n, 32*b:32*b+32]).sum().item() for b in range(4)]
    sx = [2.0 ** (int(xs[b, m]) - 127) for b in range(4)]; sw_ = [2.0 ** (int(ws[b, n]) - 127) for b in range(4)]
    print('   unscaled block sums', [round(u, 1) for u in ub], 'sx', sx, 'sw', sw_)
