import sys, torch, math
sys.path.insert(0, '.')
from chadavit_amd import ops
from tests.test_kernels_gpu import _mx_dequant
dev = torch.device('cuda')
torch.manual_seed(0)
M = N = K = 128
def run(x, w, tag):
    xq, xs = ops.mx8_quantize(x); wq, ws = ops.mx8_quantize(w)
    ref = _mx_dequant(xq, xs) @ _mx_dequant(wq, ws).t()
    out = ops.gemm_nt_mx8(xq, xs, wq, ws).float()
    print(tag, 'max err', float((out - ref).abs().max()), 'ref max', float(ref.abs().max()), 'xs uniq', xs.unique().tolist(), 'ws uniq', ws.unique().tolist())
    return out, ref, xs, ws
sgn = lambda s: torch.where(torch.rand(s) > 0.5, 1.0, -1.0)
x = ((1 + torch.rand(M, K)) * sgn((M, K))).bfloat16().to(dev); x[:, ::32] = 1.9
w = ((1 + torch.rand(N, K)) * sgn((N, K))).bfloat16().to(dev); w[:, ::32] = 1.9
run(x, w, 'uniform scales:')
x2 = x.clone(); x2[5] *= 4   # one row of X with another scale
out, ref, xs, ws = run(x2, w, 'row 5 of X x4:')
print(' err rows', ((out - ref).abs() > 0.5).any(1).nonzero().flatten().tolist()[:20])
w2 = w.clone(); w2[7] *= 4
out, ref, xs, ws = run(x, w2, 'row 7 of W x4:')
print(' err cols', ((out - ref).abs() > 0.5).any(0).nonzero().flatten().tolist()[:20])
x3 = x.clone(); x3[:, 32:64] *= 4   # one k-block of X
out, ref, xs, ws = run(x3, w, 'k-block 1 of X x4:')
print(' ratio out/ref sample', (out[:2, :4] / ref[:2, :4]).tolist())
