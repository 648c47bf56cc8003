import sys, torch, math
sys.path.insert(0, '.')
from chadavit_amd import ops
from tests.test_kernels_gpu import _mx_dequant
dev = torch.device('cuda')
torch.manual_seed(0)
for (M, N, K) in [(128, 128, 128), (128, 128, 256), (256, 256, 128)]:
    x = torch.randn(M, K).bfloat16().to(dev)
    w = (torch.randn(N, K) / math.sqrt(K)).bfloat16().to(dev)
    xq, xs = ops.mx8_quantize(x)
    wq, ws = ops.mx8_quantize(w)
    ref = _mx_dequant(xq, xs) @ _mx_dequant(wq, ws).t()
    out = ops.gemm_nt_mx8(xq, xs, wq, ws).float()
    err = (out - ref).abs()
    bad = err > 2e-2 + 1e-2 * ref.abs()
    print(M, N, K, 'bad', int(bad.sum()), 'of', bad.numel(), 'max err', float(err.max()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten().tolist()
        cols = bad.any(0).nonzero().flatten().tolist()
        print(' bad rows', rows[:40], len(rows)); print(' bad cols', cols[:40], len(cols))
        # is out a permutation of ref?  match out rows to ref rows
        for r in rows[:4]:
            d = (ref - out[r][None]).abs().sum(1)
            print('  out row', r, 'closest ref row', int(d.argmin()), float(d.min()))
        outT = out.t(); refT = ref.t()
        for c in cols[:4]:
            d = (refT - outT[c][None]).abs().sum(1)
            print('  out col', c, 'closest ref col', int(d.argmin()), float(d.min()))
