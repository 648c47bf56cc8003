"""Fused FFN: correctness vs the two-GEMM path and timing."""
import sys, torch
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else '.')
from chadavit_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
torch.manual_seed(0)
D, FF = 192, 2048
def run(M, rpw, write_h):
    x = torch.randn((M, D), device=dev).to(bf)
    w1 = (torch.randn((FF, D), device=dev) / D ** .5).to(bf); b1 = torch.randn(FF, device=dev) * 0.1
    w2 = (torch.randn((D, FF), device=dev) / FF ** .5).to(bf); b2 = torch.randn(D, device=dev) * 0.1
    pk = ops.ffn_pack(w1, w2)
    h_ref = ops.gemm_nt(x, w1, bias=b1, epilogue=1)
    o_ref = ops.gemm_nt(h_ref, w2, bias=b2, epilogue=3, aux=x)
    h = torch.empty((M, FF), device=dev, dtype=bf) if write_h else None
    o = ops.ffn_fwd(x, pk, b1, b2, resid=x, h=h, rows_per_wave=rpw)
    torch.cuda.synchronize()
    eo = (o.float() - o_ref.float()).abs().max().item()
    eh = (h.float() - h_ref.float()).abs().max().item() if write_h else -1
    # fp64 reference
    hr = torch.relu(x.double() @ w1.double().T + b1.double()).to(bf).double()
    orf = hr @ w2.double().T + b2.double() + x.double()
    e64 = (o.double() - orf).abs().max().item(); e64r = (o_ref.double() - orf).abs().max().item()
    def t(fn, reps=10):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps
    tf = t(lambda: ops.ffn_fwd(x, pk, b1, b2, resid=x, out=o, h=h, rows_per_wave=rpw))
    t2 = t(lambda: (ops.gemm_nt(x, w1, out=h_ref, bias=b1, epilogue=1), ops.gemm_nt(h_ref, w2, out=o_ref, bias=b2, epilogue=3, aux=x)))
    fl = 4.0 * M * D * FF
    print(f"M={M} rpw={rpw} H={write_h}: max|o-o2gemm|={eo:.4f} max|h-h2|={eh:.4f} err64 fused={e64:.4f} two={e64r:.4f} | fused {tf:.1f} us ({fl/tf/1e6:.0f} TF/s)  two-GEMM {t2:.1f} us", flush=True)
for M in (1000, 301568, 223232):
    for rpw in (32, 64):
        for wh in (False, True):
            run(M, rpw, wh)
