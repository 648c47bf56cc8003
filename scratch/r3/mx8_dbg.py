"""Phase cycle counts of the MX-fp8 GEMM (instrumented build: CHADAVIT_HIP_LIB=.../libchadavit_hip_dbg.so), block 0, consumer wave 0 + producer wave 8."""
import sys, ctypes, torch
sys.path.insert(0, ".")
from chadavit_amd import ops
from chadavit_amd._lib import lib
dev = torch.device("cuda:0")
M = 125504
def run(N, K, epi, emit_q):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    xq, xs = ops.mx8_quantize(x); wq, ws = ops.mx8_quantize(w)
    aux = torch.randn(M, N, device=dev, dtype=torch.bfloat16) if epi == 3 else None
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_nt_mx8(xq, xs, wq, ws, bias=bias, epilogue=epi, aux=aux, out=out, emit_q=emit_q)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    rc = lib().chadavit_mx8_dbg_read(buf)
    v = list(buf)
    Q = v[13]; tiles = Q // (K // 128)
    print(f"N {N} K {K} epi {epi} q {int(emit_q)}: Q {Q} tiles {tiles}")
    print(f"  consumer per k-step: lgkm wait {v[0]/Q:7.0f}  barrier {v[1]/Q:7.0f}  read+mfma issue {v[2]/Q:7.0f}  deferred stores {v[3]/Q:7.0f} | per tile: epilogue {v[4]/tiles:7.0f} (mfma drain {v[7]/tiles:7.0f}, store issue {v[6]/tiles:7.0f}) | total {v[5]} cycles = {v[5]/tiles:.0f} per tile")
run(2304, 768, 0, False)
run(768, 768, 3, False)
run(2048, 768, 1, True)
run(768, 2048, 3, False)
