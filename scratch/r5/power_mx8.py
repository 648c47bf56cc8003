"""Board power / shader clock while the MX-fp8 GEMM runs back to back at cfg5's shapes (set-up of scratch/r3/mx8_bench.py), and the bf16 GEMMs beside it."""
import sys, time, threading, glob, torch
sys.path.insert(0, ".")
from chadavit_amd import ops
dev = torch.device("cuda:0"); M = 125504
hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
def sample():
    best = (0, 0)
    for h in hw:
        try: v = (int(open(h + "/power1_input").read()) / 1e6, int(open(h + "/freq1_input").read()) / 1e6)
        except Exception: continue
        if v[0] > best[0]: best = v
    return best
def run(label, fn, flops):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    stop = False; rows = []
    def th():
        while not stop: rows.append(sample()); time.sleep(0.05)
    t = threading.Thread(target=th); t.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); n = 0; e0.record()
    while time.time() - t0 < 2.5:
        for _ in range(20): fn()
        n += 20; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop = True; t.join()
    half = rows[len(rows) // 2:]; us = 1e3 * e0.elapsed_time(e1) / n
    print(f"{label:44s} {us:8.1f} us {flops / us * 1e-6:7.0f} TF/s  {sum(r[0] for r in half) / len(half):7.1f} W  {sum(r[1] for r in half) / len(half):7.1f} MHz", flush=True)
torch.manual_seed(0)
for N, K, epi, q in ((2304, 768, 0, False), (2048, 768, 1, True), (768, 2048, 3, False)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    xq, xs = ops.mx8_quantize(x); wq, ws = ops.mx8_quantize(w)
    aux = torch.randn(M, N, device=dev, dtype=torch.bfloat16) if epi == 3 else None
    bias = torch.zeros(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run(f"gemm_nt_mx8 N {N} K {K} epi {epi} q {int(q)}", lambda: ops.gemm_nt_mx8(xq, xs, wq, ws, bias=bias, epilogue=epi, aux=aux, out=out, emit_q=q), 2.0 * M * N * K)
    if N == 2304:
        o2 = torch.empty(M, K, device=dev, dtype=torch.bfloat16); dq = torch.randn(M, N, device=dev, dtype=torch.bfloat16); wt = w.t().contiguous()
        run("gemm_nt bf16 (dqkv -> dh) 768 x 2304", lambda: ops.gemm_nt(dq, wt, out=o2), 2.0 * M * N * K)
        c = torch.empty((N, K), device=dev); cs = torch.empty(N, device=dev); wsz = torch.empty(24 << 20, device=dev)
        run("gemm_tn bf16 (dW_qkv) 2304 x 768", lambda: ops.gemm_tn(dq, x, c, colsum=cs, workspace=wsz), 2.0 * M * N * K)
    del x, w, xq, xs, wq, ws, aux, out
