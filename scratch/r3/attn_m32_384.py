"""attention_m32 forward at dh = 384 (Base) vs the 16x16x32 kernel (eight waves x 16 rows): correctness + timing."""
import ctypes, sys, torch
sys.path.insert(0, '.')
exec(open('scratch/r3/attn_m32.py').read().split("import os\nvs96")[0])   # fwd / ref / check / t helpers
for name, nch in (("base 10ch x3", [10] * 3), ("base mixed", [1, 2, 5, 10, 3]), ("base boundaries", None)):
    if nch is None:
        check(name, [1, 14, 15, 16, 30, 31, 32, 62, 63, 64, 95, 96, 126, 127, 128, 191, 192, 256], 1, 768, 2, [0, 5, 1])
    else:
        check(name, nch, 196, 768, 2, [0, 5, 1])
check("base re-run", [3] * 3, 196, 768, 2, [0, 5], spike=(300, 500, 1.5))
for name, nch, p in (("base 10ch 64 img", [10] * 64, 196), ("base local 10ch 256", [10] * 256, 36)):
    rb = RaggedBatch(nch, p, dev)
    D, H = 768, 2
    qkv = torch.randn((rb.T, 3 * D), device=dev).to(bf)
    o = torch.empty((rb.T, D), device=dev, dtype=bf); lse = torch.empty((H, rb.T), device=dev)
    fl = 4.0 * sum(n * n for n in rb.lens) * D
    for rep in range(2):
        for v in (0, 5):
            us = t(lambda: fwd(qkv, rb, H, v, out=o, lse=lse))
            print(f"{name:24s} T={rb.T} variant {v}: {us:8.1f} us  {fl/us/1e6:7.0f} TF/s  ({fl/us/1e6/2500:.3f} of 2.5 PF)", flush=True)
