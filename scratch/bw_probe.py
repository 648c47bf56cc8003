import torch
dev = torch.device('cuda:0')
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
n = 1 << 30   # 2 GiB of bf16
a = torch.empty(n, device=dev, dtype=torch.bfloat16); b = torch.empty(n, device=dev, dtype=torch.bfloat16)
us = t(lambda: a.fill_(1.0)); print(f"fill 2 GiB: {us:.0f} us  write {2 * n / us / 1e6:.2f} TB/s")
us = t(lambda: a.zero_()); print(f"zero 2 GiB: {us:.0f} us  write {2 * n / us / 1e6:.2f} TB/s")
us = t(lambda: b.copy_(a)); print(f"copy 2 GiB: {us:.0f} us  read+write {4 * n / us / 1e6:.2f} TB/s")
us = t(lambda: a.sum()); print(f"sum 2 GiB: {us:.0f} us  read {2 * n / us / 1e6:.2f} TB/s")
c = torch.empty(n // 8, device=dev, dtype=torch.bfloat16)
us = t(lambda: c.fill_(1.0)); print(f"fill 256 MiB: {us:.0f} us  write {2 * (n // 8) / us / 1e6:.2f} TB/s")
