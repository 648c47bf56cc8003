"""What does the reference's arithmetic cost on this GPU when it is run the way the reference runs it -- PyTorch eager ops, 10-channel padding +
key mask (chada_vit.py:226-239), explicit softmax(QK^T) (nn.MultiheadAttention with need_weights=True takes the unfused path)?

The reference itself cannot travel to the GPU box; oracle/chada_ref.py is its plain-torch restatement (pinned to it by tests/golden), written
with device-agnostic torch ops, so `torch.set_default_device("cuda")` runs the same restatement on the MI355X.  cfg2 shape: Tiny/16, 3 channels,
2 global + 8 local crops, head 2048/256/4096, training_step (student fwd + bwd, local-crop fwd, teacher fwd, loss) -- optimiser / EMA not
included (they are small).  Variants: padded (what the reference executes) / ragged (padding-free restatement), fp32 / bf16 autocast."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

torch.set_default_device("cuda")
from oracle import chada_ref as R
from oracle import procedural as P
from tests.golden_util import build_sd

D, PROT = 192, 4096
for B in (32, 64):
    imgs = P.make_images([3] * B, [224, 224] + [96] * 8, seed=1)
    crops, _, ncl = R.collate(imgs)
    crops = [c.cuda() for c in crops]
    sd = {k: v.cuda() for k, v in build_sd(D, PROT).items()}
    for variant in ("padded", "ragged"):
        for prec in ("fp32", "bf16-autocast"):
            def step():
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(prec != "fp32")):
                    return R.training_step(sd, crops, ncl, 2, 0.04, padded=(variant == "padded"))
            try:
                for _ in range(2):
                    loss = step()[0]
                torch.cuda.synchronize(); t0 = time.perf_counter()
                n = 3
                for _ in range(n):
                    loss = step()[0]
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                print(f"B={B:3d} {variant:7s} {prec:14s} {1e3 * dt:9.1f} ms/step  {B / dt:8.1f} images/s  loss {float(loss):.4f}  "
                      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
            except Exception as e:  # noqa: BLE001
                print(f"B={B} {variant} {prec}: failed: {repr(e)[:200]}", flush=True)
            torch.cuda.reset_peak_memory_stats()
