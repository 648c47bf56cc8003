"""Hunt for reads of uninitialised / out-of-range memory: the same training step repeated in one process with the caching allocator's
free blocks POISONED between runs (large finite values, then NaN); every gradient tensor must reproduce the first run bit for bit
(or to fp32 summation-order noise where a kernel's split count depends on nothing but shapes: it does not).
    python scratch/r3/poison_fuzz.py [golden name]"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import torch

from oracle import procedural as P
from tests.golden_util import build_sd
from tests.test_model_gpu import _cfg, GOLDEN


def poison(dev, value, gb=40):
    """Fill ~gb GB of fresh allocations with `value`, then free them (the blocks go back to the caching allocator's pool)."""
    bufs = []
    for size_mb in (2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 2, 1):
        n = max(1, int(gb * 1024 / 12 / size_mb))
        for _ in range(min(n, 24)):
            bufs.append(torch.full((size_mb * 1024 * 1024 // 2,), value, device=dev, dtype=torch.bfloat16))
    torch.cuda.synchronize()
    del bufs


def main():
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    name = sys.argv[1] if len(sys.argv) > 1 else "step_tiny_fused_rows"
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    D, PR = int(g["D"]), int(g["P"])
    nch, sizes = [int(c) for c in g["nch"]], [int(s) for s in g["sizes"]]
    n_large, epoch = int(g["n_large"]), int(g["epoch"])
    crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, sizes, seed=7))
    ref = None
    for it, val in enumerate([None, 3.0e4, float("nan"), -3.0e4, float("nan")]):
        if val is not None:
            poison(dev, val)
        use_bn = bool(int(g["use_bn"])) if "use_bn" in g.files else False
        cfg = _cfg(D, PR, n_large, len(sizes) - n_large, clip_grad=float(g["clip_grad"]), lr=float(g["lr"]), wd=float(g["wd"]),
                   base_tau=float(g["base_tau"]), use_bn_in_head=use_bn)
        if os.environ.get("FP8"):
            cfg.backbone.kwargs.weight_dtype = "fp8"
        model = DINO(cfg)
        model.load_state_dict(build_sd(D, PR, use_bn=use_bn))
        model = model.to(dev)
        if os.environ.get("FUSED0"):   # the whole-block kernels at this (small) row count too
            model.backbone.fused_min_rows = model.momentum_backbone.fused_min_rows = 0
        if os.environ.get("OVERLAP"):
            model.overlap_streams = True
            model.backbone.dw_side_stream = True
        tr = Trainer(max_epochs=10, steps_per_epoch=10)
        tr.current_epoch = epoch
        tr.attach(model)
        model.current_epoch = epoch
        model.on_train_epoch_start()
        loss = model.training_step(([c.to(dev) for c in crops], labels.to(dev), ncl), 1)
        loss.backward()
        model.on_after_backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        if ref is None:
            ref = (float(loss), grads)
            print(f"run 0: loss {float(loss):.6f}, {len(grads)} gradient tensors", flush=True)
            continue
        bad = []
        for n, gt in grads.items():
            r = ref[1][n]
            if not torch.equal(gt, r):
                d = float((gt - r).norm() / (r.norm() + 1e-30)) if torch.isfinite(gt).all() else float("inf")
                bad.append((d, n))
        bad.sort(reverse=True)
        print(f"run {it} (pool poisoned with {val}): loss {float(loss):.6f} (ref {ref[0]:.6f}); {len(bad)} tensors differ; worst: {bad[:6]}", flush=True)
        del model, tr, grads


if __name__ == "__main__":
    main()
