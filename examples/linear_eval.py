"""Linear (or fine-tune) evaluation of a pretrained ChAda-ViT on the HIP engine -- the reference's `main_linear.py` without Lightning:

    python examples/linear_eval.py --ckpt pretrain.ckpt [--finetune] [--epochs 2]

Synthetic labelled data stands in for a dataset: images of 3 channels whose class decides the mean of each channel."""
import argparse
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

from chadavit_amd.backbones import vit_channels
from chadavit_amd.data.channels_strategies import one_channel_collate_fn
from chadavit_amd.methods.linear import LinearModel
from chadavit_amd.trainer import Trainer
from chadavit_amd.utils.checkpoint import load_backbone
from chadavit_amd.utils.misc import AttrDict


def make_split(n, n_cls, seed, size=224):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        y = i % n_cls
        means = torch.tensor([((y >> k) & 1) * 0.8 - 0.4 for k in range(3)]).view(3, 1, 1)
        out.append((torch.randn(3, size, size, generator=g) * 0.5 + means, y))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=None, help="Lightning-style checkpoint of a DINO run (examples/pretrain.py --out, or the reference's)")
    ap.add_argument("--embed-dim", type=int, default=192)
    ap.add_argument("--finetune", action="store_true")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n_cls = 8
    backbone = vit_channels("dino", patch_size=16, embed_dim=a.embed_dim, return_all_tokens=False, max_number_channels=10)
    if a.ckpt:
        load_backbone(backbone, a.ckpt)                      # main_linear.py:103-110 key rewrite
    cfg = AttrDict({"backbone": {"name": "vit_channels", "kwargs": {"embed_dim": a.embed_dim, "patch_size": 16, "return_all_tokens": False,
                                                                    "max_number_channels": 10}},
                    "data": {"dataset": "synthetic", "num_classes": n_cls, "img_channels": 3, "max_img_channels": 10},
                    "channels_strategy": "multi_channels", "mixed_channels": False, "max_epochs": a.epochs, "finetune": a.finetune,
                    "optimizer": {"name": "adamw", "batch_size": a.batch, "lr": 1e-4 if a.finetune else 1e-2, "weight_decay": 0.0},
                    "scheduler": {"name": "warmup_cosine", "warmup_epochs": 1 if a.epochs > 1 else 0}})
    model = LinearModel(backbone, cfg).to(dev)
    train, val = make_split(8 * a.batch, n_cls, 1), make_split(2 * a.batch, n_cls, 2)
    batches = lambda data: [one_channel_collate_fn(data[i:i + a.batch]) for i in range(0, len(data), a.batch)]
    to_dev = lambda b: (b[0].to(dev), b[1].to(dev), b[2])
    trainer = Trainer(max_epochs=a.epochs, steps_per_epoch=len(train) // a.batch).attach(model)
    for epoch in range(a.epochs):
        trainer.current_epoch = epoch
        model.train()
        for i, b in enumerate(batches(train)):
            trainer.train_step(to_dev(b), i)
        model.eval()
        for i, b in enumerate(batches(val)):
            model.validation_step(to_dev(b), i)
        model.on_validation_epoch_end()
        m = model.logged_metrics()
        print(f"epoch {epoch}: train_loss {m['train_loss']:.4f} val_loss {m['val_loss']:.4f} val_acc1 {m['val_acc1']:.1f} val_acc5 {m['val_acc5']:.1f}", flush=True)


if __name__ == "__main__":
    main()
