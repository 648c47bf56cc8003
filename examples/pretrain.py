"""DINO pretraining of a ChAda-ViT on the HIP engine -- what the reference's `main_pretrain.py` does, without Lightning:

    python examples/pretrain.py [--data /path/to/IDRCell100k-format-dir] [--embed-dim 192] [--batch 64] [--epochs 1] [--steps 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/pretrain.py ...   (one rank per GPU, RCCL)

Without --data a synthetic in-memory set (1-10 channel 256 x 256 float planes) stands in for decoded images.  Reader threads decode,
crop / jitter / blur / flip run as HIP kernels on a side stream (chadavit_amd.data), the step is `Trainer.train_step`
(training_step -> backward [+ gradient all-reduce] -> on_after_backward -> fused AdamW -> EMA), the checkpoint is Lightning-shaped."""
import argparse
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch

from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
from chadavit_amd.data.loader import DevicePrefetcher, InMemoryPlanes
from chadavit_amd.data.sampler import TokenBalancedBatchSampler
from chadavit_amd.methods.dino import DINO
from chadavit_amd.parallel import GradSync, init_from_env
from chadavit_amd.trainer import Trainer
from chadavit_amd.utils.checkpoint import save_checkpoint
from chadavit_amd.utils.misc import AttrDict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None, help="IDRCell100k-format directory (train.csv + images/); default: synthetic planes")
    ap.add_argument("--embed-dim", type=int, default=192)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="stop after this many steps (0 = whole epochs)")
    ap.add_argument("--local-crops", type=int, default=6)
    ap.add_argument("--out", default=None, help="checkpoint path")
    a = ap.parse_args()
    rank, world, local = init_from_env()
    dev = torch.device("cuda", local)
    if a.data:
        from chadavit_amd.data.idrcell import IDRCell100K
        ds = IDRCell100K(root_dir=a.data, train=True)
    else:
        rs = np.random.RandomState(0)
        pool = {c: rs.rand(c, 256, 256).astype(np.float32) for c in range(1, 11)}
        ds = InMemoryPlanes([pool[1 + (i * 7) % 10] for i in range(max(4 * a.batch * world, 256))])
    sampler = TokenBalancedBatchSampler(ds.num_channels(), a.batch * world, rank, world)
    aug = dict(crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, flip_prob=0.5)
    specs = [CropSpec(crop_size=224, num_crops=1, blur_prob=1.0, **aug), CropSpec(crop_size=224, num_crops=1, blur_prob=0.1, solarize_prob=0.2, **aug)]
    if a.local_crops:
        specs.append(CropSpec(crop_size=96, num_crops=a.local_crops, crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5))
    cfg = AttrDict({
        "method": "dino", "backbone": {"name": "vit_channels", "kwargs": {"embed_dim": a.embed_dim, "patch_size": 16, "return_all_tokens": False,
                                                                          "max_number_channels": 10}},
        "data": {"dataset": "idrcell", "num_classes": 1, "max_img_channels": 10, "img_channels": 1, "num_large_crops": 2, "num_small_crops": a.local_crops},
        "channels_strategy": "multi_channels", "mixed_channels": True, "weights_init": "random", "max_epochs": a.epochs,
        "optimizer": {"name": "adamw", "batch_size": a.batch, "lr": 5e-4, "weight_decay": 0.04, "classifier_lr": 0.1},
        "scheduler": {"name": "warmup_cosine", "warmup_epochs": min(10, max(1, a.epochs // 10))}, "momentum": {"base_tau": 0.996, "final_tau": 1.0},
        "method_kwargs": {"proj_hidden_dim": 2048, "proj_output_dim": 256, "num_prototypes": 4096, "clip_grad": 3.0, "freeze_last_layer": 1,
                          "warmup_teacher_temperature_epochs": min(30, a.epochs)}})
    torch.manual_seed(0)                      # same initial weights on every rank (GradSync broadcasts rank 0's anyway)
    model = DINO(cfg).to(dev)
    trainer = Trainer(max_epochs=a.epochs, steps_per_epoch=len(sampler), grad_sync=GradSync() if world > 1 else None).attach(model)
    done = 0
    for epoch in range(a.epochs):
        trainer.current_epoch = epoch
        sampler.set_epoch(epoch)
        # (variable-channel datasets: opt in to the allocator rounding that keeps reserved memory flat -- process-wide, said once in the log)
        loader = DevicePrefetcher(ds, sampler, DeviceMultiCropPipeline(specs, dev, seed=1000 * epoch + rank), depth=2, workers=8,
                                  tune_allocator=len(set(ds.num_channels())) > 1 if callable(getattr(ds, "num_channels", None)) else False)
        for i, batch in enumerate(loader):
            loss = trainer.train_step(batch, i)
            done += 1
            if rank == 0 and (done % 10 == 0 or done == 1):
                print(f"epoch {epoch} step {i}: loss {float(loss):.4f} lr {trainer.optimizer.param_groups[0]['lr']:.2e} tau {model.momentum_updater.cur_tau:.5f}", flush=True)
            if a.steps and done >= a.steps:
                break
        if a.steps and done >= a.steps:
            break
    torch.cuda.synchronize()
    if rank == 0 and a.out:
        save_checkpoint(model, a.out, epoch=trainer.current_epoch, global_step=trainer.global_step)
        torch.save(trainer.optimizer.state_dict(), a.out + ".optimizer")
        print("saved", a.out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
