"""Sustained run of the cfg2 step fed by the device data path (round 4's prefetcher + native draws + sub-banded kernels): N steps from float32
planes, N from uint8 planes; the loss stays finite, reserved memory stops growing after the first steps, throughput per 50 steps is steady."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
from chadavit_amd.data.loader import DevicePrefetcher, InMemoryPlanes

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=150); ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--keep-losses", action="store_true", help="keep the loss tensors themselves (with their graphs), as a careless loop would")
ap.add_argument("--mixed", action="store_true", help="1-10 channel samples (a new channel mix every batch) instead of 3-channel ones")
a = ap.parse_args()
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["cfg2-mixed" if a.mixed else "cfg2"]); wl["batch"] = a.batch
model, tr, _, _, _, _ = bench.build_workload(wl, argparse.Namespace(serial=False, overlap=False), 0, 1, dev)
rs = np.random.RandomState(0)
specs = [CropSpec(224, 1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=1.0, flip_prob=0.5),
         CropSpec(224, 1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.1, solarize_prob=0.2, flip_prob=0.5),
         CropSpec(96, 8, crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5)]
B = a.batch
for kind in ("float32", "uint8"):
    chans = [int(rs.randint(1, 11)) if a.mixed else 3 for _ in range(32)]
    planes = [(rs.rand(c, 256, 256).astype(np.float32) if kind == "float32" else rs.randint(0, 256, size=(c, 256, 256)).astype(np.uint8)) for c in chans]
    ds = InMemoryPlanes([planes[i % 32] for i in range(64 * B)] if a.mixed else [planes[i % 8] for i in range(2 * B)])
    if a.mixed:   # every batch its own draw of samples -> its own channel mix, ragged description and buffer sizes
        batches = [[int(j) for j in rs.randint(0, 64 * B, size=B)] for _ in range(a.steps)]
    else:
        batches = [list(range(B)), list(range(B, 2 * B))] * (a.steps // 2 + 1)
    t0 = time.perf_counter(); losses = []; mem = []
    for i, batch in enumerate(DevicePrefetcher(ds, batches[:a.steps], DeviceMultiCropPipeline(specs, dev, seed=1), depth=2, workers=16, raw_planes=True)):
        l_ = tr.train_step(batch, i)
        losses.append(l_ if a.keep_losses else l_.detach().clone())   # (before the round-4 fix a kept loss kept its whole batch alive)
        if (i + 1) % 50 == 0:
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_reserved() / 2**30)
            print(f"{kind}: step {i + 1}: {50 * B / (time.perf_counter() - t0):7.1f} images/s  loss {float(losses[-1]):.4f}  reserved {mem[-1]:.1f} GiB  peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
            t0 = time.perf_counter()
    vals = torch.stack([l.detach() for l in losses]).float()
    assert bool(torch.isfinite(vals).all()), "non-finite loss"
    assert len(mem) < 2 or mem[-1] <= mem[0] * (1.5 if a.mixed else 1.05) + 0.5, ("reserved memory keeps growing", mem)
print("fed soak OK")
