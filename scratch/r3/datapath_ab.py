"""The cfg2 step fed by the device data path: reader-thread count and interpreter switch interval (same box, 12 steps each).
    python scratch/r3/datapath_ab.py"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import torch

import bench
from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
from chadavit_amd.data.loader import DevicePrefetcher, InMemoryPlanes


def main():
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device("cuda:0")
    wl = dict(bench.WORKLOADS["cfg2"])
    model, tr, _, batch, nch, _ = bench.build_workload(wl, args, 0, 1, dev)
    B, side, n_samples, steps = wl["batch"], 256, 1536, 12
    for i in range(3):
        tr.train_step(batch, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.train_step(batch, 3 + i)
    torch.cuda.synchronize()
    print(f"resident: {B * steps / (time.perf_counter() - t0):.1f} images/s", flush=True)
    rs = np.random.RandomState(0)
    chans = bench.channel_list(wl["channels"], n_samples, seed=7)
    by_c = {c: rs.rand(c, side, side).astype(np.float32) for c in sorted(set(chans))}
    ds = InMemoryPlanes([by_c[c] for c in chans])
    specs = [CropSpec(crop_size=224, num_crops=1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=1.0, flip_prob=0.5),
             CropSpec(crop_size=224, num_crops=1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.1, solarize_prob=0.2, flip_prob=0.5),
             CropSpec(crop_size=96, num_crops=wl["n_local"], crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5)]
    batches = [list(range(i, i + B)) for i in range(0, n_samples - B + 1, B)]
    for workers, interval in ((32, 0.005), (4, 0.005), (32, 0.0005), (4, 0.0005), (4, 0.02), (1, 0.005)):
        sys.setswitchinterval(interval)
        ld = DevicePrefetcher(ds, batches * ((steps + 3 + len(batches) - 1) // len(batches)), DeviceMultiCropPipeline(specs, dev, seed=1), depth=2, workers=workers)
        n, host = 0, 0.0
        for i, b in enumerate(ld):
            if i == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            h0 = time.perf_counter()
            tr.train_step(b, 100 + i)
            if i >= 2:
                n += B
                host += time.perf_counter() - h0
            if i == steps + 1:
                break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"workers {workers:2d} switch interval {interval * 1e3:4.1f} ms: {n / dt:.1f} images/s; host time inside train_step {1e3 * host / steps:.1f} ms/step", flush=True)
    sys.setswitchinterval(0.005)


if __name__ == "__main__":
    main()
