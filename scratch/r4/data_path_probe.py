"""Where does the fed step lose against the resident one?  (round 4, verdict item 6)

Runs the cfg2 step (1024 images / GPU) in these modes, N timed steps each, interleaved twice:
  resident            one resident pipeline batch, nothing else running
  host-only           resident batch + a producer thread that does the HOST work of the data path only (draws, tables, staging memcpy)
  copies-only         resident batch + producer that also issues the H2D copies (kernels never launched)
  producer            the real thing, augmentation kernels on the side stream
  producer-lowprio    same, side stream created with the lowest priority
  consumer            kernels at the head of the step's stream
MODE=alone: the pipeline alone (for a rocprofv3 --kernel-trace of the kernels' stand-alone durations).
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
from chadavit_amd.data.loader import DevicePrefetcher, InMemoryPlanes

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--modes", default="resident,host-only,copies-only,producer,producer-lowprio,consumer,resident")
ap.add_argument("--batch", type=int, default=1024)
a = ap.parse_args()

dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["cfg2"]); wl["batch"] = a.batch
args = argparse.Namespace(serial=False, overlap=False)
model, tr, gs, batch0, nch0, _ = bench.build_workload(wl, args, 0, 1, dev)
B, steps, side, n_samples = wl["batch"], a.steps, 256, 1536 if a.batch <= 1024 else 2 * a.batch
rs = np.random.RandomState(0)
nch = bench.channel_list(wl["channels"], n_samples, seed=7)
by_c = {c: rs.rand(c, side, side).astype(np.float32) for c in sorted(set(nch))}
ds = InMemoryPlanes([by_c[c] for c in nch])
specs = [CropSpec(crop_size=224, num_crops=1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=1.0, flip_prob=0.5),
         CropSpec(crop_size=224, num_crops=1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.1, solarize_prob=0.2, flip_prob=0.5),
         CropSpec(crop_size=96, num_crops=wl["n_local"], crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5)]
batches = [list(range(i, i + B)) for i in range(0, n_samples - B + 1, B)]
many = batches * ((steps + 4 + len(batches) - 1) // len(batches))


def loader(kernels_on="producer", depth=2, **kw):
    return DevicePrefetcher(ds, many, DeviceMultiCropPipeline(specs, dev, seed=1), depth=depth, workers=32, kernels_on=kernels_on, **kw)


one = next(iter(loader()))
torch.cuda.synchronize()
for i in range(3):
    tr.train_step(one, i)
torch.cuda.synchronize()


def timed_resident(n, base):
    for i in range(2):
        tr.train_step(one, base + i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        tr.train_step(one, base + 2 + i)
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)


def fed(kernels_on, base, **kw):
    n = 0
    for i, batch in enumerate(loader(kernels_on, **kw)):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.train_step(batch, base + i)
        if i >= 2:
            n += B
        if i == steps + 1:
            break
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def background(kind, stop):
    """The producer's work without (host-only) or with (copies-only) its H2D copies; kernels never run."""
    torch.cuda.set_device(dev)
    st = torch.cuda.Stream(device=dev)
    if kind == "host-only":
        pipe = DeviceMultiCropPipeline(specs, torch.device("cpu"), seed=1)
        # the CPU device path would run the kernels' CPU stand-in: only the draw + tables + staging are wanted
        while not stop.is_set():
            planes = [ds.read_planes(i) for i in batches[0]]
            shapes = [tuple(p.shape) for p in planes]
            offs, tot = [], 0
            for (C, H, W) in shapes:
                offs.append(tot); tot += C * H * W
            host = np.empty(tot, dtype=np.float32)
            o = 0
            for p in planes:
                host[o:o + p.size] = p.reshape(-1); o += p.size
            for spec in specs:
                for _ in range(spec.num_crops):
                    cp = pipe._draw(spec, shapes)
                    pipe._prepare_crop(spec, cp, shapes, offs)
            time.sleep(0.03)
    else:
        pipe = DeviceMultiCropPipeline(specs, dev, seed=1)
        with torch.cuda.stream(st):
            while not stop.is_set():
                planes = [ds.read_planes(i) for i in batches[0]]
                out = pipe(planes, defer=True)
                st.synchronize()
                del out
                time.sleep(0.03)


def low_priority_stream():
    """torch only hands out default- or higher-priority streams: the lowest HIP priority comes from the runtime itself."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    lo, hi = ctypes.c_int(0), ctypes.c_int(0)
    assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, lo.value) == 0   # 1 = hipStreamNonBlocking
    print("priority range (least, greatest):", lo.value, hi.value, flush=True)
    return torch.cuda.ExternalStream(s.value, device=dev)


res = {}
base = 1000
for rep in range(2):
    for mode in a.modes.split(","):
        base += 100
        if mode == "resident":
            v = timed_resident(steps, base)
        elif mode in ("host-only", "copies-only"):
            stop = threading.Event()
            th = threading.Thread(target=background, args=(mode, stop), daemon=True)
            th.start()
            time.sleep(0.5)
            v = timed_resident(steps, base)
            stop.set(); th.join()
        elif mode == "producer":
            v = fed("producer", base)
        elif mode == "producer-lowprio":
            v = fed("producer", base, stream=low_priority_stream())
        elif mode == "producer-depth4":
            v = fed("producer", base, depth=4)
        elif mode == "consumer":
            v = fed("consumer", base)
        elif mode == "alone":
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
            for i, b_ in enumerate(loader()):
                n += B
                if i == steps:
                    break
            torch.cuda.synchronize()
            v = n / (time.perf_counter() - t0)
        else:
            raise SystemExit(mode)
        res.setdefault(mode, []).append(round(v, 1))
        print(mode, round(v, 1), flush=True)
print(res)
