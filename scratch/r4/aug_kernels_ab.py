"""Old (HEAD) vs new augmentation kernels: bit comparison on mixed shapes + stand-alone durations at the bench's batch."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from chadavit_amd import ops
from chadavit_amd._lib import lib
from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline

old = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "old_aug", "libold_augment.so"))
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())


def old_crop(src, d, S, shift, gamma):
    out = torch.empty((d.shape[0], 1, S, S), device=dev)
    rc = old.chadavit_crop_resize(P(src), P(d), P(shift), P(gamma), P(out), ctypes.c_int(d.shape[0]), ctypes.c_int(S), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return out


def old_blur(x, fin):
    out = torch.empty_like(x)
    rc = old.chadavit_blur_finish(P(x), P(fin), P(out), ctypes.c_int(x.shape[0]), ctypes.c_int(x.shape[-1]), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return out


def run(pipe, planes, n_time=0):
    shapes = [tuple(p.shape) for p in planes]
    offs, tot = [], 0
    for (C, H, W) in shapes:
        offs.append(tot); tot += C * H * W
    src = torch.from_numpy(np.concatenate([p.reshape(-1) for p in planes])).to(dev)
    worst = 0.0
    times = {}
    for spec in pipe.specs:
        for _ in range(spec.num_crops):
            cp = pipe._draw(spec, shapes)
            prep = pipe._prepare_crop(spec, cp, shapes, offs)
            a = ops.crop_resize(src, prep["d"], spec.crop_size, prep["shift"], prep["gamma"])
            b = old_crop(src, prep["d"], spec.crop_size, prep["shift"], prep["gamma"])
            eq1 = torch.equal(a, b)
            d1 = (a - b).abs().max().item()
            eq2, d2 = True, 0.0
            if prep["fin"] is not None:
                a2 = ops.blur_finish(a, prep["fin"])
                b2 = old_blur(a, prep["fin"])
                eq2 = torch.equal(a2, b2)
                d2 = (a2 - b2).abs().max().item()
            print(f"S={spec.crop_size} n={a.shape[0]} resize identical={eq1} (max diff {d1:.2e}) finish identical={eq2} (max diff {d2:.2e})", flush=True)
            worst = max(worst, d1, d2)
            if n_time:
                for name, fn in (("new_crop", lambda: ops.crop_resize(src, prep["d"], spec.crop_size, prep["shift"], prep["gamma"], out=a)),
                                 ("old_crop", lambda: old_crop(src, prep["d"], spec.crop_size, prep["shift"], prep["gamma"])),
                                 ("new_blur", (lambda: ops.blur_finish(a, prep["fin"], out=a2)) if prep["fin"] is not None else None),
                                 ("old_blur", (lambda: old_blur(a, prep["fin"])) if prep["fin"] is not None else None)):
                    if fn is None:
                        continue
                    fn(); torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n_time):
                        fn()
                    e1.record(); torch.cuda.synchronize()
                    times.setdefault((name, spec.crop_size), []).append(e0.elapsed_time(e1) / n_time * 1e3)
    return worst, times


rs = np.random.RandomState(0)
# (1) mixed shapes, every transform, odd sizes (non-vector paths), copy path (crop == output size without rrc)
planes = [rs.rand(c, h, w).astype(np.float32) for c, h, w in ((3, 256, 256), (1, 200, 310), (5, 97, 131), (10, 64, 64), (2, 224, 224), (4, 300, 180))]
specs = [CropSpec(crop_size=224, num_crops=2, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.7, solarize_prob=0.5, solarize_threshold=0.6,
                  flip_prob=0.5, normalize=([0.4, 0.5, 0.6], [0.2, 0.25, 0.3], 1.0), normalize_prob=0.6),
         CropSpec(crop_size=96, num_crops=3, crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5),
         CropSpec(crop_size=224, num_crops=1, rrc_enabled=False, blur_prob=0.5),
         CropSpec(crop_size=97, num_crops=2, crop_min_scale=0.1, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.8, flip_prob=0.5),
         CropSpec(crop_size=30, num_crops=1, crop_min_scale=0.1, crop_max_scale=1.0, blur_prob=1.0, flip_prob=0.5),
         CropSpec(crop_size=8, num_crops=1, blur_prob=1.0), CropSpec(crop_size=512, num_crops=1, jitter_prob=0.8, blur_prob=1.0)]
w, _ = run(DeviceMultiCropPipeline(specs, dev, seed=5), planes)
print("worst difference, mixed shapes:", w)
# (2) the bench's batch: 1024 x 3 x 256 x 256, asymmetric DINO specs
plane = rs.rand(3, 256, 256).astype(np.float32)
specs = [CropSpec(crop_size=224, num_crops=1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=1.0, flip_prob=0.5),
         CropSpec(crop_size=224, num_crops=1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.1, solarize_prob=0.2, flip_prob=0.5),
         CropSpec(crop_size=96, num_crops=8, crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5)]
w, times = run(DeviceMultiCropPipeline(specs, dev, seed=1), [plane] * 1024, n_time=5)
print("worst difference, bench batch:", w)
tot = {}
for (name, S), v in sorted(times.items()):
    print(name, S, [round(x, 1) for x in v], "us")
    tot[name] = tot.get(name, 0.0) + sum(v)
print({k: round(v / 1e3, 3) for k, v in tot.items()}, "ms per 1024-image batch (10 crops)")
