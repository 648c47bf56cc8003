"""(f)2 data path, CPU side: the oracle's restatement of the cv2 / albumentations arithmetic (oracle/augment_ref.py) against
independent implementations and closed forms, the host-side parameter draws, and the IDRCell100k-format reader.
cv2 / albumentations are absent from the image: what is checked is the published algorithm, see the oracle's header."""
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import augment_ref as A


def _img(h, w, seed):
    return np.random.RandomState(seed).rand(h, w).astype(np.float32)


@pytest.mark.parametrize("h,w,S", [(224, 224, 96), (150, 201, 224), (37, 53, 96), (224, 224, 224), (64, 48, 96), (300, 300, 224)])
def test_resize_cubic_matches_torch_bicubic(h, w, S):
    """torch's bicubic (align_corners=False, no antialias) is documented to match OpenCV's INTER_CUBIC: A = -0.75, half-pixel
    centres, border taps clamped.  Independent implementation of the same published algorithm -> agreement up to the rounding of the
    source coordinate (OpenCV forms it in double and casts, torch in float32: a few 1e-5 on [0, 1] images)."""
    img = _img(h, w, 1)
    ref = F.interpolate(torch.from_numpy(img)[None, None], size=(S, S), mode="bicubic", align_corners=False)[0, 0].numpy()
    np.testing.assert_allclose(A.resize_cubic(img, S), ref, atol=6e-5, rtol=0)


def test_cubic_closed_forms():
    t = np.linspace(0, 1, 33, dtype=np.float32)
    w = A.cubic_weights(t)
    np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-6)                       # partition of unity: constants are reproduced
    np.testing.assert_allclose(A.cubic_weights(np.float32(0.0)), [0, 1, 0, 0], atol=1e-7)   # interpolating at the nodes
    np.testing.assert_allclose(w[:, ::-1], A.cubic_weights(1 - t), atol=1e-6)   # symmetric
    np.testing.assert_allclose(A.cubic_weights(np.float32(0.5)), [-0.09375, 0.59375, 0.59375, -0.09375], atol=1e-7)  # A = -0.75
    const = np.full((40, 56), 0.37, dtype=np.float32)
    np.testing.assert_allclose(A.resize_cubic(const, 96), 0.37, atol=1e-6)
    # integer upscale x2 of a delta: the response is the tap pattern at t = 0.25 / 0.75
    d = np.zeros((9, 9), dtype=np.float32); d[4, 4] = 1
    up = A.resize_cubic(d, 18)
    w25, w75 = A.cubic_weights(np.float32(0.25)), A.cubic_weights(np.float32(0.75))
    # output pixel 8: fx = 8.5 * 0.5 - 0.5 = 3.75 -> taps 2..5, weight of source 4 is w75[2]; pixel 9: fx = 4.25 -> taps 3..6, w25[1]
    np.testing.assert_allclose(up[8, 8], w75[2] * w75[2], atol=1e-6)
    np.testing.assert_allclose(up[9, 8], w25[1] * w75[2], atol=1e-6)
    assert A.resize_cubic(d, 9) is not d and np.array_equal(A.resize_cubic(d, 9), d)   # same size: a copy


def test_gaussian_blur_closed_forms():
    from scipy import ndimage
    for k, s in ((3, 0.1), (3, 0.8), (5, 1.3), (7, 2.0)):
        g = A.gaussian_kernel1d(k, s)
        assert abs(g.sum() - 1) < 1e-6 and np.allclose(g, g[::-1]) and g.argmax() == k // 2
        img = _img(30, 41, k)
        ref = ndimage.correlate1d(ndimage.correlate1d(img, g, axis=1, mode="mirror"), g, axis=0, mode="mirror")  # mirror = reflect-101
        np.testing.assert_allclose(A.gaussian_blur(img, k, s), ref, atol=2e-6)
        d = np.zeros((15, 15), dtype=np.float32); d[7, 7] = 1
        np.testing.assert_allclose(A.gaussian_blur(d, k, s)[7 - k // 2:8 + k // 2, 7 - k // 2:8 + k // 2], np.outer(g, g), atol=1e-7)
    np.testing.assert_allclose(A.gaussian_blur(np.full((9, 9), 2.5, np.float32), 7, 1.0), 2.5, atol=1e-5)


def test_pointwise_steps():
    x = np.array([[0.0, 0.2, 0.5, 0.9, 1.0]], dtype=np.float32)
    np.testing.assert_allclose(A.solarize(x, 0.5), [[0.0, 0.2, 0.5, 0.1, 0.0]], atol=1e-7)
    assert np.array_equal(A.solarize(x, 128.0), x)          # albumentations' default threshold on [0, 1] floats: no-op
    np.testing.assert_allclose(A.normalize(x * 255, 0.5, 0.25), (x - 0.5) / 0.25, atol=1e-5)
    np.testing.assert_allclose(A.color_jitter(x, 0.1, 2.0), np.clip(2.0 * (x + 0.1), 0, 1), atol=1e-7)
    # the reference-owned jitter itself is pinned by tests/golden/jitter.npz (tests/test_oracle_golden.py)


def test_gating_and_draw_order_of_the_host_side_draws():
    """The reference's CustomColorJitter overrides __call__ (custom_transforms.py:309-311): applied to EVERY crop once it is in the
    list, no Python-`random` draw; every albumentations transform draws `random.random() < p` per sample, ToTensorV2 included;
    A.Normalize honours its p (pretrain_dataloader.py:322-323)."""
    from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline

    class Counting(random.Random):
        def __init__(self, seed):
            super().__init__(seed)
            self.n = 0

        def random(self):
            self.n += 1
            return super().random()

    shapes = [(3, 64, 64)] * 200
    # jitter only: crop p + ToTensor p per sample, plus the crop's own parameter draws -- the jitter adds none
    a = DeviceMultiCropPipeline([CropSpec(crop_size=32, jitter_prob=0.8, rrc_enabled=False)], "cpu", seed=1)
    b = DeviceMultiCropPipeline([CropSpec(crop_size=32, jitter_prob=0.0, rrc_enabled=False)], "cpu", seed=1)
    a.rng, b.rng = Counting(1), Counting(1)
    pa, pb = a._draw(a.specs[0], shapes), b._draw(b.specs[0], shapes)
    assert a.rng.n == b.rng.n == 2 * len(shapes)
    assert all(s is not None for s in pa.shifts) and all(s is None for s in pb.shifts)
    assert all(g.shape == (3,) and (0.5 <= g).all() and (g <= 1.5).all() for g in pa.gammas)
    # normalize.prob: fires for about that fraction of the samples, one extra draw each; gray likewise
    c = DeviceMultiCropPipeline([CropSpec(crop_size=32, rrc_enabled=False, gray_prob=0.3, normalize=([0.5], [0.25], 1.0), normalize_prob=0.25)],
                                "cpu", seed=2)
    c.rng = Counting(2)
    pc = c._draw(c.specs[0], shapes)
    assert c.rng.n == 4 * len(shapes)
    assert 0.15 < sum(pc.normalized) / len(shapes) < 0.35 and 0.2 < sum(pc.grays) / len(shapes) < 0.4
    assert A.draw_order(True, True, False, False, False, True) == ["crop:p", "crop:params...", "gray:p", "to_tensor:p", "normalize:p"]
    # cfg mapping (keys of pretrain_dataloader.py:232-255)
    aug = {"crop_size": 96, "num_crops": 2, "rrc": {"enabled": True, "crop_min_scale": 0.05, "crop_max_scale": 0.25},
           "color_jitter": {"prob": 0.8, "int_min_shift": -0.2, "int_max_shift": 0.2, "gamma_min": 0.7, "gamma_max": 1.3},
           "grayscale": {"prob": 0.2}, "gaussian_blur": {"prob": 0.1}, "solarization": {"prob": 0.2}, "equalization": {"prob": 0.0},
           "horizontal_flip": {"prob": 0.5}, "normalize": {"prob": 0.7}, "mean": [0.1, 0.2], "std": [0.3, 0.4]}
    sp = CropSpec.from_cfg(aug)
    assert (sp.gray_prob, sp.normalize_prob, sp.jitter_prob, sp.blur_prob, sp.flip_prob) == (0.2, 0.7, 0.8, 0.1, 0.5)
    assert sp.normalize == ([0.1, 0.2], [0.3, 0.4], 255.0)
    with pytest.raises(RuntimeError, match="Equalize needs uint8"):
        CropSpec.from_cfg(dict(aug, equalization={"prob": 0.1}))
    x = np.random.RandomState(0).rand(3, 5, 5).astype(np.float32)
    g = A.to_gray(x)
    np.testing.assert_allclose(g[1], 0.299 * x[0] + 0.587 * x[1] + 0.114 * x[2], atol=1e-6)
    assert np.array_equal(g[0], g[2])


def test_rrc_parameter_draws():
    """Host-side draws of the device pipeline == the oracle's restatement for the same seed; boxes are inside the image, the area
    fraction is in `scale`, aspect ratio in [3/4, 4/3] up to integer rounding."""
    from chadavit_amd.data.device_pipeline import rrc_box
    r1, r2 = random.Random(5), random.Random(5)
    for H, W in ((224, 224), (300, 180), (97, 131)):
        for _ in range(50):
            b1 = rrc_box(H, W, (0.25, 1.0), (3 / 4, 4 / 3), r1)
            b2 = A.random_resized_crop_params(H, W, (0.25, 1.0), r2)
            assert b1 == b2
            y0, x0, h, w = b1
            assert 0 <= y0 and y0 + h <= H and 0 <= x0 and x0 + w <= W and h > 0 and w > 0
            assert 0.24 <= h * w / (H * W) <= 1.0 and 0.7 <= w / h <= 1.4


def test_native_draws_continue_the_same_stream_as_the_python_statement():
    """chadavit_draw_crop_params (csrc/host_draw.hip, a host function of the C-ABI library) draws the per-sample parameters from the
    exported state of the pipeline's `random.Random` and writes the state back: for every transform mix -- RandomResizedCrop with its
    10 attempts and the central fallback, plain Resize, even blur sizes moved to the next odd one, a solarize threshold, gray,
    normalize.prob -- and for degenerate and oblong planes it must hand out exactly the values the Python loop hands out (itself pinned
    to oracle/augment_ref.py above), leave the generator in exactly the same state, and lead to identical descriptor tables."""
    from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
    specs = [CropSpec(224, 1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=1.0, flip_prob=0.5),
             CropSpec(224, 1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.1, solarize_prob=0.2, flip_prob=0.5, gray_prob=0.2,
                      normalize=([0.5], [0.2], 255.0), normalize_prob=0.7, int_min_shift=-0.1, gamma_max=1.7),
             CropSpec(96, 3, crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5),
             CropSpec(64, 1, rrc_enabled=False, blur_prob=0.5, blur_limit=(2, 6), solarize_prob=0.5, solarize_threshold=0.4),
             CropSpec(64, 2, crop_min_scale=0.9, crop_max_scale=1.0, ratio=(0.2, 0.3)),     # mostly the central fallback
             CropSpec(64, 1, crop_min_scale=0.9, crop_max_scale=1.0, ratio=(3.0, 4.0))]     # ... its other branch
    rs = np.random.RandomState(0)
    mixed = [(3 if i % 5 else int(rs.randint(1, 11)), int(rs.randint(20, 400)), int(rs.randint(20, 400))) for i in range(300)]
    mixed += [(3, 1, 1), (3, 2, 5), (3, 1000, 7), (1, 7, 1000)]
    three = [(3, h, w) for _, h, w in mixed]
    fallbacks = 0
    for seed in (0, 1, 12345):
        a, b = DeviceMultiCropPipeline(specs, "cpu", seed=seed), DeviceMultiCropPipeline(specs, "cpu", seed=seed)
        b.python_draws = True
        for sp in specs:
            shapes = three if sp.gray_prob else mixed
            offs, tot = [], 0
            for (C, H, W) in shapes:
                offs.append(tot); tot += C * H * W
            for _ in range(sp.num_crops):
                pa, pb = a._draw(sp, shapes), b._draw(sp, shapes)
                assert pa.arrays is not None and pb.arrays is None   # (really the two implementations)
                assert pa.boxes == pb.boxes and pa.grays == pb.grays and pa.blurs == pb.blurs and pa.solarize == pb.solarize
                assert pa.flips == pb.flips and pa.normalized == pb.normalized
                assert all(type(x) is type(y) for x, y in zip(pa.boxes[0], pb.boxes[0]))
                assert a.rng.getstate() == b.rng.getstate()
                if min(min(bx[2], bx[3]) for bx in pa.boxes) > 0:
                    ta, tb = a._prepare_crop(sp, pa, shapes, offs), b._prepare_crop(sp, pb, shapes, offs)
                    for k in ("d", "shift", "gamma", "fin"):
                        assert (ta[k] is None and tb[k] is None) or np.array_equal(ta[k].numpy(), tb[k].numpy(), equal_nan=True), k
                if sp.ratio != (3.0 / 4.0, 4.0 / 3.0):
                    fallbacks += sum(1 for (y0, x0, h, w), (_, H, W) in zip(pa.boxes, shapes) if (h == H or w == W))
                for (y0, x0, h, w), (_, H, W) in zip(pa.boxes, shapes):
                    assert 0 <= y0 and y0 + h <= H and 0 <= x0 and x0 + w <= W and h >= 0 and w >= 0
                    assert (h > 0 and w > 0) or min(H, W) < 4   # (round(1 * 0.3) = 0: albumentations' own fallback on a 1-pixel plane)
    assert fallbacks > 100
    # ... and an empty crop window is refused where the tables are built (cv2.resize raises on it in the reference)
    e = DeviceMultiCropPipeline([specs[4]], "cpu", seed=0)
    with pytest.raises(RuntimeError, match="empty crop window"):
        e._prepare_crop(specs[4], e._draw(specs[4], [(3, 1, 1)]), [(3, 1, 1)], [0])
    # a generator object of any other type (here: one that counts) keeps the Python loop -- its draws stay observable
    class Counting(random.Random):
        n = 0

        def random(self):
            self.n += 1
            return super().random()
    c = DeviceMultiCropPipeline([specs[3]], "cpu", seed=3)
    c.rng = Counting(3)
    assert c._draw(specs[3], mixed).arrays is None and c.rng.n > 2 * len(mixed)
    # ToGray on a sample that is not 3-channel raises on either path, as albumentations does
    for py in (False, True):
        d = DeviceMultiCropPipeline([CropSpec(32, 1, gray_prob=1.0)], "cpu", seed=0)
        d.python_draws = py
        with pytest.raises(RuntimeError, match="ToGray fired"):
            d._draw(d.specs[0], [(3, 40, 40), (2, 40, 40)])


def test_idrcell_reader(tmp_path):
    """IDRCell100k format (custom_datasets.py:153-220): csv of (id, list of per-channel files under images/) -> HWC float32."""
    from PIL import Image
    from chadavit_amd.data.idrcell import IDRCell100K
    root = tmp_path
    os.makedirs(root / "images" / "exp1")
    rs = np.random.RandomState(0)
    rows, truth = [], []
    for i, c in enumerate([3, 1, 5]):
        paths, planes = [], []
        for ch in range(c):
            arr = rs.randint(0, 65535, size=(20 + i, 24), dtype=np.uint16) if ch % 2 == 0 else rs.randint(0, 255, size=(20 + i, 24), dtype=np.uint8)
            rel = f"exp1/img{i}_ch{ch}.{'tif' if ch % 2 == 0 else 'png'}"
            Image.fromarray(arr).save(root / "images" / rel)
            paths.append(rel); planes.append(arr.astype(np.float32))
        rows.append((f"id{i}", paths)); truth.append(np.stack(planes, 0))
    with open(root / "train.csv", "w") as f:
        for iid, paths in rows:
            f.write(f'{iid},"{paths}"\n')
    ds = IDRCell100K(root_dir=str(root), train=True)
    assert len(ds) == 3 and ds.num_channels() == [3, 1, 5]
    for i in range(3):
        planes = ds.read_planes(i)
        assert planes.dtype == np.float32 and np.array_equal(planes, truth[i])
        # raw: the stored type when the sample's files share one (sample 1 is a single 16-bit file), float32 otherwise
        raw = ds.read_planes(i, raw=True)
        assert raw.dtype == (np.uint16 if i == 1 else np.float32) and np.array_equal(raw.astype(np.float32), truth[i])
        img, label = ds[i]
        assert label == -1 and img.shape == (20 + i, 24, len(rows[i][1])) and np.array_equal(img.transpose(2, 0, 1), truth[i])
    seen = {}
    ds2 = IDRCell100K(root_dir=str(root), train=True, transform=lambda image: seen.setdefault("shape", image.shape) and {"image": image})
    out, _ = ds2[2]
    assert seen["shape"] == (22, 24, 5)


def test_device_prefetcher_plumbing_without_a_gpu():
    """DevicePrefetcher's host side (reader pool, bounded queue, ordering, error hand-over, early exit) with a stand-in pipeline on the
    CPU: batches arrive in sampler order, each built from exactly the planes of its indices; a reader error reaches the consumer; a
    consumer that stops early does not leave the producer thread behind."""
    import threading
    from chadavit_amd.data.loader import DevicePrefetcher, InMemoryPlanes

    class FakePipe:
        device = torch.device("cpu")

        def __call__(self, planes, labels=None):
            x = torch.stack([torch.from_numpy(p).sum() for p in planes])
            return [x, x + 1], torch.as_tensor(labels if labels is not None else [-1] * len(planes)), [[p.shape[0] for p in planes]] * 2

    rs = np.random.RandomState(0)
    ds = InMemoryPlanes([rs.rand(1 + i % 3, 4, 5).astype(np.float32) for i in range(12)])
    batches = [[0, 5, 7], [1, 2, 3], [11, 4, 6], [8, 9, 10]]
    got = list(DevicePrefetcher(ds, batches, FakePipe(), depth=2, workers=3, labels=list(range(100, 112))))
    assert len(got) == 4
    for idx, (crops, labels, ncl) in zip(batches, got):
        np.testing.assert_allclose(crops[0].numpy(), [ds.planes[i].sum() for i in idx], rtol=1e-6)
        assert labels.tolist() == [100 + i for i in idx] and ncl[0] == [ds.planes[i].shape[0] for i in idx]

    class Broken(InMemoryPlanes):
        def read_planes(self, index):
            if index == 3:
                raise OSError("unreadable channel file")
            return super().read_planes(index)

    with pytest.raises(OSError, match="unreadable channel file"):
        list(DevicePrefetcher(Broken(ds.planes), batches, FakePipe(), depth=1, workers=2))
    n0 = threading.active_count()
    it = iter(DevicePrefetcher(ds, batches * 50, FakePipe(), depth=1, workers=2))
    next(it); next(it)
    it.close()   # generator exit: the producer is told to stop and joined
    assert threading.active_count() <= n0

    class Stored(InMemoryPlanes):   # a reader with the two modes of IDRCell100K.read_planes: stored type on request, float32 otherwise
        def read_planes(self, index, raw=False):
            return self.planes[index] if raw else self.planes[index].astype(np.float32)
    ds8 = Stored([(rs.rand(c, 4, 5) * 255).astype(np.uint8) for c in (3, 1, 2, 3)])
    seen = set()

    class TypePipe(FakePipe):
        def __call__(self, planes, labels=None):
            seen.update(p.dtype for p in planes)
            return super().__call__([p.astype(np.float32) for p in planes], labels=labels)
    for raw, want in ((False, np.float32), (True, np.uint8)):   # raw_planes: the producer asks for the stored type
        seen.clear()
        assert len(list(DevicePrefetcher(ds8, [[0, 1], [2, 3]], TypePipe(), depth=1, workers=2, raw_planes=raw))) == 2
        assert seen == {np.dtype(want)}


def test_deferred_kernels_return_the_same_batch_and_a_launch_callable():
    """DeviceMultiCropPipeline(..., defer=True) (DevicePrefetcher(kernels_on="consumer")): the call itself only prepares -- on the CPU
    the kernels cannot run at all (no CPU fallback), so the deferred form must get through the host side and hand back the buffers, the
    labels, the channel lists and ONE launch callable; the plain form must fail loudly in the first kernel."""
    import pytest
    from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
    rs = np.random.RandomState(0)
    imgs = [rs.rand(c, 40, 48).astype(np.float32) for c in (2, 1)]
    specs = [CropSpec(crop_size=32, num_crops=2, crop_min_scale=0.3, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5),
             CropSpec(crop_size=16, num_crops=1, crop_min_scale=0.1, crop_max_scale=0.3)]
    pipe = DeviceMultiCropPipeline(specs, "cpu", seed=3)
    crops, lab, ncl, launch = pipe(imgs, labels=[4, 5], defer=True)
    assert [tuple(c.shape) for c in crops] == [(3, 1, 32, 32), (3, 1, 32, 32), (3, 1, 16, 16)] and lab.tolist() == [4, 5] and ncl == [[2, 1]] * 3
    assert crops[0].data_ptr() + crops[0].numel() * 4 == crops[1].data_ptr()     # crops of one resolution back to back (adjacent_view)
    assert callable(launch) and len(pipe.last_params) == 3
    with pytest.raises(Exception):
        launch()                                       # the kernels need the GPU library's device path
    with pytest.raises(Exception):
        DeviceMultiCropPipeline(specs, "cpu", seed=3)(imgs)


def test_allocator_tuning_is_opt_in_and_keeps_the_users_configuration(monkeypatch):
    """the process-wide allocator rounding is never a side effect of building a DevicePrefetcher; when asked for it is
    APPENDED to the user's allocator configuration, and a configuration that already sets a roundup option is left alone."""
    import torch
    from chadavit_amd.data import loader as L
    calls = []
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch._C, "_accelerator_setAllocatorSettings", lambda conf: calls.append(conf), raising=False)
    monkeypatch.setattr(L, "_ALLOCATOR_TUNED", False)
    for k in ("PYTORCH_ALLOC_CONF", "PYTORCH_HIP_ALLOC_CONF", "PYTORCH_CUDA_ALLOC_CONF"):
        monkeypatch.delenv(k, raising=False)

    class DS:
        def num_channels(self):
            return [1, 3, 5]

        def read_planes(self, i, raw=False):
            raise AssertionError("not read")

    class Pipe:
        device = torch.device("cpu")

    L.DevicePrefetcher(DS(), [[0]], Pipe())                 # mixed channel counts, no opt-in: nothing happens
    assert calls == []
    monkeypatch.setenv("PYTORCH_HIP_ALLOC_CONF", "max_split_size_mb:512,garbage_collection_threshold:0.8")
    assert L.tune_allocator_for_ragged_batches() is True
    assert calls == ["max_split_size_mb:512,garbage_collection_threshold:0.8,roundup_power2_divisions:8"]
    assert L.tune_allocator_for_ragged_batches() is False   # once per process
    monkeypatch.setattr(L, "_ALLOCATOR_TUNED", False)
    monkeypatch.setenv("PYTORCH_HIP_ALLOC_CONF", "roundup_power2_divisions:[256:1,512:2,>:4]")
    assert L.tune_allocator_for_ragged_batches() is False and len(calls) == 1   # the user's own roundup list wins
