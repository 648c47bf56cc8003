"""(f)2 data path on the GPU: chadavit_amd.data.device_pipeline (HIP crop-resize / jitter / flip / blur / solarize / normalise
kernels) against the oracle's restatement of the reference pipeline's arithmetic (oracle/augment_ref.py) on the same drawn
parameters, and its output layout against the reference collate (A1)."""
import numpy as np
import pytest
import torch

from oracle import augment_ref as A
from oracle import chada_ref as R

pytestmark = pytest.mark.gpu


def _batch(seed=0):
    rs = np.random.RandomState(seed)
    shapes = [(3, 224, 224), (1, 150, 201), (5, 97, 131), (10, 224, 224), (2, 64, 48)]
    return [rs.rand(*s).astype(np.float32) for s in shapes]


def test_device_multicrop_matches_oracle_and_collate_layout():
    from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
    dev = torch.device("cuda:0")
    imgs = _batch()
    specs = [CropSpec(crop_size=224, num_crops=2, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.7, solarize_prob=0.5,
                      solarize_threshold=0.6, flip_prob=0.5, normalize=([0.4, 0.5, 0.6], [0.2, 0.25, 0.3], 1.0), normalize_prob=0.6),
             CropSpec(crop_size=96, num_crops=3, crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5),
             CropSpec(crop_size=224, num_crops=1, rrc_enabled=False)]   # plain Resize branch (pretrain_dataloader.py:292-299)
    pipe = DeviceMultiCropPipeline(specs, dev, seed=3)
    crops, labels, ncl = pipe(imgs, labels=[0, 1, 2, 3, 4])
    assert len(crops) == 6 and [tuple(c.shape) for c in crops] == [(21, 1, 224, 224)] * 2 + [(21, 1, 96, 96)] * 3 + [(21, 1, 224, 224)]
    assert ncl == [[3, 1, 5, 10, 2]] * 6 and labels.tolist() == [0, 1, 2, 3, 4]
    drew = {"jit": 0, "blur": 0, "sol": 0, "flip": 0, "norm": 0, "nonorm": 0}
    k = 0
    for spec in specs:
        for _ in range(spec.num_crops):
            cp = pipe.last_params[k]
            ref_planes = []
            for i, im in enumerate(imgs):
                for c in range(im.shape[0]):
                    norm = None
                    if cp.normalized[i]:
                        mean, std, mpv = spec.normalize
                        norm = (mean[c % 3], std[c % 3], mpv)
                    ref_planes.append(A.augment_plane(im[c], spec.crop_size, cp.boxes[i],
                                                      None if cp.shifts[i] is None else cp.shifts[i][c],
                                                      None if cp.gammas[i] is None else cp.gammas[i][c], cp.flips[i], cp.blurs[i],
                                                      cp.solarize[i], norm))
                drew["jit"] += cp.shifts[i] is not None; drew["blur"] += cp.blurs[i] is not None
                drew["sol"] += cp.solarize[i] is not None; drew["flip"] += cp.flips[i]
                if spec.normalize is not None:
                    drew["norm"] += cp.normalized[i]; drew["nonorm"] += not cp.normalized[i]
                # CustomColorJitter ignores its p (custom_transforms.py:309-311): in the list <=> applied to every crop
                assert (cp.shifts[i] is not None) == bool(spec.jitter_prob)
            ref = np.stack(ref_planes)[:, None]
            got = crops[k].cpu().numpy()
            # solarize is discontinuous: a value within round-off of the threshold may land on the other side -> compare away from it
            near = np.zeros_like(ref, dtype=bool)
            if any(t is not None for t in cp.solarize):
                near = np.abs(np.abs(got - ref)) > 0.05
                assert near.mean() < 1e-4
            scale = 1.0 if spec.normalize is None else 5.0
            np.testing.assert_allclose(got[~near], ref[~near], atol=2e-5 * scale, rtol=0)
            k += 1
    assert all(v > 0 for v in drew.values()), drew   # every transform really fired somewhere
    # same layout as the reference collate of per-image (C, S, S) crops (channels_strategies.py:31-85)
    per_image = []
    off = 0
    for im in imgs:
        per_image.append(([crops[j][off:off + im.shape[0], 0].cpu() for j in range(6)], 0))
        off += im.shape[0]
    cc, _, ncl2 = R.collate(per_image)
    assert ncl2 == ncl and all(torch.equal(a, b.cpu()) for a, b in zip(cc, crops))


def _against_oracle(pipe, imgs, atol=2e-5):
    """Every crop of `pipe(imgs)` against oracle/augment_ref.augment_plane on the parameters the pipeline drew."""
    crops, _, _ = pipe(imgs)
    crops = crops if isinstance(crops, (list, tuple)) else [crops]
    k = 0
    for spec in pipe.specs:
        for _ in range(spec.num_crops):
            cp = pipe.last_params[k]
            ref = np.stack([A.augment_plane(im[c], spec.crop_size, cp.boxes[i], None if cp.shifts[i] is None else cp.shifts[i][c],
                                            None if cp.gammas[i] is None else cp.gammas[i][c], cp.flips[i], cp.blurs[i], cp.solarize[i], None)
                            for i, im in enumerate(imgs) for c in range(im.shape[0])])[:, None]
            np.testing.assert_allclose(crops[k].cpu().numpy(), ref, atol=atol, rtol=0)
            k += 1
    return crops


def test_every_path_of_the_resize_and_blur_kernels_vs_oracle():
    """chadavit_crop_resize stages the source rows of a band of output rows in LDS when they fit 32 KB and reads its taps from global
    memory otherwise; chadavit_blur_finish has a quad path (S % 4 == 0, aligned) and an element path.  Each of them, plus the size-preserving
    copy path, small and odd output sizes, flips and every blur width, against the oracle's restatement of cv2's arithmetic."""
    from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(7)
    # large planes scaled down by 3-5x: a 16-row band of a 224-pixel crop needs ~70 source rows of 700-1000 pixels -> the global-tap path
    big = [rs.rand(2, 700, 900).astype(np.float32), rs.rand(1, 1024, 1000).astype(np.float32)]
    _against_oracle(DeviceMultiCropPipeline([CropSpec(224, 2, crop_min_scale=0.5, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=1.0, flip_prob=0.5),
                                             CropSpec(96, 1, crop_min_scale=0.3, crop_max_scale=1.0, blur_prob=0.5, flip_prob=0.5),
                                             CropSpec(97, 1, crop_min_scale=0.5, crop_max_scale=1.0, blur_prob=1.0, flip_prob=0.5)], dev, seed=2), big)
    # small planes scaled up and down around 1x: the LDS-staged path; odd and tiny output sizes: the element paths of both kernels
    small = _batch(4)
    _against_oracle(DeviceMultiCropPipeline([CropSpec(224, 1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=1.0, flip_prob=0.5),
                                             CropSpec(96, 2, crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, blur_prob=1.0, flip_prob=0.5),
                                             CropSpec(30, 1, crop_min_scale=0.1, crop_max_scale=1.0, blur_prob=1.0, flip_prob=0.5),
                                             CropSpec(9, 1, blur_prob=1.0), CropSpec(8, 1, jitter_prob=0.8, blur_prob=1.0), CropSpec(5, 1, blur_prob=1.0),
                                             CropSpec(512, 1, jitter_prob=0.8, blur_prob=1.0, flip_prob=0.5)], dev, seed=3), small)
    # size-preserving Resize (cv2.resize returns a copy): 224-pixel planes without RandomResizedCrop, with and without a flip
    same = [rs.rand(3, 224, 224).astype(np.float32), rs.rand(1, 224, 224).astype(np.float32)]
    crops = _against_oracle(DeviceMultiCropPipeline([CropSpec(224, 2, rrc_enabled=False, flip_prob=0.5, jitter_prob=0.8)], dev, seed=1), same, atol=1e-6)
    assert len(crops) == 2
    # limits of the entry points: loud, not silent
    from chadavit_amd import ops
    with pytest.raises(RuntimeError):
        ops.blur_finish(torch.zeros(1, 1, 2, 2, device=dev), torch.zeros(1, 12, device=dev))          # S < 4
    with pytest.raises(RuntimeError):
        ops.crop_resize(torch.zeros(16, device=dev), torch.tensor([[0, 4, 4, 0, 0, 4, 4, 0]], device=dev), 1028)   # S > 1024


def test_integer_source_planes_equal_their_float_cast():
    """Planes uploaded as the files store them (uint8 / uint16, `IDRCell100K.read_planes(raw=True)`) and converted on the device give
    bit for bit the crops of the same values cast to float32 on the host (the reference reader's `.astype(np.float32)`,
    custom_datasets.py:190) -- LDS-staged bands, global-tap bands (large planes) and the size-preserving copy path; a batch that mixes
    storage types travels as float32."""
    from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(11)
    specs = [CropSpec(224, 1, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5, int_min_shift=-40, int_max_shift=40),
             CropSpec(96, 2, crop_min_scale=0.05, crop_max_scale=0.4, blur_prob=0.5, flip_prob=0.5),
             CropSpec(97, 1, crop_min_scale=0.5, crop_max_scale=1.0),
             CropSpec(224, 1, rrc_enabled=False, flip_prob=0.5)]
    for dt, hi in ((np.uint8, 256), (np.uint16, 65536)):
        imgs = [rs.randint(0, hi, size=s).astype(dt) for s in ((3, 224, 224), (1, 150, 201), (2, 700, 900), (5, 97, 131), (1, 224, 224))]
        a = DeviceMultiCropPipeline(specs, dev, seed=4)(imgs)[0]
        b = DeviceMultiCropPipeline(specs, dev, seed=4)([im.astype(np.float32) for im in imgs])[0]
        assert len(a) == len(b) == 5
        for x, y in zip(a, b):
            assert x.dtype == torch.float32 and torch.equal(x, y) and float(x.abs().max()) >= 1.0   # (the jittered crop is clamped to [0, 1])
    mixed = [rs.randint(0, 256, size=(2, 64, 64)).astype(np.uint8), rs.randint(0, 65536, size=(1, 64, 64)).astype(np.uint16)]
    a = DeviceMultiCropPipeline(specs[:2], dev, seed=5)(mixed)[0]
    b = DeviceMultiCropPipeline(specs[:2], dev, seed=5)([im.astype(np.float32) for im in mixed])[0]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    from chadavit_amd import ops
    with pytest.raises(RuntimeError, match="float32, uint8 or uint16"):
        ops.crop_resize(torch.zeros(16, device=dev, dtype=torch.int32), torch.tensor([[0, 4, 4, 0, 0, 4, 4, 0]], device=dev), 4)


def test_gray_and_draw_order_on_three_channel_samples():
    """A.ToGray (pretrain_dataloader.py:303-304) on 3-channel samples, between the jitter and the blur; it raises on any other
    channel count when it fires, as albumentations does.  Also pins the number of Python-`random` draws per sample against the
    oracle's statement of albumentations' order (jitter: none)."""
    import random
    from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(4)
    imgs = [rs.rand(3, 120 + 7 * i, 131).astype(np.float32) for i in range(6)]
    spec = CropSpec(crop_size=96, num_crops=2, jitter_prob=0.8, gray_prob=0.5, blur_prob=0.5, flip_prob=0.5,
                    normalize=([0.4, 0.5, 0.6], [0.2, 0.25, 0.3], 1.0), normalize_prob=0.5)
    pipe = DeviceMultiCropPipeline([spec], dev, seed=9)
    crops, _, _ = pipe(imgs)
    fired = 0
    for k in range(2):
        cp = pipe.last_params[k]
        ref = []
        for i, im in enumerate(imgs):
            norms = [(spec.normalize[0][c], spec.normalize[1][c], 1.0) for c in range(3)] if cp.normalized[i] else None
            ref.append(A.augment_sample(im, 96, cp.boxes[i], cp.shifts[i], cp.gammas[i], cp.grays[i], cp.flips[i], cp.blurs[i], None, norms))
            fired += cp.grays[i]
            if cp.grays[i] and not cp.normalized[i]:
                got = crops[k][3 * i:3 * i + 3, 0]
                assert torch.equal(got[0], got[1]) and torch.equal(got[1], got[2])
        np.testing.assert_allclose(crops[k][:, 0].cpu().numpy(), np.concatenate(ref), atol=1e-4, rtol=0)
    assert 0 < fired < 12
    # replaying the host draws with a counting RNG: per sample = crop p + crop params + gray p + blur p (+2) + flip p + ToTensor p
    # + normalize p -- and nothing for the jitter
    class Counting(random.Random):
        n = 0
        def random(self):
            Counting.n += 1
            return super().random()
    pipe2 = DeviceMultiCropPipeline([spec], dev, seed=9)
    pipe2.rng = Counting(9)
    cp = pipe2._draw(spec, [(3, 64, 64)])
    fixed = len([d for d in A.draw_order(True, True, True, False, True, True) if d.endswith(":p")])
    assert fixed == 6 and Counting.n >= fixed
    with pytest.raises(RuntimeError, match="ToGray fired"):
        p3 = DeviceMultiCropPipeline([CropSpec(crop_size=96, num_crops=1, gray_prob=1.0)], dev, seed=0)
        p3([rs.rand(5, 64, 64).astype(np.float32)])


def test_pipeline_feeds_the_training_step():
    """Real-data shaped path end to end: raw planes -> device pipeline -> DINO.training_step (no host-side crop tensors)."""
    from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    from tests.test_model_gpu import _cfg
    dev = torch.device("cuda:0")
    pipe = DeviceMultiCropPipeline([CropSpec(224, 2, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5),
                                    CropSpec(96, 2, crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, flip_prob=0.5)], dev, seed=1)
    model = DINO(_cfg(192, 4096, 2, 2)).to(dev)
    tr = Trainer(max_epochs=2, steps_per_epoch=4).attach(model)
    losses = [tr.train_step(pipe(_batch(s)), s).item() for s in range(2)]
    assert all(np.isfinite(losses)) and abs(losses[0] - np.log(4096)) < 1.0, losses
    # the crops of one resolution leave the pipeline back to back in one buffer: the step takes them as a view, not a cat copy
    from chadavit_amd.data.channels_strategies import adjacent_view
    crops, _, _ = pipe(_batch(5))
    g, l = adjacent_view(crops[:2]), adjacent_view(crops[2:])
    assert g is not None and l is not None and g.data_ptr() == crops[0].data_ptr()
    assert torch.equal(g, torch.cat(crops[:2])) and torch.equal(l, torch.cat(crops[2:]))


def test_prefetcher_kernel_placements_give_identical_batches():
    """DevicePrefetcher(kernels_on="producer") -- augmentation kernels beside the step on the prefetcher's stream -- and
    kernels_on="consumer" -- only the copies run ahead, the kernels are launched at the head of the consumer's stream when the batch is
    handed over -- must hand out bit-identical crops (same seeds, same descriptors, same kernels; only the stream differs), also while the
    consumer's stream is busy and the allocator re-issues blocks between batches."""
    from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
    from chadavit_amd.data.loader import DevicePrefetcher, InMemoryPlanes
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(3)
    ds = InMemoryPlanes([rs.rand(c, 72, 80).astype(np.float32) for c in (3, 1, 2, 5, 1, 3, 2, 4, 1, 2, 3, 1)])
    batches = [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [3, 2, 1, 0]]
    specs = [CropSpec(64, 2, jitter_prob=0.8, blur_prob=0.6, solarize_prob=0.3, flip_prob=0.5),
             CropSpec(32, 3, crop_min_scale=0.05, crop_max_scale=0.3, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5)]
    got = {}
    for mode in ("producer", "consumer"):
        out = []
        for crops, lab, ncl in DevicePrefetcher(ds, batches, DeviceMultiCropPipeline(specs, dev, seed=11), depth=2, workers=3, kernels_on=mode):
            busy = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)   # keeps the consumer's stream occupied
            out.append(([c.clone() for c in crops], lab.clone(), ncl))
            del busy
        torch.cuda.synchronize()
        got[mode] = out
    assert len(got["producer"]) == len(got["consumer"]) == 4
    for (ca, la, na), (cb, lb, nb) in zip(got["producer"], got["consumer"]):
        assert na == nb and torch.equal(la, lb) and len(ca) == len(cb) == 5
        for a, b in zip(ca, cb):
            assert torch.isfinite(a).all() and torch.equal(a, b)
