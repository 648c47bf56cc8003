"""Multi-crop augmentation on the GPU, emitting the collate layout the tokenizer consumes.

Reference: `build_transform_pipeline` + `NCropAugmentation` + `one_channel_collate_fn`
(src/data/pretrain_dataloader.py:132-154, 272-328; src/data/channels_strategies.py:31-85).  There every crop of every image goes
through albumentations / OpenCV on a DataLoader worker, channel by channel for the jitter, and the collate function then stacks
channels on the batch axis.  Here the host only DRAWS the random parameters (a few dozen numbers per image) and uploads the raw
planes once; two HIP kernels per crop size produce the `(sum C, 1, S, S)` tensors directly:

    chadavit_crop_resize : RandomResizedCrop / Resize (cv2.INTER_CUBIC) -> CustomColorJitter -> HorizontalFlip
    chadavit_blur_finish : GaussianBlur (reflect-101) -> Solarize -> Normalize          (skipped when none of them fires)

Random draws follow the order in which albumentations 1.3.1's Compose consumes Python's `random` / numpy's global RNG for this
pipeline (every transform draws `random.random() < p` first; RandomResizedCrop: area, log-ratio, corner; CustomColorJitter:
np.random.uniform shifts then gammas; GaussianBlur: kernel size, sigma; Solarize: threshold); that order is restated from the
pinned version's published source and is NOT verifiable here (albumentations is absent) -- statistics, not streams, are what the
training contract needs.  ToGray / Equalize of the reference pipeline require 3-channel / uint8 images and are rejected.
"""
from __future__ import annotations

import contextlib
import ctypes
import math
import random
import dataclasses
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib, ops


@dataclass
class CropSpec:
    """One entry of the reference's augmentation list (cfg.augmentations[i]): crop_size, rrc, color_jitter, gaussian_blur,
    solarization, horizontal_flip, normalize, num_crops."""
    crop_size: int = 224
    num_crops: int = 2
    rrc_enabled: bool = True
    crop_min_scale: float = 0.25
    crop_max_scale: float = 1.0
    jitter_prob: float = 0.0            # NON-ZERO = the jitter is applied to EVERY crop (see _draw); the value itself is never drawn against
    int_min_shift: float = -0.3
    int_max_shift: float = 0.3
    gamma_min: float = 0.5
    gamma_max: float = 1.5
    gray_prob: float = 0.0              # A.ToGray(p): 3-channel samples only (anything else raises when it fires, as in albumentations)
    blur_prob: float = 0.0
    blur_limit: Tuple[int, int] = (3, 7)
    sigma_limit: Tuple[float, float] = (0.1, 2.0)
    solarize_prob: float = 0.0
    solarize_threshold: float = 128.0   # albumentations' default (a uint8 scale: never fires on [0, 1] floats)
    solarize_max: float = 1.0           # MAX_VALUES_BY_DTYPE[float32]
    flip_prob: float = 0.0
    normalize: Optional[Tuple[Sequence[float], Sequence[float], float]] = None   # (mean per channel, std per channel, max_pixel_value)
    normalize_prob: float = 1.0         # A.Normalize(..., p=cfg.normalize.prob) (pretrain_dataloader.py:322-323): drawn per sample
    ratio: Tuple[float, float] = (3.0 / 4.0, 4.0 / 3.0)

    @staticmethod
    def from_cfg(aug) -> "CropSpec":
        """From one node of the reference's `augmentations` cfg list (keys of pretrain_dataloader.py:232-255)."""
        g = lambda node, key, default: (node.get(key, default) if hasattr(node, "get") else getattr(node, key, default))
        if g(g(aug, "equalization", {}), "prob", 0):
            # albumentations' Equalize accepts uint8 images only ("Image must have uint8 channel type") and the IDRCell reader hands
            # out float32 planes (custom_datasets.py:199-215): in the reference a firing Equalize raises.  Out of scope, loudly.
            raise RuntimeError("equalization.prob > 0: A.Equalize needs uint8 images; the channel-adaptive path is float32 (the "
                               "reference's own pipeline raises when it fires)")
        cj, rrc = g(aug, "color_jitter", {}), g(aug, "rrc", {})
        norm = g(aug, "normalize", None)
        norm_prob = g(norm, "prob", 1.0) if norm and not isinstance(norm, bool) else 1.0
        return CropSpec(crop_size=g(aug, "crop_size", 224), num_crops=g(aug, "num_crops", 1), rrc_enabled=bool(g(rrc, "enabled", True)),
                        crop_min_scale=g(rrc, "crop_min_scale", 0.08), crop_max_scale=g(rrc, "crop_max_scale", 1.0),
                        jitter_prob=g(cj, "prob", 0.0), int_min_shift=g(cj, "int_min_shift", -0.3), int_max_shift=g(cj, "int_max_shift", 0.3),
                        gamma_min=g(cj, "gamma_min", 0.5), gamma_max=g(cj, "gamma_max", 1.5),
                        gray_prob=g(g(aug, "grayscale", {}), "prob", 0.0), normalize_prob=float(norm_prob),
                        blur_prob=g(g(aug, "gaussian_blur", {}), "prob", 0.0), solarize_prob=g(g(aug, "solarization", {}), "prob", 0.0),
                        flip_prob=g(g(aug, "horizontal_flip", {}), "prob", 0.0),
                        normalize=(g(aug, "mean", None), g(aug, "std", None), 255.0) if norm and g(aug, "mean", None) is not None else None)


def rrc_box(H: int, W: int, scale, ratio, rng: random.Random) -> Tuple[int, int, int, int]:
    """(y0, x0, h, w): albumentations 1.3.1 RandomResizedCrop parameter draw (10 attempts, central fallback).
    (Hot on the producer thread -- 10 crops x batch size calls per batch: `uniform` / `randint` are written out as the arithmetic
    `random.Random` performs on the same draws, `a + (b - a) * random()` and `_randbelow(n + 1)`: same stream, same values.)"""
    area = H * W
    rnd, below = rng.random, rng._randbelow
    s0, ds = scale[0], scale[1] - scale[0]
    l0 = math.log(ratio[0])
    dl = math.log(ratio[1]) - l0
    sqrt, exp = math.sqrt, math.exp
    for _ in range(10):
        target_area = (s0 + ds * rnd()) * area
        aspect = exp(l0 + dl * rnd())
        w = int(round(sqrt(target_area * aspect)))
        h = int(round(sqrt(target_area / aspect)))
        if 0 < w <= W and 0 < h <= H:
            i, j = below(H - h + 1), below(W - w + 1)
            return int((H - h) * (i * 1.0 / (H - h + 1e-10))), int((W - w) * (j * 1.0 / (W - w + 1e-10))), h, w
    in_ratio = W / H
    if in_ratio < min(ratio):
        w, h = W, int(round(W / min(ratio)))
    elif in_ratio > max(ratio):
        h, w = H, int(round(H * max(ratio)))
    else:
        w, h = W, H
    i, j = (H - h) // 2, (W - w) // 2
    return int((H - h) * (i * 1.0 / (H - h + 1e-10))), int((W - w) * (j * 1.0 / (W - w + 1e-10))), h, w


def gaussian_taps(ksize: int, sigma: float) -> np.ndarray:
    """7 centred taps of cv2.getGaussianKernel(ksize, sigma) (zero-padded), float32."""
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    out = np.zeros(7, dtype=np.float32)
    out[:ksize] = (k / k.sum()).astype(np.float32)
    return out


@dataclass
class CropParams:
    """Everything random about one crop of one batch (host side), for replay / tests."""
    boxes: List[Tuple[int, int, int, int]] = field(default_factory=list)     # per image (y0, x0, h, w)
    shifts: List[Optional[np.ndarray]] = field(default_factory=list)         # per image: (C,) or None
    gammas: List[Optional[np.ndarray]] = field(default_factory=list)
    blurs: List[Optional[Tuple[int, float]]] = field(default_factory=list)   # per image (ksize, sigma) or None
    solarize: List[Optional[float]] = field(default_factory=list)            # per image threshold or None
    flips: List[bool] = field(default_factory=list)
    grays: List[bool] = field(default_factory=list)                          # per image: ToGray fired
    normalized: List[bool] = field(default_factory=list)                     # per image: Normalize fired (p = normalize_prob)
    arrays: Optional[dict] = field(default=None, repr=False, compare=False)  # the same draws as numpy arrays (the C draw's outputs), if they exist
    jitter_flat: Optional[tuple] = field(default=None, repr=False, compare=False)  # (shifts, gammas) of all channel images in one array each


class _Geometry:
    """Per-batch index arrays shared by the ten crops of a batch: C / H / W per sample, cumulative channel count with a leading 0, channel
    index inside its sample per channel image."""
    __slots__ = ("n", "C", "H", "W", "hw", "cum", "chan")

    def __init__(self, shapes):
        n = self.n = len(shapes)
        a = np.asarray(shapes, dtype=np.int64).reshape(n, 3)
        self.C, self.H, self.W = a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy()
        self.hw = np.ascontiguousarray(a[:, 1:3])
        self.cum = np.concatenate([[0], np.cumsum(self.C)])
        self.chan = np.arange(int(self.cum[-1]), dtype=np.int64) - np.repeat(self.cum[:-1], self.C)


class DeviceMultiCropPipeline:
    def __init__(self, specs: Sequence[CropSpec], device, seed: int = 0):
        self.specs = list(specs)
        self.device = torch.device(device)
        self.rng = random.Random(seed)            # stands for Python's global `random` (albumentations' source of randomness)
        self.np_rng = np.random.RandomState(seed)  # stands for numpy's global RNG (CustomColorJitter draws from it)
        self.last_params: List[CropParams] = []
        # pinned staging for the raw planes: a ring of three buffers, each guarded by the event of the copy that last read it (a
        # fresh pin_memory() per batch costs more than the copy it feeds)
        self._staging: List[Optional[torch.Tensor]] = [None, None, None]
        self._staging_ev: List[Optional["torch.cuda.Event"]] = [None, None, None]
        self._calls = 0
        self._copy_pool = None
        self.python_draws = False   # True: the per-sample draw loop in Python even for a plain random.Random (A/B, tests)

    @property
    def num_crops(self) -> int:
        return sum(s.num_crops for s in self.specs)

    def _draw(self, spec: CropSpec, shapes: Sequence[Tuple[int, int, int]], geo: Optional[_Geometry] = None) -> CropParams:
        """The random parameters of one crop of every sample, drawn in the order albumentations 1.3.1 consumes Python's `random`
        inside `Compose.__call__` for the reference's list (pretrain_dataloader.py:281-326).  Every `BasicTransform.__call__`
        evaluates `random.random() < p` first -- also for p = 1.0 and for always_apply transforms (ToTensorV2) --, EXCEPT the
        reference's own `CustomColorJitter`: it overrides `__call__` (custom_transforms.py:309-311) and calls `apply` directly, so
        its `p` is never consulted and no draw is consumed: with `color_jitter.prob` non-zero EVERY crop is jittered (the
        shipped 0.8 behaves as 1.0).  Reproduced as is."""
        p = CropParams()
        if spec.jitter_prob:
            # numpy's stream, the samples' (shifts, gammas) pairs in order: ONE block of uniform doubles, cut as the per-sample
            # `uniform(lo, hi, C)` calls would consume it (lo + (hi - lo) * u, RandomState's own arithmetic) -- same values, one call
            # instead of two per sample, and the arithmetic on the whole block (the per-sample entries of the lists are views)
            geo = geo if geo is not None else _Geometry(shapes)
            C, cum, chan = geo.C, geo.cum, geo.chan
            u = self.np_rng.random_sample(2 * int(cum[-1]))
            at = np.repeat(2 * cum[:-1], C) + chan                     # sample i's shifts start at 2 * (channels before it), its gammas C_i later
            shift = spec.int_min_shift + (spec.int_max_shift - spec.int_min_shift) * u[at]
            gamma = spec.gamma_min + (spec.gamma_max - spec.gamma_min) * u[at + np.repeat(C, C)]
            bounds = cum.tolist()
            p.shifts = [shift[a:b] for a, b in zip(bounds[:-1], bounds[1:])]
            p.gammas = [gamma[a:b] for a, b in zip(bounds[:-1], bounds[1:])]
            p.jitter_flat = (shift, gamma)
        else:
            p.shifts = [None] * len(shapes)
            p.gammas = [None] * len(shapes)
        if type(self.rng) is random.Random and not self.python_draws:
            return self._draw_native(spec, shapes, p, geo if geo is not None else _Geometry(shapes))
        rng = self.rng
        rnd = rng.random
        scale = (spec.crop_min_scale, spec.crop_max_scale)
        gray_p, blur_p, sol_p, flip_p, norm_on, norm_p = spec.gray_prob, spec.blur_prob, spec.solarize_prob, spec.flip_prob, spec.normalize is not None, spec.normalize_prob
        boxes, grays, blurs, sols, flips, normed = p.boxes, p.grays, p.blurs, p.solarize, p.flips, p.normalized
        for (C, H, W) in shapes:
            rnd()  # RandomResizedCrop / Resize: p = 1.0, the draw still happens (BasicTransform.__call__)
            boxes.append(rrc_box(H, W, scale, spec.ratio, rng) if spec.rrc_enabled else (0, 0, H, W))
            # (the jitter sits here in the reference's list: in it at all <=> prob != 0 (pretrain_dataloader.py:301), then unconditional and
            #  without a draw from THIS stream -- its numpy draws were taken above)
            gray = bool(gray_p and rnd() < gray_p)
            if gray and C != 3:
                raise RuntimeError(f"ToGray fired on a {C}-channel sample: albumentations raises TypeError there (3-channel images only)")
            grays.append(gray)
            if blur_p and rnd() < blur_p:
                k = rng.randrange(spec.blur_limit[0], spec.blur_limit[1] + 1)
                if k != 0 and k % 2 != 1:
                    k = (k + 1) % (spec.blur_limit[1] + 1)
                blurs.append((k, rng.uniform(*spec.sigma_limit)))
            else:
                blurs.append(None)
            if sol_p and rnd() < sol_p:
                sols.append(rng.uniform(spec.solarize_threshold, spec.solarize_threshold))
            else:
                sols.append(None)
            flips.append(bool(flip_p and rnd() < flip_p))
            rnd()  # ToTensorV2(always_apply=True): `random.random() < p or always_apply` still draws
            normed.append(bool(norm_on and rnd() < norm_p))
        return p

    def _draw_native(self, spec: CropSpec, shapes, p: CropParams, geo: _Geometry) -> CropParams:
        """The per-sample loop of `_draw` in C (`chadavit_draw_crop_params`, csrc/host_draw.hip): continues `self.rng` from its exported
        Mersenne-Twister state and writes the state back -- same stream, same values as the Python loop below (held equal in
        tests/test_augment_cpu.py), ~1 ms instead of ~130 ms per 1 024-image 10-crop batch, and not under the interpreter lock the
        training thread needs for its launches.  Any other generator object (a subclass that counts draws, say) takes the Python loop."""
        n = len(shapes)
        version, mt, gauss = self.rng.getstate()
        st = np.array(mt, dtype=np.uint32)
        hw = geo.hw
        boxes = np.empty((n, 4), dtype=np.int64)
        gray, blur_k, sol_on, flip, normed = (np.empty(n, dtype=np.int32) for _ in range(5))
        blur_sigma, sol_value = np.empty(n, dtype=np.float64), np.empty(n, dtype=np.float64)
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data)
        dbl = ctypes.c_double
        rc = _lib.lib().chadavit_draw_crop_params(
            ptr(st), ctypes.c_int(n), ptr(hw), ctypes.c_int(bool(spec.rrc_enabled)), dbl(spec.crop_min_scale), dbl(spec.crop_max_scale),
            dbl(spec.ratio[0]), dbl(spec.ratio[1]), dbl(spec.gray_prob), dbl(spec.blur_prob), ctypes.c_int(spec.blur_limit[0]),
            ctypes.c_int(spec.blur_limit[1]), dbl(spec.sigma_limit[0]), dbl(spec.sigma_limit[1]), dbl(spec.solarize_prob),
            dbl(spec.solarize_threshold), dbl(spec.flip_prob), ctypes.c_int(spec.normalize is not None), dbl(spec.normalize_prob),
            ptr(boxes), ptr(gray), ptr(blur_k), ptr(blur_sigma), ptr(sol_on), ptr(sol_value), ptr(flip), ptr(normed))
        if rc != 0:
            raise RuntimeError(f"chadavit_draw_crop_params failed with code {rc}")
        self.rng.setstate((version, tuple(st.tolist()), gauss))
        C = geo.C
        if gray.any() and (C[gray != 0] != 3).any():
            i = int(np.nonzero((gray != 0) & (C != 3))[0][0])
            raise RuntimeError(f"ToGray fired on a {int(C[i])}-channel sample: albumentations raises TypeError there (3-channel images only)")
        p.boxes = list(map(tuple, boxes.tolist()))
        p.grays = (gray != 0).tolist()
        p.blurs = [None if k < 0 else (k, sg) for k, sg in zip(blur_k.tolist(), blur_sigma.tolist())]
        p.solarize = [v if on else None for on, v in zip(sol_on.tolist(), sol_value.tolist())]
        p.flips = (flip != 0).tolist()
        p.normalized = (normed != 0).tolist()
        p.arrays = {"boxes": boxes, "gray": gray != 0, "blur_k": blur_k, "blur_sigma": blur_sigma, "sol_on": sol_on != 0,
                    "sol_value": sol_value, "flip": flip != 0, "normed": normed != 0}
        return p

    def __call__(self, images: Sequence[np.ndarray], labels: Optional[Sequence[int]] = None, params: Optional[List[CropParams]] = None,
                 defer: bool = False):
        """images: per sample an array (C_i, H_i, W_i) of channel planes (sizes may differ between samples): float32, or uint8 / uint16 as
        the files store them (`IDRCell100K.read_planes(i, raw=True)`) -- a batch whose samples all share one of the two integer types is
        uploaded in it, anything else as float32.
        Returns what `one_channel_collate_fn` returns for the same batch: (crops, labels, list_num_channels) with
        crops[k] (sum C, 1, S_k, S_k) fp32 on the device.

        defer = True: only the COPIES happen here (raw planes, descriptor tables -- copy-engine work on whatever stream is current);
        the crop / blur kernels are handed back as a fourth element `launch(stream)`, to be called once, on the stream that will
        read the crops, after that stream waits for this call's copies.  `DevicePrefetcher` uses it to run the augmentation kernels
        at the head of the training step's own stream instead of beside the step on a side stream, where their blocks take CU
        slots away from kernels that are tuned to fill every one of them."""
        # planes travel as float32, or -- when every sample of the batch is stored that way -- as the 8 / 16-bit unsigned integers the image
        # files hold: a quarter / half of the staging copy and of the PCIe traffic, converted on the device (exact, = the reference
        # reader's `.astype(np.float32)`, custom_datasets.py:190)
        kinds = {np.asarray(im).dtype for im in images}
        dt = kinds.pop() if len(kinds) == 1 and next(iter(kinds)) in (np.dtype(np.uint8), np.dtype(np.uint16)) else np.dtype(np.float32)
        planes = [np.ascontiguousarray(im, dtype=dt) for im in images]
        shapes = [tuple(im.shape) for im in planes]
        nch = [s[0] for s in shapes]
        offs, tot = [], 0
        for (C, H, W) in shapes:
            offs.append(tot)
            tot += C * H * W
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.uint16}[dt]
        if self.device.type == "cuda":
            k = self._calls % 3
            self._calls += 1
            if self._staging_ev[k] is not None:
                self._staging_ev[k].synchronize()   # (three batches ago: long complete)
            nbytes = tot * dt.itemsize
            if self._staging[k] is None or self._staging[k].numel() < nbytes:
                self._staging[k] = torch.empty(max((nbytes + 15) // 16 * 16, 16), dtype=torch.uint8).pin_memory()   # bytes: any plane type
            host = self._staging[k][:nbytes].view(tdt)
        else:
            host = torch.empty(tot, dtype=tdt)
        hn = host.numpy()
        # the raw planes into the pinned staging buffer: 0.8 MB per 3-channel 256 x 256 float image, 0.8 GB per 1024-image batch -- one
        # memcpy stream from the producer thread was the slowest stage of the path (~100 ms per batch); numpy releases the GIL inside large
        # copies, so a few helper threads run them side by side
        def put(lo, hi):
            for o, im in zip(offs[lo:hi], planes[lo:hi]):
                hn[o:o + im.size] = im.reshape(-1)
        n_img = len(planes)
        if tot * dt.itemsize >= (1 << 26) and n_img >= 16:
            if self._copy_pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._copy_pool = ThreadPoolExecutor(max_workers=8, thread_name_prefix="chadavit-stage")
            step = (n_img + 7) // 8
            list(self._copy_pool.map(lambda lo: put(lo, min(lo + step, n_img)), range(0, n_img, step)))
        else:
            put(0, n_img)
        src = host.to(self.device, non_blocking=True)
        if self.device.type == "cuda":
            self._staging_ev[k] = torch.cuda.Event()
            self._staging_ev[k].record(torch.cuda.current_stream(self.device))
        crops, used, pending = [], [], []
        geo = _Geometry(shapes)
        it = iter(params) if params is not None else None
        nchan = sum(nch)
        for spec in self.specs:
            # the crops of one spec (one resolution) go back to back into ONE buffer: DINO.training_step takes "all global
            # crops" / "all local crops" as a view of it (channels_strategies.adjacent_view) instead of a torch.cat copy
            buf = torch.empty((spec.num_crops * nchan, 1, spec.crop_size, spec.crop_size), device=self.device, dtype=torch.float32)
            for k in range(spec.num_crops):
                # replayed parameters: the PUBLIC list fields drive the kernels (a caller may have edited them); the draw's private array
                # forms of the same values are dropped instead of silently taking precedence
                cp = dataclasses.replace(next(it), arrays=None, jitter_flat=None) if it is not None else self._draw(spec, shapes, geo)
                used.append(cp)
                out = buf[k * nchan:(k + 1) * nchan]
                prep = self._prepare_crop(spec, cp, shapes, offs, geo)
                if defer:
                    pending.append((spec, prep, out))
                    crops.append(out)
                else:
                    crops.append(self._launch_crop(spec, prep, src, out=out))
        self.last_params = used
        lab = torch.as_tensor(list(labels) if labels is not None else [-1] * len(planes), dtype=torch.int64, device=self.device)
        res = (crops[0], lab, nch) if len(crops) == 1 else (crops, lab, [list(nch) for _ in crops])   # (channels_strategies.py:81)
        if not defer:
            return res

        def launch(stream=None):
            """The augmentation kernels, on `stream` (default: the current one), which must already wait for this batch's copies."""
            cur = torch.cuda.current_stream(self.device) if stream is None else stream
            if src.is_cuda:   # allocated and filled on the producer's stream, read here
                src.record_stream(cur)
                for _, prep_, _ in pending:
                    for t_ in prep_["device_tensors"]:
                        t_.record_stream(cur)
            with (torch.cuda.stream(cur) if src.is_cuda else contextlib.nullcontext()):
                for spec_, prep_, out_ in pending:
                    self._launch_crop(spec_, prep_, src, out=out_)
            pending.clear()
        return res + (launch,)

    def _run_crop(self, spec: CropSpec, cp: CropParams, src, shapes, offs, out=None) -> torch.Tensor:
        return self._launch_crop(spec, self._prepare_crop(spec, cp, shapes, offs), src, out=out)

    def _prepare_crop(self, spec: CropSpec, cp: CropParams, shapes, offs, geo: Optional[_Geometry] = None) -> dict:
        """Per-channel-image descriptor tables for the two kernels, built with numpy (a Python loop over the ~1500 channel images of
        a 512-image batch, ten crops per batch, was the slowest stage of the whole data path) and uploaded: host work + copies only."""
        n = len(shapes)
        geo = geo if geo is not None else _Geometry(shapes)
        C, H, W, chan = geo.C, geo.H, geo.W, geo.chan
        off = np.asarray(offs, dtype=np.int64)
        if n and int((H * W).max()) >= 1 << 30:
            raise RuntimeError("source planes of 2^30 pixels or more are not supported (chadavit_crop_resize addresses a crop window with 32-bit byte offsets)")
        ar = cp.arrays   # (the C draw's outputs: the same values as the lists, already arrays)
        grays = cp.grays if cp.grays else [False] * n
        if ar is not None:
            normed, has_blur, has_sol = ar["normed"], ar["blur_k"] > 1, ar["sol_on"]
        else:
            normed = np.asarray(cp.normalized if cp.normalized else [spec.normalize is not None] * n, dtype=bool)
            has_blur = np.fromiter((b is not None and b[0] > 1 for b in cp.blurs), dtype=bool, count=n)
            has_sol = np.fromiter((t is not None for t in cp.solarize), dtype=bool, count=n)
        any_jit = any(s_ is not None for s_ in cp.shifts)
        any_fin = bool(normed.any() or has_blur.any() or has_sol.any())
        rep = lambda a: np.repeat(a, C)                                   # per sample -> per channel image
        box = ar["boxes"] if ar is not None else np.asarray(cp.boxes, dtype=np.int64).reshape(n, 4)          # (y0, x0, h, w)
        flips = ar["flip"].astype(np.int64) if ar is not None else np.asarray(cp.flips, dtype=np.int64)
        if n and int(box[:, 2:4].min()) <= 0:
            raise RuntimeError("empty crop window (RandomResizedCrop's fallback rounds to 0 pixels on a plane this small): cv2.resize raises on it too")
        desc = np.stack([rep(off) + chan * rep(H * W), rep(H), rep(W), rep(box[:, 1]), rep(box[:, 0]), rep(box[:, 3]), rep(box[:, 2]),
                         rep(flips)], axis=1)
        dev = self.device
        prep = {"grays": grays if any(grays) else None, "C": C, "n": n, "shift": None, "gamma": None, "fin": None}
        prep["d"] = torch.from_numpy(desc).to(dev, non_blocking=True)
        if any_jit:   # gamma = -1 marks the channel images whose sample did not draw the jitter (no clamp for them)
            if cp.jitter_flat is not None:   # every sample drew it (as _draw does): the two arrays as they are
                shift, gamma = cp.jitter_flat[0].astype(np.float32), cp.jitter_flat[1].astype(np.float32)
            else:
                shift = np.concatenate([np.zeros(c, np.float32) if s_ is None else np.asarray(s_, np.float32) for c, s_ in zip(C, cp.shifts)])
                gamma = np.concatenate([np.full(c, -1.0, np.float32) if g_ is None else np.asarray(g_, np.float32) for c, g_ in zip(C, cp.gammas)])
            prep["shift"], prep["gamma"] = torch.from_numpy(shift).to(dev, non_blocking=True), torch.from_numpy(gamma).to(dev, non_blocking=True)
        if any_fin:
            fin = np.zeros((int(C.sum()), 12), dtype=np.float32)
            fin[:, 8] = np.inf
            fin[:, 9] = spec.solarize_max
            fin[:, 11] = 1.0
            if has_blur.any():
                # gaussian_taps for all blurred samples at once (same arithmetic: float64 exponentials, normalised, rounded to float32)
                idx = np.nonzero(has_blur)[0]
                if ar is not None:
                    ks, sg = ar["blur_k"][idx].astype(np.int64), ar["blur_sigma"][idx]
                else:
                    ks = np.fromiter((cp.blurs[i][0] for i in idx), dtype=np.int64, count=len(idx))
                    sg = np.fromiter((cp.blurs[i][1] for i in idx), dtype=np.float64, count=len(idx))
                if ks.max() > 7:
                    raise RuntimeError("GaussianBlur kernel sizes above 7 are not supported on the device path")
                t = np.arange(7, dtype=np.float64)[None, :] - (ks[:, None] - 1) / 2.0
                kk = np.where(np.arange(7)[None, :] < ks[:, None], np.exp(-(t * t) / (2.0 * sg[:, None] * sg[:, None])), 0.0)
                rows = np.zeros((n, 8), dtype=np.float32)
                rows[idx, 0] = ks
                rows[idx, 1:8] = (kk / kk.sum(1, keepdims=True)).astype(np.float32)
                fin[:, 0:8] = np.repeat(rows, C, axis=0)
            if has_sol.any():
                thr = (np.where(ar["sol_on"], ar["sol_value"], np.inf).astype(np.float32) if ar is not None
                       else np.asarray([np.inf if t is None else t for t in cp.solarize], dtype=np.float32))
                fin[:, 8] = rep(thr)
            if normed.any():
                mean, std, mpv = spec.normalize
                mean, std = np.asarray(mean, np.float32), np.asarray(std, np.float32)
                on = rep(normed)
                fin[on, 10] = (mean[chan % len(mean)] * mpv)[on]
                fin[on, 11] = (1.0 / (std[chan % len(std)] * mpv))[on]
            prep["fin"] = torch.from_numpy(fin).to(dev, non_blocking=True)
        prep["device_tensors"] = [t_ for t_ in (prep["d"], prep["shift"], prep["gamma"], prep["fin"]) if t_ is not None and t_.is_cuda]
        return prep

    def _launch_crop(self, spec: CropSpec, prep: dict, src, out=None) -> torch.Tensor:
        """The kernels of one crop on the current stream: resize (+ jitter, flip), optional ToGray, optional blur / solarize / normalise."""
        first = None if prep["fin"] is not None else out   # the finishing pass reads the resized planes and writes the caller's buffer
        if prep["shift"] is not None:
            res = ops.crop_resize(src, prep["d"], spec.crop_size, prep["shift"], prep["gamma"], out=first)
        else:
            res = ops.crop_resize(src, prep["d"], spec.crop_size, out=first)
        if prep["grays"] is not None:
            # A.ToGray on the (rare) 3-channel samples that drew it: cv2 RGB2GRAY weights, replicated to the three planes
            # (albumentations functional.to_gray).  Sits between the jitter and the blur as in the reference's list; it commutes
            # with the flip fused into the resize pass.  Plain tensor arithmetic on three planes per firing sample.
            w = torch.tensor([0.299, 0.587, 0.114], device=self.device, dtype=torch.float32).view(3, 1, 1, 1)
            c0 = 0
            for i in range(prep["n"]):
                if prep["grays"][i]:
                    res[c0:c0 + 3] = (res[c0:c0 + 3] * w).sum(0, keepdim=True)
                c0 += int(prep["C"][i])
        if prep["fin"] is not None:
            res = ops.blur_finish(res, prep["fin"], out=out)
        return res
