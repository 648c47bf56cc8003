"""CPU restatement (numpy) of the multi-crop augmentation arithmetic of the reference's data path -- TEST INFRASTRUCTURE ONLY.

Reference: `build_transform_pipeline` (src/data/pretrain_dataloader.py:272-328) for the IDRCell100k branch chains
albumentations 1.3.1 / opencv-python 4.7.0.72 transforms (pinned in pyproject.toml:13,21) around the reference's own
`CustomColorJitter` (src/data/custom_transforms.py:301-351):

    RandomResizedCrop(INTER_CUBIC) | Resize(INTER_CUBIC) -> CustomColorJitter -> [ToGray] -> GaussianBlur(sigma 0.1..2) ->
    Solarize -> [Equalize] -> HorizontalFlip -> ToTensorV2 -> Normalize(p)

Gating (albumentations `BasicTransform.__call__`: `random.random() < p` per transform and sample, drawn even for p = 1 and for
always_apply transforms) -- with ONE exception owned by the reference: `CustomColorJitter` overrides `__call__`
(custom_transforms.py:309-311) and applies unconditionally without a draw, so a non-zero `color_jitter.prob` jitters every crop.
`draw_order()` below states the sequence of `random` draws per sample; chadavit_amd/data/device_pipeline.py::_draw follows it.

Neither albumentations nor OpenCV is installed in this image, so **parity of this file is UNPINNED by the
third-party code itself**; each function restates the PUBLISHED algorithm of the pinned version and is anchored by
(a) torch's bicubic interpolation, documented to match OpenCV's INTER_CUBIC (same A = -0.75, half-pixel centres, clamped
border) -- tests/test_augment_cpu.py; (b) closed forms (constant / affine images, delta responses, kernel normalisation);
(c) the reference-owned jitter, pinned by tests/golden/jitter.npz.  Only tests/ may import this module.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

_A = -0.75  # OpenCV's cubic coefficient (imgproc/resize.cpp: interpolateCubic)


def cubic_weights(t: np.ndarray) -> np.ndarray:
    """(..., 4) float32 taps for fractional position t (OpenCV interpolateCubic)."""
    t = np.asarray(t, dtype=np.float32)
    A = np.float32(_A)
    w0 = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
    w1 = ((A + 2) * t - (A + 3)) * t * t + 1
    w2 = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
    w3 = 1 - w0 - w1 - w2
    return np.stack([w0, w1, w2, w3], -1).astype(np.float32)


def resize_cubic(img: np.ndarray, S: int) -> np.ndarray:
    """cv2.resize(img, (S, S), interpolation=cv2.INTER_CUBIC) for a 2-D float32 image: half-pixel centre mapping
    fx = (dx + 0.5) * (w / S) - 0.5, 4 taps at floor(fx) - 1 .. + 2 with indices clamped to the image, rows then columns."""
    img = np.asarray(img, dtype=np.float32)
    h, w = img.shape
    if h == S and w == S:
        return img.copy()

    def taps(n_src, n_dst):
        f = ((np.arange(n_dst) + 0.5) * (n_src / n_dst) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        wts = cubic_weights(f - s)
        idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
        return idx, wts

    xi, xw = taps(w, S)
    yi, yw = taps(h, S)
    rows = (img[:, xi] * xw[None]).sum(-1, dtype=np.float32)        # (h, S): horizontal pass
    return (rows[yi] * yw[:, :, None]).sum(1, dtype=np.float32)     # (S, S): vertical pass


def random_resized_crop_params(H: int, W: int, scale: Tuple[float, float], rng, ratio=(3.0 / 4.0, 4.0 / 3.0)) -> Tuple[int, int, int, int]:
    """(y0, x0, h, w) as albumentations 1.3.1 RandomResizedCrop.get_params_dependent_on_targets draws them from Python's `random`
    (`rng` = a random.Random): 10 attempts of area ~ U(scale) * H W, log-uniform aspect ratio, uniform integer corner; central
    fallback.  The crop corner goes through albumentations' fractional h_start / w_start (functional.get_random_crop_coords)."""
    area = H * W
    for _ in range(10):
        target_area = rng.uniform(*scale) * area
        log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
        aspect = math.exp(rng.uniform(*log_ratio))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= W and 0 < h <= H:
            i = rng.randint(0, H - h)
            j = rng.randint(0, W - w)
            h_start = i * 1.0 / (H - h + 1e-10)
            w_start = j * 1.0 / (W - w + 1e-10)
            return int((H - h) * h_start), int((W - w) * w_start), h, w
    in_ratio = W / H
    if in_ratio < min(ratio):
        w = W
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = H
        w = int(round(h * max(ratio)))
    else:
        w, h = W, H
    i, j = (H - h) // 2, (W - w) // 2
    h_start = i * 1.0 / (H - h + 1e-10)
    w_start = j * 1.0 / (W - w + 1e-10)
    return int((H - h) * h_start), int((W - w) * w_start), h, w


def gaussian_kernel1d(ksize: int, sigma: float) -> np.ndarray:
    """cv2.getGaussianKernel(ksize, sigma) for sigma > 0: exp(-(i - (k-1)/2)^2 / (2 sigma^2)), normalised to sum 1 (float32 as
    cv2.GaussianBlur uses for CV_32F images)."""
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def gaussian_blur(img: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """cv2.GaussianBlur(img, (ksize, ksize), sigma) on a 2-D float32 image: separable, BORDER_REFLECT_101 (cv2's default)."""
    if ksize <= 1:
        return np.asarray(img, dtype=np.float32).copy()
    k = gaussian_kernel1d(ksize, sigma)
    r = ksize // 2
    p = np.pad(np.asarray(img, dtype=np.float32), r, mode="reflect")   # numpy 'reflect' = d c b | a b c d | c b a = REFLECT_101
    rows = sum(k[i] * p[:, i:i + img.shape[1]] for i in range(ksize))
    return sum(k[j] * rows[j:j + img.shape[0], :] for j in range(ksize)).astype(np.float32)


def solarize(img: np.ndarray, threshold: float, max_val: float = 1.0) -> np.ndarray:
    """albumentations.functional.solarize for float images: values >= threshold become max_val - value.  NB the transform's
    default threshold is 128 (a uint8 scale): on [0, 1] float images, which is what IDRCell100k loads, it never fires."""
    out = np.asarray(img, dtype=np.float32).copy()
    m = out >= threshold
    out[m] = max_val - out[m]
    return out


def normalize(img: np.ndarray, mean: float, std: float, max_pixel_value: float = 255.0) -> np.ndarray:
    """albumentations.Normalize per channel: (img - mean * max_pixel_value) / (std * max_pixel_value)."""
    return ((np.asarray(img, dtype=np.float32) - np.float32(mean * max_pixel_value)) * np.float32(1.0 / (std * max_pixel_value))).astype(np.float32)


def color_jitter(plane: np.ndarray, shift: float, gamma: float) -> np.ndarray:
    """One channel of CustomColorJitter.apply (custom_transforms.py:327-345): clip(gamma * (x + shift), 0, 1)."""
    return np.clip(np.float32(gamma) * (np.asarray(plane, dtype=np.float32) + np.float32(shift)), 0.0, 1.0).astype(np.float32)


def draw_order(jitter: bool, gray: bool, blur: bool, solarize: bool, flip: bool, normalize: bool) -> List[str]:
    """Names of the Python-`random` draws one sample consumes, in order, for a list built by build_transform_pipeline
    (a transform is in the list iff its cfg prob is non-zero; the crop / ToTensorV2 always are).  The jitter never draws from
    `random` (its parameters come from numpy's global RNG inside `apply`)."""
    seq = ["crop:p", "crop:params..."]
    if gray:
        seq.append("gray:p")
    if blur:
        seq += ["blur:p", "blur:ksize,sigma (if fired)"]
    if solarize:
        seq += ["solarize:p", "solarize:threshold (if fired)"]
    if flip:
        seq.append("flip:p")
    seq.append("to_tensor:p")
    if normalize:
        seq.append("normalize:p")
    return seq


def to_gray(planes: np.ndarray) -> np.ndarray:
    """albumentations.ToGray on a (3, H, W) float image: cv2.cvtColor RGB2GRAY (0.299, 0.587, 0.114) then GRAY2RGB."""
    g = (np.float32(0.299) * planes[0] + np.float32(0.587) * planes[1] + np.float32(0.114) * planes[2]).astype(np.float32)
    return np.stack([g, g, g])


def augment_plane(plane: np.ndarray, S: int, box: Tuple[int, int, int, int], shift: Optional[float] = None, gamma: Optional[float] = None,
                  flip: bool = False, blur: Optional[Tuple[int, float]] = None, sol_threshold: Optional[float] = None,
                  norm: Optional[Tuple[float, float, float]] = None) -> np.ndarray:
    """One channel through the whole chain in the reference's order (see the module docstring); box = (y0, x0, h, w)."""
    y0, x0, h, w = box
    out = resize_cubic(plane[y0:y0 + h, x0:x0 + w], S)
    if shift is not None:
        out = color_jitter(out, shift, gamma)
    if blur is not None:
        out = gaussian_blur(out, blur[0], blur[1])
    if sol_threshold is not None:
        out = solarize(out, sol_threshold)
    if flip:
        out = out[:, ::-1]
    if norm is not None:
        out = normalize(out, *norm)
    return np.ascontiguousarray(out, dtype=np.float32)


def augment_sample(planes: np.ndarray, S: int, box: Tuple[int, int, int, int], shifts=None, gammas=None, gray: bool = False,
                   flip: bool = False, blur: Optional[Tuple[int, float]] = None, sol_threshold: Optional[float] = None,
                   norms: Optional[Sequence[Tuple[float, float, float]]] = None) -> np.ndarray:
    """All channels of one sample through the chain, including the one step that mixes channels (ToGray, 3-channel samples):
    resize -> jitter (per channel) -> [gray] -> blur -> solarize -> flip -> [normalize (per channel)]."""
    y0, x0, h, w = box
    C = planes.shape[0]
    out = np.stack([resize_cubic(planes[c][y0:y0 + h, x0:x0 + w], S) for c in range(C)])
    if shifts is not None:
        out = np.stack([color_jitter(out[c], shifts[c], gammas[c]) for c in range(C)])
    if gray:
        assert C == 3, "ToGray is defined for 3-channel images only"
        out = to_gray(out)
    res = []
    for c in range(C):
        o = out[c]
        if blur is not None:
            o = gaussian_blur(o, blur[0], blur[1])
        if sol_threshold is not None:
            o = solarize(o, sol_threshold)
        if flip:
            o = o[:, ::-1]
        if norms is not None:
            o = normalize(o, *norms[c])
        res.append(np.ascontiguousarray(o, dtype=np.float32))
    return np.stack(res)

