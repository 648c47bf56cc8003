"""RNG-free, platform-independent weight / input generator (test infrastructure).

Golden vectors are produced in the build container by filling the *reference's* modules with
these values; the GPU box regenerates bit-identical weights and inputs from the same integer
hash, so no weights are committed ("Keeping fixtures small").

Everything is integer arithmetic on uint64 followed by one exact int->float64 conversion, so the
values are identical on every machine.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(h: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser (vectorised, wrap-around uint64 arithmetic)."""
    h = h.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        h ^= h >> np.uint64(30)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(27)
        h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
    return h


def uniform_pm1(n: int, tag: str, seed: int = 0) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (tag, seed, index)."""
    key = np.uint64(zlib.crc32(tag.encode()) & 0xFFFFFFFF) | (np.uint64(seed & 0xFFFFFFFF) << np.uint64(32))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _mix(idx * np.uint64(0x9E3779B97F4A7C15) + _mix(np.array([key], dtype=np.uint64))[0])
    # top 53 bits -> [0,1)
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return 2.0 * u - 1.0


def tensor(shape: Sequence[int], tag: str, scale: float = 1.0, shift: float = 0.0, seed: int = 0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    v = uniform_pm1(n, tag, seed) * scale + shift
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape))


def gaussish(shape: Sequence[int], tag: str, std: float = 1.0, seed: int = 0) -> torch.Tensor:
    """Sum of 4 uniforms -> roughly normal, unit variance * std (exact, RNG-free)."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.float64)
    for j in range(4):
        acc += uniform_pm1(n, f"{tag}#{j}", seed)
    acc *= std * (np.sqrt(3.0) / 2.0)  # var of U(-1,1) is 1/3 -> sum of 4 has var 4/3
    return torch.from_numpy(acc.astype(np.float32)).reshape(tuple(shape))


# --------------------------------------------------------------------------------------
# state_dict fill
# --------------------------------------------------------------------------------------
def _scale_for(name: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(scale, shift) per parameter kind.  Chosen so activations stay O(1) through 12 post-norm
    blocks and no LayerNorm / bias parameter is at its trivial init (so parity tests see them)."""
    leaf = name.split(".")[-1]
    if ("norm" in name or len(shape) == 1) and leaf == "weight":   # LayerNorm / BatchNorm gamma
        return 0.25, 1.0
    if leaf in ("bias", "in_proj_bias"):
        return 0.10, 0.0
    if leaf in ("cls_token", "channel_token", "pos_embed"):
        return 0.50, 0.0
    if leaf == "weight_g":
        return 0.0, 1.0
    if leaf == "center":
        return 0.05, 0.0
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return float(1.7 / np.sqrt(fan_in)), 0.0
    return 0.1, 0.0


def fill_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, prefix: str = "") -> Dict[str, torch.Tensor]:
    """Deterministic tensors for every (name -> shape).  ``prefix`` is stripped from the hash tag so
    `backbone.x` and `momentum_backbone.x` can be given identical or different values on purpose."""
    out: Dict[str, torch.Tensor] = {}
    for name, shape in shapes.items():
        scale, shift = _scale_for(name, tuple(shape))
        out[name] = tensor(shape, prefix + name, scale, shift, seed)
    return out


def backbone_shapes(embed_dim: int, depth: int = 12, patch: int = 16, img: int = 224, max_channels: int = 10,
                    ffn: int = 2048) -> Dict[str, Tuple[int, ...]]:
    """state_dict layout of reference ChAdaViT (chada_vit.py:136-183)."""
    D = embed_dim
    p = (img // patch) ** 2
    s: Dict[str, Tuple[int, ...]] = {
        "cls_token": (1, 1, D),
        "channel_token": (1, max_channels, 1, D),
        "pos_embed": (1, 1, p + 1, D),
        "token_learner.proj.weight": (D, 1, patch, patch),
        "token_learner.proj.bias": (D,),
    }
    for i in range(depth):
        b = f"blocks.{i}."
        s[b + "self_attn.in_proj_weight"] = (3 * D, D)
        s[b + "self_attn.in_proj_bias"] = (3 * D,)
        s[b + "self_attn.out_proj.weight"] = (D, D)
        s[b + "self_attn.out_proj.bias"] = (D,)
        s[b + "linear1.weight"] = (ffn, D)
        s[b + "linear1.bias"] = (ffn,)
        s[b + "linear2.weight"] = (D, ffn)
        s[b + "linear2.bias"] = (D,)
        s[b + "norm1.weight"] = (D,)
        s[b + "norm1.bias"] = (D,)
        s[b + "norm2.weight"] = (D,)
        s[b + "norm2.bias"] = (D,)
    s["norm.weight"] = (D,)
    s["norm.bias"] = (D,)
    return s


def head_shapes(in_dim: int, hidden: int = 2048, bottleneck: int = 256, prototypes: int = 4096, use_bn: bool = False) -> Dict[str, Tuple[int, ...]]:
    """state_dict layout of reference DINOHead (dino.py:59-84): use_bn=False by default; with use_bn the Sequential gains a
    BatchNorm1d after each of the first two Linears (mlp.1, mlp.4; their running statistics are buffers and keep torch's init)."""
    if use_bn:
        return {
            "mlp.0.weight": (hidden, in_dim), "mlp.0.bias": (hidden,),
            "mlp.1.weight": (hidden,), "mlp.1.bias": (hidden,),
            "mlp.3.weight": (hidden, hidden), "mlp.3.bias": (hidden,),
            "mlp.4.weight": (hidden,), "mlp.4.bias": (hidden,),
            "mlp.6.weight": (bottleneck, hidden), "mlp.6.bias": (bottleneck,),
            "last_layer.weight_g": (prototypes, 1),
            "last_layer.weight_v": (prototypes, bottleneck),
        }
    return {
        "mlp.0.weight": (hidden, in_dim), "mlp.0.bias": (hidden,),
        "mlp.2.weight": (hidden, hidden), "mlp.2.bias": (hidden,),
        "mlp.4.weight": (bottleneck, hidden), "mlp.4.bias": (bottleneck,),
        "last_layer.weight_g": (prototypes, 1),
        "last_layer.weight_v": (prototypes, bottleneck),
    }


# --------------------------------------------------------------------------------------
# synthetic batches (per image `randn(C_i, S, S)`-shaped crops)
# --------------------------------------------------------------------------------------
def make_images(num_channels: Sequence[int], sizes: Sequence[int], seed: int = 0) -> List[Tuple[List[torch.Tensor], int]]:
    """List of ([crop_k (C_i,S_k,S_k)], label) per image -- the input format of
    one_channel_collate_fn (channels_strategies.py:31-85)."""
    batch = []
    for i, c in enumerate(num_channels):
        crops = [gaussish((c, s, s), f"img{i}.crop{k}", 1.0, seed) for k, s in enumerate(sizes)]
        batch.append((crops, i % 7))
    return batch
