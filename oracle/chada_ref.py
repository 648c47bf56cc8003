"""CPU fp32 restatement of the reference hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function cites the reference file:line (relative to /root/reference) it restates.  Two
forms are given for the backbone:

  * ``*_ragged``  -- padding-free packed tokens + cu_seqlens: what the HIP path computes;
  * ``*_padded``  -- 10-channel zero padding + key mask: what the reference executes
                     (used to show ragged == padded, and as the "padded" CPU baseline).

Parameters are passed as plain ``{state_dict key: tensor}`` dicts using the reference's key names
(SURVEY.md section 8(b)), so the same dict can be loaded into the reference modules.
Parity of this file with the reference is pinned by tests/golden/*.npz (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# ======================================================================================
# A1  collate  (src/data/channels_strategies.py:31-85)
# ======================================================================================
def collate(batch):
    """[(..., [crop_k (C_i,H_k,W_k)] | tensor, label)] -> (crops, labels, num_channels).

    crops[k]: (sum C_i, 1, H_k, W_k), image-major then channel (channels_strategies.py:64-66,77-78);
    a bare tensor / list is returned when there is one crop (:81)."""
    first = batch[0][-2:][0]
    num_crops = len(first) if isinstance(first, list) else 1
    crop_lists: List[List[torch.Tensor]] = [[] for _ in range(num_crops)]
    nch: List[List[int]] = [[] for _ in range(num_crops)]
    labels = []
    for item in batch:
        image_list, label = item[-2:]
        if isinstance(image_list, torch.Tensor):
            image_list = [image_list]
        for k, crop in enumerate(image_list):
            nch[k].append(crop.shape[0])
            crop_lists[k].append(crop)  # (C,H,W) rows are already channel-major
        labels.append(label)
    crops = [torch.cat(c, dim=0).unsqueeze(1) for c in crop_lists]
    crops_out = crops[0] if num_crops == 1 else crops
    return crops_out, torch.tensor(labels), nch


# ======================================================================================
# A2/A3  tokenizer  (src/backbones/vit/chada_vit.py:118-134, 185-270)
# ======================================================================================
def patch_pos_embed(p: Params, S: int, patch: int = 16) -> torch.Tensor:
    """(g*g, D) positional rows for the patch tokens of an S x S crop (chada_vit.py:185-217).

    Same-size crops use pos_embed[1:] as is (:200-201); otherwise bicubic with the *scale_factor*
    form and the +0.1 fudge (:206-214) -- `size=` would differ by up to 6.5e-2."""
    pos = p["pos_embed"][0, 0]  # (1+N, D)
    N = pos.shape[0] - 1
    g = S // patch
    if g * g == N:
        return pos[1:]
    D = pos.shape[1]
    n0 = int(math.sqrt(N))
    w0 = g + 0.1
    grid = pos[1:].reshape(1, n0, n0, D).permute(0, 3, 1, 2)
    out = F.interpolate(grid, scale_factor=(w0 / math.sqrt(N), w0 / math.sqrt(N)), mode="bicubic")
    assert out.shape[-1] == g and out.shape[-2] == g  # chada_vit.py:215
    return out.permute(0, 2, 3, 1).reshape(g * g, D)


def cu_seqlens_of(num_channels: Sequence[int], p: int) -> torch.Tensor:
    lens = torch.tensor([1 + c * p for c in num_channels], dtype=torch.int64)
    return torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)])


def tokenize_ragged(p: Params, x: torch.Tensor, num_channels: Sequence[int], patch: int = 16,
                    add_channel_token: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """x (sum C,1,S,S) -> packed tokens (sum_i (1+C_i*g*g), D), cu_seqlens (B+1).

    tok[i, 1 + c*pp + (r*g+q)] = conv(x[off_i+c])[:, r, q] + pos[1+r*g+q] + chan[c];
    tok[i, 0] = cls + pos[0]   (chada_vit.py:223-265)."""
    S = x.shape[-1]
    g = S // patch
    pp = g * g
    W = p["token_learner.proj.weight"]
    b = p["token_learner.proj.bias"]
    t = F.conv2d(x, W, b, stride=patch).flatten(2).transpose(1, 2)  # (sumC, pp, D)  chada_vit.py:131-133
    pos = patch_pos_embed(p, S, patch)  # (pp, D)
    chan = p["channel_token"][0, :, 0]  # (maxC, D)
    cls = p["cls_token"][0, 0] + p["pos_embed"][0, 0, 0]  # chada_vit.py:259-262
    rows = []
    off = 0
    for c_i in num_channels:
        ti = t[off:off + c_i] + pos[None]
        if add_channel_token:
            ti = ti + chan[:c_i, None, :]
        rows.append(cls[None])
        rows.append(ti.reshape(c_i * pp, -1))
        off += c_i
    return torch.cat(rows, 0), cu_seqlens_of(num_channels, pp)


def tokenize_padded(p: Params, x: torch.Tensor, num_channels: Sequence[int], patch: int = 16,
                    max_channels: int = 10, model_max_channels: int = 10):
    """What the reference materialises: (B, 1+max_channels*pp, D) + bool key mask (chada_vit.py:219-270)."""
    S = x.shape[-1]
    g = S // patch
    pp = g * g
    W = p["token_learner.proj.weight"]
    b = p["token_learner.proj.bias"]
    t = F.conv2d(x, W, b, stride=patch).flatten(2).transpose(1, 2)
    chunks = torch.split(t, list(num_channels), dim=0)
    padded = torch.stack([
        torch.cat([c, torch.zeros((max_channels - c.size(0), c.size(1), c.size(2)))], 0) if c.size(0) < max_channels else c
        for c in chunks], 0)  # (B, maxC, pp, D)
    B = padded.shape[0]
    flat = padded.reshape(B, -1, padded.size(3))
    mask = torch.all(flat == 0.0, dim=-1)  # chada_vit.py:239
    padded = padded + patch_pos_embed(p, S, patch)[None, None]
    if max_channels == model_max_channels:  # chada_vit.py:248
        padded = padded + p["channel_token"].expand(B, -1, pp, -1)
    emb = padded.reshape(B, -1, padded.size(3))
    cls = (p["cls_token"] + p["pos_embed"][:, :, 0]).expand(B, -1, -1)
    emb = torch.cat([cls, emb], 1)
    mask = torch.cat([torch.zeros(B, 1, dtype=torch.bool), mask], 1)
    return emb, mask


# ======================================================================================
# A4  transformer block  (chada_vit.py:75-116)
# ======================================================================================
def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _mha_ragged(h: torch.Tensor, cu: torch.Tensor, p: Params, pre: str, nheads: int) -> torch.Tensor:
    D = h.shape[-1]
    dh = D // nheads
    qkv = h @ p[pre + "self_attn.in_proj_weight"].t() + p[pre + "self_attn.in_proj_bias"]
    q, k, v = qkv.split(D, dim=-1)
    outs = []
    for i in range(len(cu) - 1):
        s, e = int(cu[i]), int(cu[i + 1])
        qi = q[s:e].reshape(e - s, nheads, dh).transpose(0, 1)
        ki = k[s:e].reshape(e - s, nheads, dh).transpose(0, 1)
        vi = v[s:e].reshape(e - s, nheads, dh).transpose(0, 1)
        att = torch.softmax(qi @ ki.transpose(1, 2) / math.sqrt(dh), dim=-1)
        outs.append((att @ vi).transpose(0, 1).reshape(e - s, D))
    a = torch.cat(outs, 0)
    return a @ p[pre + "self_attn.out_proj.weight"].t() + p[pre + "self_attn.out_proj.bias"]


def block_ragged(p: Params, i: int, x: torch.Tensor, cu: torch.Tensor, nheads: int = 2, eps: float = 1e-5) -> torch.Tensor:
    """Post-norm block with norm1 applied twice and a ReLU FFN (chada_vit.py:95-100,105-116)."""
    pre = f"blocks.{i}."
    g1, b1 = p[pre + "norm1.weight"], p[pre + "norm1.bias"]
    a = _mha_ragged(_ln(x, g1, b1, eps), cu, p, pre, nheads)         # :96
    x1 = _ln(x + a, g1, b1, eps)                                      # :99
    f = torch.relu(x1 @ p[pre + "linear1.weight"].t() + p[pre + "linear1.bias"])
    f = f @ p[pre + "linear2.weight"].t() + p[pre + "linear2.bias"]  # :115
    return _ln(x1 + f, p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)  # :100


def block_padded(p: Params, i: int, x: torch.Tensor, mask: torch.Tensor, nheads: int = 2, eps: float = 1e-5) -> torch.Tensor:
    pre = f"blocks.{i}."
    B, N, D = x.shape
    dh = D // nheads
    g1, b1 = p[pre + "norm1.weight"], p[pre + "norm1.bias"]
    h = _ln(x, g1, b1, eps)
    qkv = h @ p[pre + "self_attn.in_proj_weight"].t() + p[pre + "self_attn.in_proj_bias"]
    q, k, v = [t.reshape(B, N, nheads, dh).transpose(1, 2) for t in qkv.split(D, -1)]
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, N, D)
    a = a @ p[pre + "self_attn.out_proj.weight"].t() + p[pre + "self_attn.out_proj.bias"]
    x1 = _ln(x + a, g1, b1, eps)
    f = torch.relu(x1 @ p[pre + "linear1.weight"].t() + p[pre + "linear1.bias"])
    f = f @ p[pre + "linear2.weight"].t() + p[pre + "linear2.bias"]
    return _ln(x1 + f, p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)


# ======================================================================================
# A5  backbone forward  (chada_vit.py:272-289)
# ======================================================================================
def depth_of(p: Params) -> int:
    return 1 + max(int(k.split(".")[1]) for k in p if k.startswith("blocks."))


def backbone_ragged(p: Params, x: torch.Tensor, num_channels: Sequence[int], nheads: int = 2, final_eps: float = 1e-6,
                    return_all_tokens: bool = False, patch: int = 16, add_channel_token: bool = True,
                    collect: Optional[list] = None) -> torch.Tensor:
    """Factory configuration: 2 heads, final LN eps 1e-6 (chada_vit.py:333-339)."""
    t, cu = tokenize_ragged(p, x, num_channels, patch, add_channel_token)
    if collect is not None:
        collect.append(t)
    for i in range(depth_of(p)):
        t = block_ragged(p, i, t, cu, nheads)
        if collect is not None:
            collect.append(t)
    t = _ln(t, p["norm.weight"], p["norm.bias"], final_eps)  # :281
    if return_all_tokens:  # :283-287 -- valid non-CLS tokens, image-major
        keep = torch.ones(t.shape[0], dtype=torch.bool)
        keep[cu[:-1]] = False
        return t[keep]
    return t[cu[:-1]]  # :289


def backbone_padded(p: Params, x: torch.Tensor, num_channels: Sequence[int], nheads: int = 2, final_eps: float = 1e-6,
                    return_all_tokens: bool = False, patch: int = 16) -> torch.Tensor:
    t, mask = tokenize_padded(p, x, num_channels, patch)
    for i in range(depth_of(p)):
        t = block_padded(p, i, t, mask, nheads)
    t = _ln(t, p["norm.weight"], p["norm.bias"], final_eps)
    if return_all_tokens:
        return t[:, 1:][~mask[:, 1:]]
    return t[:, 0]

# ======================================================================================
# A5b  attention-map export  (chada_vit.py:313-320; consumer main_attn.py:202-207)
# ======================================================================================
def last_selfattention(p: Params, x: torch.Tensor, nheads: int = 2, patch: int = 16, eps: float = 1e-5) -> torch.Tensor:
    """get_last_selfattention(x): x (B, 1, S, S) one-channel images, tokenised with max_channels=1 (so NO channel token is
    added: 1 != self.max_channels, chada_vit.py:248), blocks 0..depth-2 run normally and the last block returns the per-head
    softmax probabilities of self_attn(norm1(x)) (need_weights, average_attn_weights=False; chada_vit.py:88-92,105-110).
    Returns (B, H, N, N) fp32, N = 1 + (S/patch)^2."""
    B = x.shape[0]
    nch = [1] * B
    t, cu = tokenize_ragged(p, x, nch, patch, add_channel_token=False)
    depth = depth_of(p)
    for i in range(depth - 1):
        t = block_ragged(p, i, t, cu, nheads, eps)
    pre = f"blocks.{depth - 1}."
    D = t.shape[-1]
    dh = D // nheads
    h = _ln(t, p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps)
    qkv = h @ p[pre + "self_attn.in_proj_weight"].t() + p[pre + "self_attn.in_proj_bias"]
    q, k, _ = qkv.split(D, dim=-1)
    N = t.shape[0] // B
    q = q.reshape(B, N, nheads, dh).transpose(1, 2)
    k = k.reshape(B, N, nheads, dh).transpose(1, 2)
    return torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)


# ======================================================================================
# A8  DINO head  (src/methods/dino.py:98-111)
# ======================================================================================
def head_forward(hp: Params, f: torch.Tensor, bn_stats: Optional[Dict[str, torch.Tensor]] = None, training: bool = True) -> torch.Tensor:
    """DINOHead.forward (src/methods/dino.py:98-111).  With `use_bn` (keys mlp.6.* present: Linear, BatchNorm1d, GELU, Linear,
    BatchNorm1d, GELU, Linear -- dino.py:59-77) `bn_stats` holds the running estimates {"mlp.1.running_mean", ...}; they are
    updated in place in training mode, exactly as torch.nn.BatchNorm1d does (momentum 0.1, eps 1e-5)."""
    if "mlp.6.weight" in hp:
        t = f
        for lin, bn in (("mlp.0", "mlp.1"), ("mlp.3", "mlp.4")):
            t = t @ hp[lin + ".weight"].t() + hp[lin + ".bias"]
            rm = bn_stats[bn + ".running_mean"] if bn_stats is not None else None
            rv = bn_stats[bn + ".running_var"] if bn_stats is not None else None
            t = F.gelu(F.batch_norm(t, rm, rv, hp[bn + ".weight"], hp[bn + ".bias"], training or rm is None, 0.1, 1e-5))
        t = t @ hp["mlp.6.weight"].t() + hp["mlp.6.bias"]
    else:
        t = F.gelu(f @ hp["mlp.0.weight"].t() + hp["mlp.0.bias"])
        t = F.gelu(t @ hp["mlp.2.weight"].t() + hp["mlp.2.bias"])
        t = t @ hp["mlp.4.weight"].t() + hp["mlp.4.bias"]
    t = F.normalize(t, dim=-1)  # eps 1e-12
    v = hp["last_layer.weight_v"]
    w = hp["last_layer.weight_g"] * v / v.norm(dim=1, keepdim=True)  # old-style weight_norm, dim=0
    return t @ w.t()


# ======================================================================================
# A9  DINO loss  (src/losses/dino.py:69-118)
# ======================================================================================
def teacher_temp_schedule(warmup_temp: float, temp: float, warmup_epochs: int, num_epochs: int) -> np.ndarray:
    return np.concatenate((np.linspace(warmup_temp, temp, warmup_epochs),
                           np.ones(num_epochs - warmup_epochs) * temp))  # losses/dino.py:62-67


def dino_loss(student: torch.Tensor, teacher: torch.Tensor, center: torch.Tensor, teacher_temp: float,
              student_temp: float = 0.1, n_student_views: int = 2) -> torch.Tensor:
    """losses/dino.py:69-100.  `n_student_views` is the reference's `self.num_large_crops` (:82): 2 in the reference's DINO; the number
    of crops for the standard-DINO multi-crop loss (the build's flagged option: local crops as extra student views)."""
    s = (student / student_temp).chunk(n_student_views)
    q = F.softmax((teacher - center) / teacher_temp, dim=-1).detach().chunk(2)
    total = 0.0
    n = 0
    for iq, qq in enumerate(q):
        for iv, v in enumerate(s):
            if iv == iq:
                continue
            total = total + torch.sum(-qq * F.log_softmax(v, dim=-1), dim=-1).mean()
            n += 1
    return total / n


def center_update(center: torch.Tensor, teacher: torch.Tensor, momentum: float = 0.9, world_sum: Optional[torch.Tensor] = None,
                  world_size: int = 1) -> torch.Tensor:
    """c <- m c + (1-m) * (sum_rows t [all-reduce SUM] / world / len(t))   (losses/dino.py:103-118)."""
    bc = torch.sum(teacher, dim=0, keepdim=True) if world_sum is None else world_sum
    bc = bc / world_size / len(teacher)
    return center * momentum + bc * (1 - momentum)


# ======================================================================================
# A11  EMA / tau   (src/utils/momentum.py:63-87)      A12  LR schedule (lr_scheduler.py:76-125)
# ======================================================================================
def ema_update(student: Params, teacher: Params, tau: float) -> Params:
    return {k: tau * teacher[k] + (1 - tau) * student[k] for k in teacher}


def tau_schedule(step: int, max_steps: int, base_tau: float, final_tau: float) -> float:
    return final_tau - (final_tau - base_tau) * (math.cos(math.pi * step / max_steps) + 1) / 2


def warmup_cosine_lr(step: int, base_lr: float, warmup_steps: float, max_steps: float, warmup_start_lr: float,
                     eta_min: float) -> float:
    """Closed form of LinearWarmupCosineAnnealingLR (lr_scheduler.py:127-149); the chainable get_lr
    (:76-125) produces the same sequence when stepped once per optimiser step."""
    if step < warmup_steps:
        return warmup_start_lr + step * (base_lr - warmup_start_lr) / (warmup_steps - 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (1 + math.cos(math.pi * (step - warmup_steps) / (max_steps - warmup_steps)))


# ======================================================================================
# A6/A7/A10  one DINO training step, reference-parity crop semantics
#            (src/methods/base.py:668-733, 1186-1248; src/methods/dino.py:300-325, 367-376)
# ======================================================================================
def split_prefix(sd: Params, prefix: str) -> Params:
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def training_step(sd: Params, crops: List[torch.Tensor], num_channels: List[List[int]], num_large_crops: int,
                  teacher_temp: float, student_temp: float = 0.1, nheads: int = 2, padded: bool = False,
                  freeze_last_layer: bool = True, clip_grad: float = 0.0, norm_last_layer: bool = True, standard_multicrop: bool = False):
    """Student fwd on global crops (-> z) and on local crops (backbone only, result unused: DINO does
    not override multicrop_forward, base.py:566-620); teacher fwd on global crops; loss over the 2x2
    cross pairs; backward.  Returns (loss, grads{name: tensor|None}, new_center, aux).

    ``sd`` uses the method-level keys ``backbone.*``, ``momentum_backbone.*``, ``head.*``,
    ``momentum_head.*``, ``dino_loss_func.center``."""
    bb = {k: v.detach().clone().requires_grad_(True) for k, v in split_prefix(sd, "backbone.").items()}
    hd_all, thd_all = split_prefix(sd, "head."), split_prefix(sd, "momentum_head.")
    is_buf = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked"))
    # (norm_last_layer, dino.py:83-84: the prototypes' magnitudes weight_g are frozen at their initial 1 -- or trained like the rest)
    hd = {k: v.detach().clone().requires_grad_(k != "last_layer.weight_g" or not norm_last_layer) for k, v in hd_all.items() if not is_buf(k)}
    tbb = split_prefix(sd, "momentum_backbone.")
    thd = {k: v for k, v in thd_all.items() if not is_buf(k)}
    # BatchNorm running estimates of the two heads (use_bn_in_head): cloned, updated once per head call (= per global crop)
    hbn = {k: v.detach().clone() for k, v in hd_all.items() if is_buf(k) and v.is_floating_point()} or None
    tbn = {k: v.detach().clone() for k, v in thd_all.items() if is_buf(k) and v.is_floating_point()} or None
    fwd = backbone_padded if padded else backbone_ragged
    feats, z = [], []
    for k in range(num_large_crops):
        f = fwd(bb, crops[k], num_channels[k], nheads)
        feats.append(f)
        z.append(head_forward(hd, f, hbn))
    if standard_multicrop:
        # NOT the reference's default: the standard-DINO multi-crop loss (the reference's own DINOLoss arithmetic with num_large_crops =
        # number of crops, and a multicrop_forward that adds the head's output) -- local crops with gradient, through the head, into the loss
        for k in range(num_large_crops, len(crops)):
            f = fwd(bb, crops[k], num_channels[k], nheads)
            feats.append(f)
            z.append(head_forward(hd, f, hbn))
    with torch.no_grad():
        for k in range(num_large_crops, len(crops)):  # local crops: forward only, no loss
            if not standard_multicrop:
                feats.append(fwd(bb, crops[k], num_channels[k], nheads))
        tfeats = [fwd(tbb, crops[k], num_channels[k], nheads) for k in range(num_large_crops)]
        tz = [head_forward(thd, f, tbn) for f in tfeats]
    p_s = torch.cat(z)
    p_t = torch.cat(tz)
    center = sd["dino_loss_func.center"]
    loss = dino_loss(p_s, p_t, center, teacher_temp, student_temp, n_student_views=len(z))
    loss.backward()
    grads: Dict[str, Optional[torch.Tensor]] = {}
    for k, v in bb.items():
        g = v.grad
        if g is not None and clip_grad:  # dino.py:256-261 -- per-parameter clip, backbone only
            nrm = g.norm(2)
            coef = clip_grad / (nrm + 1e-6)
            if coef < 1:
                g = g * coef
        grads["backbone." + k] = g
    for k, v in hd.items():
        g = v.grad
        if freeze_last_layer and k.startswith("last_layer."):  # dino.py:374-376
            g = None
        grads["head." + k] = g
    new_center = center_update(center, p_t)
    # what the passes produced (student features of every crop incl. the local ones nobody reads, teacher features, both heads' logits):
    # pinned to the reference by the step goldens' `outs::*` arrays, and the checker of the HIP path's `DINO._last_outs`
    aux = {"student_logits": p_s.detach(), "teacher_logits": p_t, "feats": [f.detach() for f in feats], "teacher_feats": tfeats,
           "head_bn": hbn, "momentum_head_bn": tbn}
    return loss.detach(), grads, new_center, aux


@torch.no_grad()
def validation_step(sd: Params, crops: List[torch.Tensor], num_channels: List[List[int]], num_large_crops: int,
                    teacher_temp: float, ssl_val_loss: bool, student_temp: float = 0.1, nheads: int = 2):
    """DINO.validation_step for mixed-channel batches (dino.py:327-365 over base.py:753-870 and :1278-1375).
    ssl_val_loss: student (backbone + probe + head) on the large crops, student backbone on the small ones, teacher on the
    large crops, dino_loss_val on them and -- because the loss module updates it in forward (losses/dino.py:98) -- the moved
    centre.  Otherwise `crops` is the single validation crop: student forward only, batch_size = number of images.
    Returns a dict with feats / logits / z (lists per crop when ssl_val_loss), momentum_z, dino_loss_val, center, batch_size."""
    bb, hd = split_prefix(sd, "backbone."), split_prefix(sd, "head.")
    tbb, thd = split_prefix(sd, "momentum_backbone."), split_prefix(sd, "momentum_head.")
    W, b = sd["classifier.weight"], sd["classifier.bias"]
    if not ssl_val_loss:
        x = crops[0] if isinstance(crops, (list, tuple)) else crops
        f = backbone_ragged(bb, x, num_channels[0], nheads)
        return {"feats": f, "logits": f @ W.t() + b, "z": head_forward(hd, f), "batch_size": len(num_channels[0]),
                "center": sd["dino_loss_func.center"]}
    feats = [backbone_ragged(bb, crops[k], num_channels[k], nheads) for k in range(len(crops))]
    z = [head_forward(hd, f) for f in feats[:num_large_crops]]
    tz = [head_forward(thd, backbone_ragged(tbb, crops[k], num_channels[k], nheads)) for k in range(num_large_crops)]
    p_s, p_t = torch.cat(z), torch.cat(tz)
    center = sd["dino_loss_func.center"]
    return {"feats": feats, "logits": [f @ W.t() + b for f in feats[:num_large_crops]], "z": z, "momentum_z": tz,
            "dino_loss_val": dino_loss(p_s, p_t, center, teacher_temp, student_temp), "center": center_update(center, p_t),
            "batch_size": len(num_channels[0])}


def adamw_step(param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
               wd: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """torch.optim.AdamW single-tensor update (the optimiser the eval yamls name; base.py:67-72)."""
    param = param * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    param = param - (lr / bc1) * m / denom
    return param, m, v


# ======================================================================================
# (f)-3b  linear / fine-tune evaluation  (src/methods/linear.py:373-511; src/utils/metrics.py:26-52)
# ======================================================================================
def accuracy_at_k(outputs: torch.Tensor, targets: torch.Tensor, top_k=(1, 5)) -> List[float]:
    """Percent of rows whose target is among the k largest outputs (metrics.py:26-52)."""
    pred = outputs.topk(min(max(top_k), outputs.shape[1]), 1, True, True)[1]
    hit = pred.eq(targets.view(-1, 1))
    return [100.0 * float(hit[:, :k].any(1).sum()) / targets.numel() for k in top_k]


def linear_features(bb: Params, x: torch.Tensor, num_channels: Sequence[int], return_all_tokens: bool, mixed_channels: bool,
                    nheads: int = 2) -> torch.Tensor:
    """LinearModel.forward up to the classifier input (linear.py:383-427, multi_channels branch): CLS rows (B, D), or -- all
    tokens, fixed channel count -- the valid patch tokens viewed per channel image, grouped per image and flattened."""
    f = backbone_ragged(bb, x, num_channels, nheads, return_all_tokens=return_all_tokens)
    if not mixed_channels and return_all_tokens:
        chunks = f.view(sum(num_channels), -1, f.shape[-1])                       # :410-413
        f = torch.stack(torch.split(chunks, list(num_channels), dim=0), dim=0)     # :414-421 (equal channel counts or it fails)
        f = f.flatten(start_dim=1)                                                 # :423
    return f


def linear_step(bb: Params, W: torch.Tensor, b: torch.Tensor, x: torch.Tensor, num_channels: Sequence[int], targets: torch.Tensor,
                return_all_tokens: bool, mixed_channels: bool, finetune: bool, nheads: int = 2):
    """shared_step (linear.py:434-511) + backward: returns (loss, logits, feats, acc1, acc5, grads) where grads holds
    "classifier.weight", "classifier.bias" and, when fine-tuning, "backbone.<name>" for every backbone tensor that gets one."""
    bbp = {k: v.detach().clone().requires_grad_(finetune) for k, v in bb.items()}
    Wp, bp = W.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    with torch.set_grad_enabled(finetune):                                         # :380
        feats = linear_features(bbp, x, num_channels, return_all_tokens, mixed_channels, nheads)
    logits = feats @ Wp.t() + bp                                                   # :429-431
    loss = F.cross_entropy(logits, targets)                                        # :466
    loss.backward()
    acc1, acc5 = accuracy_at_k(logits.detach(), targets)
    grads = {"classifier.weight": Wp.grad, "classifier.bias": bp.grad}
    if finetune:
        grads.update({"backbone." + k: v.grad for k, v in bbp.items() if v.grad is not None})
    return loss.detach(), logits.detach(), feats.detach(), acc1, acc5, grads


def regression_step(bb: Params, W: torch.Tensor, b: torch.Tensor, x: torch.Tensor, num_channels: Sequence[int], targets: torch.Tensor,
                    finetune: bool, nheads: int = 2):
    """RegressionModel.shared_step + backward (src/methods/regression.py:332-436): CLS features -> one output node, nn.MSELoss against
    the targets unsqueezed to (B, 1).  Returns (loss, outputs (B, 1), grads) with "regressor.weight", "regressor.bias" and, when
    fine-tuning, "backbone.<name>"."""
    bbp = {k: v.detach().clone().requires_grad_(finetune) for k, v in bb.items()}
    Wp, bp = W.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    with torch.set_grad_enabled(finetune):
        feats = backbone_ragged(bbp, x, num_channels, nheads)
    out = feats @ Wp.t() + bp
    loss = F.mse_loss(out, targets.unsqueeze(1))      # :420-427
    loss.backward()
    grads = {"regressor.weight": Wp.grad, "regressor.bias": bp.grad}
    if finetune:
        grads.update({"backbone." + k: v.grad for k, v in bbp.items() if v.grad is not None})
    return loss.detach(), out.detach(), grads


def sgd_step(param: torch.Tensor, grad: torch.Tensor, buf: Optional[torch.Tensor], lr: float, momentum: float = 0.0,
             weight_decay: float = 0.0):
    """torch.optim.SGD single-tensor update (linear.py:51-56 "sgd"; dampening 0, no Nesterov)."""
    d_p = grad + weight_decay * param
    if momentum != 0:
        buf = d_p.clone() if buf is None else buf * momentum + d_p
        d_p = buf
    return param - lr * d_p, buf


# ======================================================================================
# (f)-1  LARS  (src/utils/lars.py:112-167)  and the bias/norm weight-decay split (src/utils/misc.py:425-454)
# ======================================================================================
def lars_step(param: torch.Tensor, grad: torch.Tensor, buf: Optional[torch.Tensor], lr: float, momentum: float = 0.9,
              dampening: float = 0.0, weight_decay: float = 0.0, nesterov: bool = False, eta: float = 1e-3, eps: float = 1e-8,
              clip_lr: bool = False, exclude_bias_n_norm: bool = False):
    """One LARS update of one tensor; returns (new_param, new_momentum_buffer)."""
    d_p = grad
    p_norm = torch.norm(param)
    g_norm = torch.norm(grad)
    if param.ndim != 1 or not exclude_bias_n_norm:           # lars.py:139
        if p_norm != 0 and g_norm != 0:                      # :140
            lars_lr = p_norm / (g_norm + p_norm * weight_decay + eps) * eta
            if clip_lr:
                lars_lr = min(lars_lr / lr, 1)               # :145-146
            d_p = (d_p + weight_decay * param) * lars_lr     # :148-149
    if momentum != 0:                                        # :152-163
        buf = d_p.clone() if buf is None else buf * momentum + (1 - dampening) * d_p
        d_p = d_p + momentum * buf if nesterov else buf
    return param - lr * d_p, buf


def split_bias_and_norm_groups(groups):
    """remove_bias_and_norm_from_weight_decay (misc.py:425-454): per group, parameters with ndim <= 1 move to a
    `<name>_no_decay` group with weight_decay 0, decay group first."""
    out = []
    for group in groups:
        decay = {k: v for k, v in group.items() if k != "params"}
        no_decay = {k: v for k, v in group.items() if k != "params"}
        no_decay["weight_decay"] = 0
        if group.get("name"):
            no_decay["name"] = group["name"] + "_no_decay"
        dp = [p for p in group["params"] if p.ndim > 1]
        ndp = [p for p in group["params"] if p.ndim <= 1]
        if dp:
            decay["params"] = dp
            out.append(decay)
        if ndp:
            no_decay["params"] = ndp
            out.append(no_decay)
    return out


# ======================================================================================
# evaluation side: weighted k-NN vote  (src/utils/knn.py:96-177)
# ======================================================================================
def knn_predict(train_features: torch.Tensor, train_targets: torch.Tensor, test_features: torch.Tensor, k: int = 20,
                T: float = 0.07, distance_fx: str = "cosine", epsilon: float = 1e-5, num_classes: Optional[int] = None):
    """Class ranking of every test sample (n_test, num_classes), best first, and the per-class vote mass.
    cosine: features L2-normalised (knn.py:116-118), similarity = dot, neighbour weight exp(sim / T) (:151-152);
    euclidean: similarity = weight = 1 / (dist + epsilon) (:138-139).  The k most similar train samples vote for their class
    with that weight (:143-160); classes are ranked by vote mass (:161)."""
    if distance_fx == "cosine":
        train_features = F.normalize(train_features)
        test_features = F.normalize(test_features)
        sims = test_features @ train_features.t()
    elif distance_fx == "euclidean":
        sims = 1 / (torch.cdist(test_features, train_features) + epsilon)
    else:
        raise NotImplementedError(distance_fx)
    k = min(k, train_targets.numel())
    if num_classes is None:
        num_classes = int(train_targets.max()) + 1
    val, idx = sims.topk(k, largest=True, sorted=True)
    w = (val / T).exp() if distance_fx == "cosine" else val
    votes = torch.zeros(test_features.shape[0], num_classes, dtype=w.dtype)
    votes.scatter_add_(1, train_targets[idx].long(), w)
    return votes.sort(1, True)[1], votes


def knn_accuracy(train_features, train_targets, test_features, test_targets, k=20, T=0.07, distance_fx="cosine", epsilon=1e-5):
    """(top1 %, top5 %) exactly as WeightedKNNClassifier.compute returns them (knn.py:163-177); num_classes is the number of
    distinct TEST targets there (:120) -- the vote table is that wide, so targets must be < that count."""
    nc = int(torch.unique(test_targets).numel())
    rank, _ = knn_predict(train_features, train_targets, test_features, k, T, distance_fx, epsilon, num_classes=nc)
    kk = min(k, train_targets.numel())
    correct = rank.eq(test_targets.view(-1, 1))
    top1 = correct[:, :1].sum().item() * 100.0 / test_targets.numel()
    top5 = correct[:, :min(5, kk, correct.shape[-1])].sum().item() * 100.0 / test_targets.numel()
    return top1, top5


def strip_backbone_prefix(state: Params) -> Params:
    """Checkpoint -> backbone state_dict exactly as the evaluation scripts rewrite it (main_linear.py:103-110):
    'encoder' -> 'backbone', then 'backbone.' removed from every key that contains 'backbone'; other keys are dropped."""
    state = dict(state)
    for k in list(state.keys()):
        if "encoder" in k:
            state[k.replace("encoder", "backbone")] = state[k]
        if "backbone" in k:
            state[k.replace("backbone.", "")] = state[k]
        del state[k]
    return state


# ======================================================================================
# data side (the reference-owned part): per-channel intensity jitter  (src/data/custom_transforms.py:301-351)
# ======================================================================================
def custom_color_jitter(img_hwc: np.ndarray, int_shifts: np.ndarray, gammas: np.ndarray) -> np.ndarray:
    """CustomColorJitter.apply with its two random draws made explicit: per channel c,
    out[..., c] = clamp(gamma_c * (img[..., c] + shift_c), 0, 1)  (:333 shift, :339-344 brightness blend against zeros with
    ratio gamma and clamp to the float bound 1.0).  float32 HWC in, float32 HWC out."""
    x = torch.from_numpy(np.ascontiguousarray(img_hwc.transpose(2, 0, 1))).float()
    for c in range(x.shape[0]):
        ch = x[c] + float(int_shifts[c])
        x[c] = (float(gammas[c]) * ch).clamp(0, 1.0)
    return x.numpy().transpose(1, 2, 0)
