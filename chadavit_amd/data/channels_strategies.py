"""Collate for variable-channel images (reference: src/data/channels_strategies.py:31-85).

Same output contract -- crops[k]: (sum C_i, 1, H_k, W_k) image-major then channel, labels (B,),
num_channels[k]: list[int] -- built with one torch.cat per crop instead of a Python loop per channel."""
from __future__ import annotations

import os

import torch


def one_channel_collate_fn(batch):
    first = batch[0][-2:][0]
    num_crops = len(first) if isinstance(first, list) else 1
    crop_lists = [[] for _ in range(num_crops)]
    num_channels_lists = [[] for _ in range(num_crops)]
    labels = []
    for item in batch:
        image_list, label = item[-2:]
        if isinstance(image_list, torch.Tensor):
            image_list = [image_list]
        for k, crop in enumerate(image_list):
            num_channels_lists[k].append(crop.shape[0])
            crop_lists[k].append(crop)
        labels.append(label)
    crops = _cat_grouped(crop_lists)
    crops = crops[0] if num_crops == 1 else crops
    return crops, torch.tensor(labels), num_channels_lists


def _cat_grouped(crop_lists):
    """crops[k] = cat(crop_lists[k]).unsqueeze(1), with consecutive crops of the same resolution written back to back into ONE
    buffer: the tensors handed out are the same as the reference's, but DINO.training_step can then take "all global crops" /
    "all local crops" as a single (sum C, 1, S, S) view of that buffer (`adjacent_view`) instead of copying them together again."""
    out = [None] * len(crop_lists)
    k = 0
    while k < len(crop_lists):
        j = k
        shp = crop_lists[k][0].shape[1:]
        while j + 1 < len(crop_lists) and crop_lists[j + 1][0].shape[1:] == shp and crop_lists[j + 1][0].dtype == crop_lists[k][0].dtype:
            j += 1
        rows = [sum(c.shape[0] for c in crop_lists[i]) for i in range(k, j + 1)]
        buf = torch.empty((sum(rows),) + tuple(shp), dtype=crop_lists[k][0].dtype, device=crop_lists[k][0].device)
        r0 = 0
        for i, n in zip(range(k, j + 1), rows):
            torch.cat(crop_lists[i], dim=0, out=buf[r0:r0 + n])
            out[i] = buf[r0:r0 + n].unsqueeze(1)
            r0 += n
        k = j + 1
    return out


def adjacent_view(tensors):
    """cat(tensors, 0) WITHOUT the copy when the tensors already lie back to back in one storage (what `_cat_grouped`, the device
    augmentation pipeline and bench.py's synthetic batches produce); None otherwise."""
    if os.environ.get("CHADAVIT_NO_ADJACENT_VIEW"):   # A/B aid
        return None
    t0 = tensors[0]
    off = t0.storage_offset()
    base = t0.untyped_storage().data_ptr()
    for t in tensors:
        if (not t.is_contiguous() or t.dtype != t0.dtype or t.device != t0.device or t.shape[1:] != t0.shape[1:]
                or t.untyped_storage().data_ptr() != base or t.storage_offset() != off):
            return None
        off += t.numel()
    return t0.as_strided((sum(t.shape[0] for t in tensors),) + tuple(t0.shape[1:]), t0.stride(), t0.storage_offset())
