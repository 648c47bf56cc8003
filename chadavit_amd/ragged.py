"""Host-side description of one ragged-packed crop batch (the packer of the north star).

The reference pads every image to 10 channels and derives a key mask from exact zeros
(chada_vit.py:226-239).  Here the layout is known on the host from `list_num_channels[index]`:
image i owns packed rows [cu_seqlens[i], cu_seqlens[i+1]) = [CLS, ch0 patches, ch1 patches, ...].
All index arrays go to the device in ONE int32 upload.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from ._lib import lib


class RaggedBatch:
    def __init__(self, num_channels: Sequence[int], patches_per_channel: int, device):
        nch = [int(c) for c in num_channels]
        if any(c < 1 for c in nch):
            raise RuntimeError("every image needs at least one channel")
        self.num_channels: List[int] = nch
        self.p = int(patches_per_channel)
        self.B = len(nch)
        self.n_chan = sum(nch)
        import numpy as np
        C = np.asarray(nch, dtype=np.int64)
        lens_a = 1 + C * self.p
        cu_a = np.concatenate([[0], np.cumsum(lens_a)])
        lens, cu = lens_a.tolist(), cu_a.tolist()
        self.lens = lens
        self.T = cu[-1]
        self.max_len = max(lens)
        tile = lib().chadavit_attn_tile_rows()
        # attention work items (image, tile).  Entry j runs on XCD j % 8 (the kernels use a 1-D grid): every image's tiles go
        # to ONE of eight sub-lists (longest image first, always to the least loaded list) so they share that XCD's L2, and
        # the sub-lists are interleaved, padded with (-1, 0), into the final list -- long work first within each XCD.
        # (Built with numpy around the one sequential piece -- which list is the least loaded -- : a variable-channel data set meets a
        # new description every batch, and ~2 500 sequences took 30-45 ms of interpreter per step as Python lists of tuples.)
        tiles = (lens_a + tile - 1) // tile
        order = np.argsort(-lens_a, kind="stable").tolist()
        fill = [0] * 8
        tgt_of, start_of = [0] * self.B, [0] * self.B
        tl = tiles.tolist()
        for b in order:
            x = fill.index(min(fill))   # the first of the least loaded lists
            tgt_of[b], start_of[b] = x, fill[x]
            fill[x] += tl[b]
        slots = max(fill)
        self.n_work = 8 * slots
        work = np.zeros((self.n_work, 2), dtype=np.int64)
        work[:, 0] = -1
        n_items = int(tiles.sum())
        img = np.repeat(np.arange(self.B, dtype=np.int64), tiles)
        t_idx = np.arange(n_items, dtype=np.int64) - np.repeat(np.cumsum(tiles) - tiles, tiles)
        at = (np.repeat(np.asarray(start_of, dtype=np.int64), tiles) + t_idx) * 8 + np.repeat(np.asarray(tgt_of, dtype=np.int64), tiles)
        work[at, 0], work[at, 1] = img, t_idx
        chan_img = np.repeat(np.arange(self.B, dtype=np.int64), C)
        chan_idx = np.arange(self.n_chan, dtype=np.int64) - np.repeat(np.cumsum(C) - C, C)
        flat = np.concatenate([cu_a, work.reshape(-1), chan_img, chan_idx, cu_a[:-1]]).astype(np.int32)
        host = torch.from_numpy(flat)
        if torch.device(device).type == "cuda":
            host = host.pin_memory()
        dev = host.to(device, non_blocking=True)
        # The upload is enqueued on whatever stream is current HERE.  Consumers on another stream (the teacher / local-crop side
        # streams of DINO.training_step share one description with the student pass) must order themselves behind it: `ready`
        # is recorded right after the copy and `use_on_current_stream` waits for it and tells the caching allocator about the
        # extra reader, so neither the copy nor a later re-issue of the buffer can race a kernel that still reads the indices.
        self._dev = dev
        self._alloc_stream = None
        self.ready = None
        if dev.is_cuda:
            self._alloc_stream = torch.cuda.current_stream(dev.device)
            self.ready = torch.cuda.Event()
            self.ready.record(self._alloc_stream)
        o = 0
        self.cu_seqlens = dev[o:o + self.B + 1]; o += self.B + 1
        self.work = dev[o:o + 2 * self.n_work].view(self.n_work, 2); o += 2 * self.n_work
        self.chan_img = dev[o:o + self.n_chan]; o += self.n_chan
        self.chan_idx = dev[o:o + self.n_chan]; o += self.n_chan
        self.cls_rows = dev[o:o + self.B]
        self._host_cu = cu

    @property
    def host_cu_seqlens(self) -> List[int]:
        return self._host_cu

    def use_on_current_stream(self) -> "RaggedBatch":
        """Make the index arrays safe to read from kernels launched on the CURRENT stream (no-op on the uploading stream)."""
        if self.ready is not None and not torch.cuda.is_current_stream_capturing():  # (a captured step: the upload completed long before)
            cur = torch.cuda.current_stream(self._dev.device)
            if cur != self._alloc_stream:
                cur.wait_event(self.ready)
                self._dev.record_stream(cur)
        return self


# The student and the teacher pass of one step (and every step of a fixed-channel dataset) describe the same batch: the index
# arrays are built and uploaded once.  Read-only on the device; a handful of entries of a few hundred KiB.
_CACHE: "dict" = {}
_CACHE_MAX = 8


# A captured hipGraph bakes in the device addresses of the index arrays it read: whoever captures must OWN the descriptions for as
# long as the graph lives (the cache above forgets an entry after eight others).  `recording_uses(lst)` appends every description
# handed out while it is active to `lst` (chadavit_amd.graphed keeps that list with the graph).
_RECORDERS: "list" = []


class recording_uses:
    def __init__(self, sink: list):
        self.sink = sink

    def __enter__(self):
        _RECORDERS.append(self.sink)
        return self.sink

    def __exit__(self, *exc):
        _RECORDERS.remove(self.sink)
        return False


def ragged_batch(num_channels: Sequence[int], patches_per_channel: int, device) -> RaggedBatch:
    key = (tuple(int(c) for c in num_channels), int(patches_per_channel), str(device))
    rb = _CACHE.pop(key, None)
    if rb is None:
        rb = RaggedBatch(key[0], patches_per_channel, device)
        while len(_CACHE) >= _CACHE_MAX:
            _CACHE.pop(next(iter(_CACHE)))
    _CACHE[key] = rb   # most recently used last
    for sink in _RECORDERS:
        if not any(r is rb for r in sink):
            sink.append(rb)
    return rb.use_on_current_stream()

