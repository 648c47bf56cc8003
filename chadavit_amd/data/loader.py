"""Prefetching front end of the device data path: reader threads -> pinned staging -> H2D copy + augmentation
kernels on a SIDE HIP stream, `depth` batches ahead of the training step.

The reference feeds its step from `DataLoader(dataset, collate_fn=one_channel_collate_fn, num_workers=...)`
(pretrain_dataloader.py:517-525): worker processes decode the channel files and run the albumentations chain on the CPU.
Here the workers only DECODE (`dataset.read_planes`, PIL releases the GIL); everything after that -- the copy of the raw planes and
the crop / jitter / blur kernels of `DeviceMultiCropPipeline` -- runs on the GPU, on a stream of its own, `depth` batches ahead, while
the previous step computes.  The consumer orders itself behind a batch with one event wait; nothing blocks the host.
`kernels_on="consumer"` moves the augmentation KERNELS (not the copies) to the head of the consumer's stream instead.  Not the default:
with the round-4 kernels (3.2 ms per 1 024-image 10-crop batch) the cfg2 step fed either way runs at 0.98 of the same steps on a resident
batch (producer 0.979, consumer 0.976; profiles/r04f_*); with the slower first kernels the side stream hid more of them.

    ds = IDRCell100K(root_dir=..., train=True)
    sampler = TokenBalancedBatchSampler(ds.num_channels(), global_batch, rank, world)
    loader = DevicePrefetcher(ds, sampler, DeviceMultiCropPipeline(specs, device, seed=rank), workers=16)
    for step, batch in enumerate(loader):        # batch = (crops, labels, list_num_channels), as one_channel_collate_fn returns
        trainer.train_step(batch, step)
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, Iterator, List, Optional, Sequence

import torch


_ALLOCATOR_TUNED = False


def tune_allocator_for_ragged_batches(divisions: int = 8) -> bool:
    """Variable-channel data gives every batch its own tensor sizes (token rows, channel images): torch's caching allocator then keeps a
    cached block for every size it has met and asks the driver for a new one whenever none fits -- reserved memory grew 147 -> 206 GiB over 150
    steps of the cfg2-mixed workload at 256 images (hipMalloc stalls: 2 040 images/s) and expandable segments are not supported on this
    platform.  `roundup_power2_divisions` makes the allocator round request sizes to 1/`divisions` of a power of two, so blocks are reused
    across batches: 118.6 GiB reached after 50 steps and flat from there, 2 290 images/s (scratch/r4/fed_soak.py --mixed).

    The setting is PROCESS-WIDE (every allocation is rounded up, by up to 1/`divisions`) and torch re-parses the whole configuration
    string, so this call (a) is explicit -- `DevicePrefetcher(tune_allocator=True)` or a direct call, never a side effect of building a
    loader --, (b) APPENDS the option to whatever PYTORCH_HIP_ALLOC_CONF / PYTORCH_CUDA_ALLOC_CONF / PYTORCH_ALLOC_CONF the user set
    instead of replacing it, (c) leaves a configuration alone that already names a roundup option, and (d) says once what it did.
    Returns whether the setting was applied."""
    global _ALLOCATOR_TUNED
    if _ALLOCATOR_TUNED or not torch.cuda.is_available():
        return False
    import logging
    import os
    log = logging.getLogger("chadavit_amd.data")
    user = next((os.environ[k] for k in ("PYTORCH_ALLOC_CONF", "PYTORCH_HIP_ALLOC_CONF", "PYTORCH_CUDA_ALLOC_CONF") if os.environ.get(k)), "")
    if "roundup_power2_divisions" in user:
        log.info("allocator: the user's configuration already sets a roundup option (%s): left alone", user)
        _ALLOCATOR_TUNED = True
        return False
    conf = ",".join(x for x in (user.strip().strip(","), f"roundup_power2_divisions:{int(divisions)}") if x)
    try:
        setter = getattr(torch._C, "_accelerator_setAllocatorSettings", None) or torch.cuda.memory._set_allocator_settings
        setter(conf)
    except Exception as e:  # noqa: BLE001 -- an allocator back end without the option: nothing to tune
        log.warning("allocator: could not apply %r (%s)", conf, e)
        return False
    _ALLOCATOR_TUNED = True
    log.warning("allocator: caching-allocator configuration set to %r for this process (requests rounded to 1/%d of a power of two: "
                "blocks are reused across variable-channel batches)", conf, int(divisions))
    return True


_WARNED_ALLOCATOR: list = []   # (one warning per process)


class DevicePrefetcher:
    def __init__(self, dataset, batch_sampler: Iterable[Sequence[int]], pipeline, depth: int = 2, workers: int = 8,
                 labels: Optional[Sequence[int]] = None, kernels_on: str = "producer", stream: Optional["torch.cuda.Stream"] = None,
                 raw_planes: bool = False, tune_allocator: bool = False):
        if kernels_on not in ("consumer", "producer"):
            raise ValueError("kernels_on: 'consumer' or 'producer'")
        self.defer = kernels_on == "consumer" and pipeline.device.type == "cuda"
        self.dataset, self.batch_sampler, self.pipeline = dataset, batch_sampler, pipeline
        self.depth, self.workers, self.labels = max(1, depth), max(1, workers), labels
        self.device = pipeline.device
        # stream: the side stream to produce on (default: a new one at the default priority)
        self.stream = (stream if stream is not None else torch.cuda.Stream(device=self.device)) if self.device.type == "cuda" else None
        # raw_planes: ask the dataset for the planes in their stored integer type (`read_planes(i, raw=True)`): the pipeline uploads 8 / 16-bit
        # planes as they are and converts on the GPU -- same crops, a quarter / half of the staging copy and the PCIe traffic
        self.read = (lambda i: dataset.read_planes(i, raw=True)) if raw_planes else dataset.read_planes
        # tune_allocator: opt in to `tune_allocator_for_ragged_batches()` (process-wide; see there) -- worth it for variable-channel datasets
        if tune_allocator and self.device.type == "cuda":
            tune_allocator_for_ragged_batches()
        elif self.device.type == "cuda" and callable(getattr(dataset, "num_channels", None)):
            try:
                mixed = len(set(dataset.num_channels())) > 1
            except Exception:   # (a dataset whose channel counts are not known up front)
                mixed = False
            if mixed and not _WARNED_ALLOCATOR:
                # variable-channel batches have a new set of buffer sizes every step: without the roundup option the caching allocator's reserved
                # memory grew 147 -> 206 GiB in 150 steps and throughput fell ~10 % (profiles/r04t_*); the option is process-wide, hence opt-in
                _WARNED_ALLOCATOR.append(True)
                import logging
                logging.getLogger("chadavit_amd").warning(
                    "DevicePrefetcher: the dataset mixes channel counts and tune_allocator=False -- pass tune_allocator=True (or call "
                    "chadavit_amd.data.loader.tune_allocator_for_ragged_batches() before the first GPU allocation) to keep the caching "
                    "allocator's reserved memory flat with variable-size batches")
        self.read_s = 0.0      # host seconds spent decoding (sum over batches; the reader threads' wall time per batch)
        self.batches = 0

    def __len__(self):
        return len(self.batch_sampler)

    def _produce(self, q: "queue.Queue", stop: threading.Event):
        import time
        try:
            if self.stream is not None:
                torch.cuda.set_device(self.device)
            with ThreadPoolExecutor(max_workers=self.workers) as pool:
                for idx in self.batch_sampler:
                    if stop.is_set():
                        break
                    t0 = time.perf_counter()
                    planes = list(pool.map(self.read, idx))
                    self.read_s += time.perf_counter() - t0
                    labs = [self.labels[i] for i in idx] if self.labels is not None else None
                    launch = None
                    if self.stream is not None:
                        with torch.cuda.stream(self.stream):
                            batch = self.pipeline(planes, labels=labs, defer=self.defer)
                            if self.defer:
                                batch, launch = batch[:-1], batch[-1]
                            ev = torch.cuda.Event()
                            ev.record(self.stream)
                    else:
                        batch, ev = self.pipeline(planes, labels=labs), None
                    self.batches += 1
                    q.put((batch, ev, launch))
            q.put(None)
        except BaseException as e:  # noqa: BLE001 - handed to the consumer
            q.put(e)

    def __iter__(self) -> Iterator:
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        th = threading.Thread(target=self._produce, args=(q, stop), daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                batch, ev, launch = item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    crops = batch[0] if isinstance(batch[0], (list, tuple)) else [batch[0]]
                    for c in crops:   # allocated on the side stream, consumed on this one
                        c.record_stream(cur)
                    batch[1].record_stream(cur)
                    if launch is not None:
                        launch(cur)   # the augmentation kernels, in front of the step that reads their output
                yield batch
        finally:
            stop.set()
            while th.is_alive():   # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)


class InMemoryPlanes:
    """A dataset of raw planes held in host memory (synthetic or pre-decoded): the reader side of the path without the disk."""

    def __init__(self, planes: List):
        self.planes = planes

    def __len__(self):
        return len(self.planes)

    def num_channels(self) -> List[int]:
        return [int(p.shape[0]) for p in self.planes]

    def read_planes(self, index: int, raw: bool = False):
        return self.planes[index]   # (held as decoded: whatever type they have)
