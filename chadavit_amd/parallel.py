"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce of the flat gradient slab in
per-block spans on a side stream, overlapped with the rest of the backward.

reference: Lightning DDPStrategy -> torch DDP bucketed all-reduce(mean) (main_pretrain.py:301-303;
SURVEY.md 2.3 C1).  Here the gradient of every parameter already lives in one contiguous fp32 slab
(chadavit_amd.flat), so a "bucket" is just an [begin, end) span of it: the backward fires
`grad_ready_hook(flat, begin, end)` when a transformer block's span is final (block 11 first), the span
is averaged in place by `dist.all_reduce(..., AVG)` on the communication stream while the compute stream
keeps running block i-1's backward.  xGMI is point-to-point: 14 spans of ~3.6 MB (Tiny) per step keep
every link busy without waiting for the whole 70 MB slab.
Parameters that get no gradient (online classifier, frozen prototypes) are simply never exchanged --
stock DDP needs find_unused_parameters for that.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def force_collectives() -> bool:
    """CHADAVIT_FORCE_COLLECTIVES=1: issue every collective of the data-parallel path even in a process group of ONE rank (they are
    identities there) -- so that the RCCL code path (ReduceOp.AVG, the communication stream's hand-over, record_stream, the event
    timing) can be executed on a single-GPU box before the first multi-GPU run (tests/test_ddp_gpu.py)."""
    return bool(os.environ.get("CHADAVIT_FORCE_COLLECTIVES")) and dist.is_available() and dist.is_initialized()


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the process group when world > 1.
    backend "nccl" is RCCL on ROCm."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CHADAVIT_SINGLE_DEVICE"):  # testing aid: several ranks share GPU 0 (needs the gloo backend)
        local = 0
    if backend is None:
        backend = os.environ.get("CHADAVIT_DIST_BACKEND")
    if (world > 1 or os.environ.get("CHADAVIT_FORCE_COLLECTIVES")) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # a finite rendezvous / collective timeout: a rank that died must not hold the others for torch's default half hour
        # (torchrun tears the group down on the first failed worker anyway; this covers launchers that do not)
        # (only when CHADAVIT_DIST_TIMEOUT_S is set -- bench.py sets 600 for its own launches; a training run keeps torch's default, so a
        # rank doing long solo work -- checkpointing, k-NN evaluation -- while the others wait in a barrier is not cut off)
        kw = {}
        if os.environ.get("CHADAVIT_DIST_TIMEOUT_S"):
            from datetime import timedelta
            kw["timeout"] = timedelta(seconds=int(os.environ["CHADAVIT_DIST_TIMEOUT_S"]))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


class SpanAllReduce:
    """Average [begin, end) spans of a flat tensor across ranks, asynchronously.  Device agnostic:
    on CUDA/HIP tensors the collective runs on a side stream; on CPU (gloo) it uses async work handles."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or force_collectives()
        # RCCL ("nccl") averages in the collective; gloo has no AVG -> SUM then scale
        self.native_avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self._stream = None
        self._works: List = []
        self.spans: List[Tuple[int, int]] = []
        self.bytes = 0
        # timing=True (bench.py): per step, HIP events on both streams around the hand-over in finish() -- how long the compute
        # stream had to WAIT for the last collective (= communication not hidden behind the backward) and how long the
        # communication stream was busy from its first collective on
        self.timing = False
        self._first_ev = None
        self._timed: List = []
        self._host_wait_s: List[float] = []

    def _comm_stream(self, device):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def submit(self, flat: torch.Tensor, begin: int, end: int):
        if not self.active or end <= begin:
            return
        chunk = flat[begin:end]
        self.spans.append((begin, end))
        self.bytes += chunk.numel() * chunk.element_size()
        if chunk.device.type == "cuda":
            cs = self._comm_stream(chunk.device)
            cs.wait_stream(torch.cuda.current_stream(chunk.device))  # span is final on the compute stream
            with torch.cuda.stream(cs):
                if self.timing and self._first_ev is None:
                    self._first_ev = torch.cuda.Event(enable_timing=True)
                    self._first_ev.record(cs)
                if self.native_avg:
                    dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
                    chunk.mul_(1.0 / self.world)
        else:
            w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((w, chunk))

    def finish(self, device=None):
        """Make the averaged gradients visible to the compute stream / host."""
        if self._stream is not None:
            main = torch.cuda.current_stream(device)
            if self.timing and self._first_ev is not None:
                ready, done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ready.record(main)          # the backward's own work ends here ...
                done.record(self._stream)   # ... the last collective here
                self._timed.append((self._first_ev, ready, done))
                self._first_ev = None
            main.wait_stream(self._stream)
        if self._works:
            import time
            t0 = time.perf_counter()
            for w, chunk in self._works:
                w.wait()
                chunk.div_(self.world)
            if self.timing:
                self._host_wait_s.append(time.perf_counter() - t0)
        self._works = []

    def timing_summary(self) -> Optional[dict]:
        """After a device synchronize: {"exposed_ms_per_step", "comm_busy_ms_per_step", "steps"} over the timed steps."""
        if self._timed:
            exp = [max(0.0, ready.elapsed_time(done)) for _, ready, done in self._timed]
            busy = [first.elapsed_time(done) for first, _, done in self._timed]
            n = len(exp)
            self._timed = []
            return {"exposed_ms_per_step": round(sum(exp) / n, 4), "exposed_ms_max": round(max(exp), 4),
                    "comm_busy_ms_per_step": round(sum(busy) / n, 4), "steps": n}
        if self._host_wait_s:
            n = len(self._host_wait_s)
            out = {"exposed_ms_per_step": round(1e3 * sum(self._host_wait_s) / n, 4), "exposed_ms_max": round(1e3 * max(self._host_wait_s), 4),
                   "comm_busy_ms_per_step": None, "steps": n}
            self._host_wait_s = []
            return out
        return None

    def reset_stats(self):
        self.spans, self.bytes = [], 0


class GradSync:
    """Hooks a method module so its gradients are averaged over the ranks during / right after backward.
    DINO: backbone + head spans from their `grad_ready_hook`s.  LinearModel (no head, no teacher): the backbone's spans when
    fine-tuning, and the classifier's two small gradient tensors in `finish()`."""

    def __init__(self, group=None):
        self.reducer = SpanAllReduce(group)
        self.model = None
        self._plain: List[torch.nn.Parameter] = []   # parameters outside the flat slabs whose gradients are exchanged in finish()

    def attach(self, model):
        self.model = model
        if not self.reducer.active:
            return self
        if hasattr(model, "head"):
            if not getattr(model, "batch_crops", False):
                raise RuntimeError("GradSync needs batch_crops=True (one backward per step, so every span is final when it fires)")
            for mod in (model.backbone, model.head):
                mod.grad_ready_hook = self._hook
        else:
            if any(p.requires_grad for p in model.backbone.parameters()):
                model.backbone.grad_ready_hook = self._hook
            # (RegressionModel deletes `classifier`: a getattr default would evaluate it eagerly and raise)
            lin = model.out_layer if hasattr(model, "out_layer") else model.classifier
            self._plain = [p for p in lin.parameters() if p.requires_grad]
        self.broadcast_parameters()
        return self

    def _hook(self, flat, begin, end):
        self.reducer.submit(flat.grad, begin, end)

    def begin_backward(self):
        self.reducer.reset_stats()

    def finish(self):
        # A backbone pass parks the span that holds pos_embed while part of that gradient still travels through autograd
        # (ChAdaViT._expect_pos_accumulation); pos_embed's post-accumulate hook releases it.  If that hook never fired in this backward
        # (torch.autograd.grad / backward(inputs=...) without pos_embed, an exception on the way) the span would stay parked: its gradients
        # would never be averaged this step and the stale hand-over would fire inside a LATER step, changing the collective order between
        # ranks.  By now the backward has returned, so whatever autograd was going to add to pos_embed.grad has been added: hand it over here.
        bb = getattr(self.model, "backbone", None)
        if bb is not None and getattr(bb, "_pos_span_deferred", None) is not None:
            deferred, bb._pos_span_deferred = bb._pos_span_deferred, None
            bb._pos_autograd_pending = False
            if self.reducer.active:
                deferred()
        for p in self._plain:
            if p.grad is not None:
                g = p.grad.view(-1)
                self.reducer.submit(g, 0, g.numel())
        self.reducer.finish()

    @torch.no_grad()
    def broadcast_parameters(self, src: int = 0):
        """Initial parameter / buffer sync (DDP broadcasts module state at construction)."""
        m = self.model
        for name in ("backbone", "momentum_backbone", "head", "momentum_head"):
            mod = getattr(m, name, None)
            if mod is not None:
                f = mod.flat_params()
                dist.broadcast(f.flat, src=src)
                f.mark_dirty()
        if hasattr(m, "dino_loss_func"):
            dist.broadcast(m.dino_loss_func.center, src=src)
        for p in (m.out_layer if hasattr(m, "out_layer") else m.classifier).parameters():
            dist.broadcast(p.data, src=src)
