"""The whole DINO training step as ONE hipGraph (small-batch regime).

At the reference's own configuration (BASELINE.json configs[0]: Tiny/16, batch 4, 2 global crops) a training step is ~700 kernel
launches whose GPU time (~4 ms) is less than half of what the host needs to issue them (~10 ms): the step is launch-bound.
`GraphedTrainStep` captures the step of `chadavit_amd.trainer.Trainer.train_step` -- student + teacher forward, loss, backward,
`on_after_backward`, fused AdamW, EMA -- once per batch signature and replays it with one host call.

What changes from step to step and is NOT frozen into the graph:
  * the crops: copied into static input buffers before each replay (one copy per resolution);
  * the learning rate per param group, Adam's bias corrections, the EMA tau and the teacher temperature: device-resident scalars
    (`optim.DeviceHyper`, `MomentumUpdater.tau_dev`, `DINOLoss.temp_dev`) uploaded in front of every replay from a small ring of
    pinned staging buffers (a staging buffer is rewritten only after the upload that read it has completed, so a host that runs
    several replays ahead of the GPU cannot overwrite values a queued step has yet to read); the LR schedule, the tau schedule (momentum.py:76-87) and the step counters keep running on the host exactly as in the
    eager loop (base.py:1250-1276 order: optimiser step, then EMA with the CURRENT tau, then the tau update);
  * which parameters receive a gradient (the prototypes are frozen while `epoch < freeze_last_layer`, dino.py:367-376) and the
    channel mix / crop sizes of the batch: part of the signature -- a new signature captures a new graph (kept, keyed; going back
    to an earlier signature replays its graph).  Meant for fixed-channel data: with a new channel mix every step every step
    would capture.

Same kernels, same results as the eager step (tests/test_model_gpu.py::test_graphed_train_step_matches_eager); by default the capture
keeps DINO's side streams on, so the graph has parallel branches (teacher || student forward, local-crop pass || backward, weight-gradient
GEMMs || the rest of the backward) -- independent launch-latency-bound kernels overlap.
Single process only: the gradient collectives of the data-parallel path are issued from Python hooks and stay eager.
"""
from __future__ import annotations

from typing import Any, Dict, List, Sequence, Tuple

import torch

from .optim import DeviceHyper, FusedAdamW


class GraphedTrainStep:
    def __init__(self, trainer, warmup: int = 2, parallel_streams: bool = True):
        """parallel_streams: capture the step with DINO's side streams ON (teacher forward || student forward, local-crop pass || loss +
        backward, weight-gradient GEMMs || the rest of the backward): the graph then has parallel branches, and in the launch-latency-bound
        regime this class exists for, independent tiny kernels of different branches overlap (cfg1: 861 -> 1 005 images/s, cfg2 at 16
        images: 1 759 -> 1 962; results bit-identical to the single-stream capture and to the eager step).  False: one stream."""
        m = trainer.model
        if trainer.grad_sync is not None and trainer.grad_sync.reducer.active:
            raise RuntimeError("GraphedTrainStep is single-process (the data-parallel hooks issue collectives from Python)")
        if not isinstance(trainer.optimizer, FusedAdamW):
            raise RuntimeError("GraphedTrainStep needs the fused AdamW (device-resident hyper-parameters)")
        if getattr(m, "knn_eval", False):
            raise RuntimeError("GraphedTrainStep: the online k-NN bank grows every step (host-side state); disable knn_eval")
        self.trainer, self.model, self.warmup = trainer, m, max(1, warmup)
        self.device = next(m.parameters()).device
        self._eager_streams = (m.overlap_streams, m.backbone.dw_side_stream)   # the eager loop's own setting, put back by close()
        self._parallel = bool(parallel_streams)
        m.overlap_streams = m.backbone.dw_side_stream = self._parallel   # False: one stream, the capture stream
        self.hyper = DeviceHyper(self.device)
        self.extra_host = torch.zeros(2, dtype=torch.float32)                # [tau, teacher temperature]
        self.extra_dev = torch.zeros(2, dtype=torch.float32, device=self.device)
        # pinned staging ring for the per-step scalars: [hyper slots | tau, temperature] per entry, one event per entry
        nh = self.hyper.host.numel()
        self._ring = [(torch.zeros(nh + 2, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(4)]
        self._ring_used = [False] * 4
        self._ring_next = 0
        self.graphs: Dict[Tuple, Dict[str, Any]] = {}
        self._current = None   # signature of the graph whose slot table / active parameters the DeviceHyper holds
        self._entered = False

    # ------------------------------------------------------------------------------------------
    def _enter(self):
        if self._entered:
            return
        self.model.overlap_streams = self.model.backbone.dw_side_stream = self._parallel   # (close() puts the eager setting back)
        self.trainer.optimizer.device_hyper = self.hyper
        self.model.momentum_updater.tau_dev = self.extra_dev[0:1]
        self.model.dino_loss_func.temp_dev = self.extra_dev[1:2]
        self._entered = True

    def close(self):
        """Back to the eager loop: step counters written back, device scalars detached."""
        if self._entered:
            self.hyper.switch(self.trainer.optimizer, [], [])
            self._current = None
            self.trainer.optimizer.device_hyper = None
            self.model.momentum_updater.tau_dev = None
            self.model.dino_loss_func.temp_dev = None
            self._entered = False
        self.model.overlap_streams, self.model.backbone.dw_side_stream = self._eager_streams

    def _fill_scalars(self):
        m = self.model
        self.hyper.fill(self.trainer.optimizer)
        self.extra_host[0] = float(m.momentum_updater.cur_tau)
        self.extra_host[1] = float(m.dino_loss_func.teacher_temp_schedule[m.dino_loss_func.epoch])

    def _upload_scalars(self):
        """This step's scalars to the device, on the current stream, in front of the step's launches (never inside the graph: a
        captured copy would read ONE host address whenever the replay gets to it)."""
        k = self._ring_next
        self._ring_next = (k + 1) % len(self._ring)
        stage, ev = self._ring[k]
        if self._ring_used[k]:
            ev.synchronize()           # the upload that last read this entry is done (only ever waits when 4 steps ahead)
        nh = self.hyper.host.numel()
        stage[:nh].copy_(self.hyper.host)
        stage[nh:].copy_(self.extra_host)
        self.hyper.dev.copy_(stage[:nh], non_blocking=True)
        self.extra_dev.copy_(stage[nh:], non_blocking=True)
        ev.record()
        self._ring_used[k] = True

    def _snapshot(self):
        """Everything a training step changes, so that the warm-up steps capture needs leave no trace."""
        tr, m = self.trainer, self.model
        tr.optimizer._index()
        flats = [mod.flat_params() for mod in (m.backbone, m.head, m.momentum_backbone, m.momentum_head)]
        dev = [(f, f.flat.clone()) for f in flats]
        dev += [(None, (sl[k], sl[k].clone())) for sl in tr.optimizer._slabs.values() for k in ("m", "v")]
        # every module buffer: the centre, and with use_bn_in_head the BatchNorm1d running estimates / batch counters of both heads
        # (ops.bn_stats updates them in place once per crop)
        dev += [(None, (b, b.clone())) for b in m.buffers()]
        return {"dev": dev, "center": m.dino_loss_func.center.clone(), "global_step": tr.global_step, "last_step": m.last_step,
                "tau": m.momentum_updater.cur_tau, "lrs": [g["lr"] for g in tr.optimizer.param_groups],
                "sched": None if tr.scheduler is None else dict(tr.scheduler.state_dict()), "taken": self.hyper.taken}

    def _restore(self, snap):
        tr, m = self.trainer, self.model
        for f, saved in snap["dev"]:
            if f is None:
                saved[0].copy_(saved[1])
            else:
                f.flat.copy_(saved)
                f.mark_dirty()
        m.dino_loss_func.center.copy_(snap["center"])
        tr.global_step, m.last_step, m.momentum_updater.cur_tau, self.hyper.taken = snap["global_step"], snap["last_step"], snap["tau"], snap["taken"]
        for g, lr in zip(tr.optimizer.param_groups, snap["lrs"]):
            g["lr"] = lr
        if tr.scheduler is not None:
            tr.scheduler.load_state_dict(snap["sched"])

    def _body(self, batch, batch_idx) -> torch.Tensor:
        """Trainer.train_step's device work, in its order (the host-side schedule updates follow in __call__)."""
        tr, m = self.trainer, self.model
        loss = m.training_step(batch, batch_idx)
        loss.backward()
        m.on_after_backward()
        tr.optimizer.step()
        m.optimizer_zero_grad(tr.current_epoch, batch_idx, tr.optimizer)
        for mp in m.momentum_pairs:     # on_train_batch_end's EMA (the tau update is host arithmetic: done in __call__)
            m.momentum_updater.update(*mp)
        return loss

    def _after_step(self, batch_idx):
        tr, m = self.trainer, self.model
        self.hyper.advance()
        tr.global_step += 1
        if tr.scheduler is not None:
            tr.scheduler.step()
        m.log("tau", m.momentum_updater.cur_tau)
        m.momentum_updater.update_tau(cur_step=tr.global_step, max_steps=tr.estimated_stepping_batches)
        m.last_step = tr.global_step

    @staticmethod
    def _signature(batch, epoch_frozen: bool) -> Tuple:
        X, _, ncl = batch
        X = [X] if isinstance(X, torch.Tensor) else list(X)
        ncl = [ncl] if isinstance(ncl[0], int) else ncl
        return (tuple(tuple(x.shape) for x in X), tuple(tuple(int(c) for c in n) for n in ncl), epoch_frozen)

    def _static_batch(self, batch):
        """Static input buffers: crops of one resolution back to back in one buffer (training_step then takes them as a view)."""
        X, labels, ncl = batch
        single = isinstance(X, torch.Tensor)
        X = [X] if single else list(X)
        bufs, views = [], []
        i = 0
        while i < len(X):
            j = i
            while j < len(X) and X[j].shape == X[i].shape:
                j += 1
            buf = torch.empty((sum(x.shape[0] for x in X[i:j]),) + tuple(X[i].shape[1:]), device=self.device, dtype=X[i].dtype)
            bufs.append((buf, i, j))
            views += list(buf.split([x.shape[0] for x in X[i:j]]))
            i = j
        lab = torch.empty_like(labels)
        return {"bufs": bufs, "views": views, "labels": lab, "batch": (views[0] if single else views, lab, ncl)}

    @staticmethod
    def _copy_in(st, batch):
        X, labels, _ = batch
        X = [X] if isinstance(X, torch.Tensor) else list(X)
        for v, x in zip(st["views"], X):
            v.copy_(x, non_blocking=True)
        st["labels"].copy_(labels, non_blocking=True)

    # ------------------------------------------------------------------------------------------
    def __call__(self, batch, batch_idx: int = 0) -> torch.Tensor:
        tr, m = self.trainer, self.model
        if not isinstance(getattr(type(m), "current_epoch", None), property):   # (read-only on a real LightningModule: trainer.py)
            m.current_epoch = tr.current_epoch
        if batch_idx == 0:
            m.on_train_epoch_start()
        frozen = tr.current_epoch < m.freeze_last_layer
        key = self._signature(batch, frozen)
        g = self.graphs.get(key)
        if g is None:
            # another set of active parameters / another batch shape: the step counters go back to the optimizer first and the
            # capture below builds this graph's own slot table
            self.hyper.switch(tr.optimizer, [], [])
            self._current = key
            self._enter()
            g = self._static_batch(batch)
            # warm-up steps run eagerly on a side stream (graph capture requires it; they fill every host-side cache: ragged
            # descriptions, workspaces, index tables), then the state they changed is put back and ONE step is captured.  A captured
            # step is not executed: the replay below is this call's training step.
            snap = self._snapshot()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self._copy_in(g, batch)
                    self._fill_scalars()
                    self._upload_scalars()
                    self._body(g["batch"], batch_idx)
                    self._after_step(batch_idx)
            torch.cuda.current_stream(self.device).wait_stream(side)
            self._restore(snap)
            torch.cuda.synchronize(self.device)
            for mod in (m.backbone, m.head, m.momentum_backbone, m.momentum_head):
                mod.flat_params().mark_dirty()   # the bf16 casts / weight packings of every network must be IN the graph
            g["graph"] = torch.cuda.CUDAGraph()
            # the graph bakes in the addresses of the ragged index arrays (cu_seqlens, work lists, ...): it owns their descriptions
            # from here on -- the 8-entry cache of chadavit_amd.ragged may forget them (a validation epoch on other channel mixes)
            from .ragged import recording_uses
            g["ragged"] = []
            with recording_uses(g["ragged"]), torch.cuda.graph(g["graph"]):
                g["loss"] = self._body(g["batch"], batch_idx)
            self._restore(snap)   # (host-side counters the captured body touched: none today; cheap and safe)
            g["hyper_slots"], g["hyper_active"] = list(self.hyper.slots), list(self.hyper.active)
            self.graphs[key] = g
        elif key != self._current:   # back to a graph captured earlier
            self.hyper.switch(tr.optimizer, g["hyper_slots"], g["hyper_active"])
            self._current = key
        self._copy_in(g, batch)
        self._fill_scalars()
        self._upload_scalars()
        g["graph"].replay()
        # the replay moved the parameters: eager / validation / k-NN passes between replays must re-cast their bf16, packed and MX-fp8
        # shadows (host-side version bump; the graph itself casts inside)
        for mod in (m.backbone, m.head, m.momentum_backbone, m.momentum_head):
            mod.flat_params().mark_dirty()
        self._after_step(batch_idx)
        return g["loss"].clone()   # (the graph's loss tensor is rewritten by every replay)
