"""Lightning-free training loop that calls the DINO hooks in Lightning's order (a hook the module does not
define -- LinearModel has no EMA / gradient hooks -- is skipped, as Lightning's no-op defaults would be):

  on_train_start; per epoch: on_train_epoch_start; per batch: training_step -> loss.backward() ->
  [gradient all-reduce finishes] -> on_after_backward -> optimizer.step() -> optimizer_zero_grad ->
  lr_scheduler.step() -> on_train_batch_end (EMA, tau).

pytorch_lightning is not installed on the GPU box; `main_pretrain.py`-style drivers that do have it can
use chadavit_amd.methods.dino.DINO directly as a LightningModule instead.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch


class Trainer:
    def __init__(self, max_epochs: int, steps_per_epoch: int, grad_sync=None, graph: bool = False):
        """graph = True: `train_step` replays the whole step as one hipGraph per batch signature (chadavit_amd.graphed.GraphedTrainStep:
        the launch-bound small-batch regime, fixed-channel data, single process, fused AdamW) -- same results as the eager step."""
        self.graph = bool(graph)
        self._graphed = None
        self.max_epochs = max_epochs
        self.steps_per_epoch = steps_per_epoch
        self.estimated_stepping_batches = max_epochs * steps_per_epoch
        self.global_step = 0
        self.current_epoch = 0
        self.grad_sync = grad_sync  # chadavit_amd.parallel.GradSync or None
        self.model = None
        self.optimizer = None
        self.scheduler = None

    def attach(self, model):
        self.model = model
        model.trainer = self
        conf = model.configure_optimizers()
        if isinstance(conf, tuple) or isinstance(conf, list):
            self.optimizer = conf[0][0]
            sched = conf[1][0]
            self.scheduler = sched["scheduler"] if isinstance(sched, dict) else sched
        else:
            self.optimizer = conf
        if self.grad_sync is not None:
            self.grad_sync.attach(model)
        self._hook("on_train_start")
        return self

    def _hook(self, name, *args):
        fn = getattr(self.model, name, None)
        return fn(*args) if callable(fn) else None

    def train_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        if self.graph:
            if self._graphed is None:
                from .graphed import GraphedTrainStep
                self._graphed = GraphedTrainStep(self)
            return self._graphed(batch, batch_idx)
        return self.eager_step(batch, batch_idx)

    def close_graph(self):
        """Back to eager steps (step counters written back to the optimiser, the eager loop's stream setting restored)."""
        if self._graphed is not None:
            self._graphed.close()
        self.graph, self._graphed = False, None

    def eager_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        m = self.model
        # (a real LightningModule reads `current_epoch` from its trainer -- a read-only property; the stand-in base holds it as an attribute)
        if not isinstance(getattr(type(m), "current_epoch", None), property):
            m.current_epoch = self.current_epoch
        if batch_idx == 0:
            self._hook("on_train_epoch_start")
        loss = m.training_step(batch, batch_idx)
        if self.grad_sync is not None:
            self.grad_sync.begin_backward()
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self._hook("on_after_backward")
        self.optimizer.step()
        self.global_step += 1
        if callable(getattr(m, "optimizer_zero_grad", None)):
            m.optimizer_zero_grad(self.current_epoch, batch_idx, self.optimizer)
        else:
            self.optimizer.zero_grad(set_to_none=True)
        if self.scheduler is not None:
            self.scheduler.step()
        self._hook("on_train_batch_end", None, batch, batch_idx)
        return loss

    def fit(self, model, batches_per_epoch: Iterable):
        self.attach(model)
        for ep in range(self.max_epochs):
            self.current_epoch = ep
            for i, batch in enumerate(batches_per_epoch):
                self.train_step(batch, i)
