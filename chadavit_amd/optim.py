"""Fused AdamW over the flat parameter slabs + the reference's warmup-cosine LR schedule.

reference: optimiser construction base.py:416-441 (torch.optim.AdamW over param groups), schedule
src/utils/lr_scheduler.py:76-149 (LinearWarmupCosineAnnealingLR).  One HIP launch updates a whole run of
consecutive parameters (a full backbone or head slab when every parameter has a gradient) instead of
one multi-tensor op list; parameters with `grad is None` are skipped exactly as torch.optim does
(frozen prototypes during epoch 0, the online classifier)."""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import torch
from torch.optim.lr_scheduler import LRScheduler

from . import ops
from .flat import ALIGN, FlatParams


class DeviceHyper:
    """Per-step AdamW scalars in device memory, for launches that are captured once and replayed (chadavit_amd.graphed).

    One slot = {lr, 1 - beta1^t, sqrt(1 - beta2^t)} for the parameters of one param group that share a step count (the prototypes
    frozen during epoch 0 lag behind the rest of their group, exactly as torch.optim.AdamW counts steps per parameter); a slot is
    remembered by one representative parameter, so its step count is always read from the live optimizer state.  While an
    optimizer runs with `device_hyper` set it does NOT advance its per-parameter counters: parameter p is at step
    `state[p]["step"] + taken`; `advance()` after every (eager or replayed) step, `commit()` writes the counters back -- before
    another captured graph (another set of active parameters) takes over, or when leaving device mode."""

    def __init__(self, device, max_slots: int = 16):
        self.host = torch.zeros(3 * max_slots, dtype=torch.float32).pin_memory()
        self.dev = torch.zeros(3 * max_slots, dtype=torch.float32, device=device)
        self.slots: List[tuple] = []   # (group index, representative parameter)
        self.taken = 0                 # optimizer steps taken through this object since the last commit
        self.active: List = []         # parameters that are stepped through it (the current graph's)

    @staticmethod
    def _entry(group, t: int):
        """{lr, 1 - beta1^t, sqrt(1 - beta2^t)} with the same float32 roundings as the by-value path (ops.adamw_step ->
        chadavit_adamw_step: bias corrections rounded to float32, then sqrtf in float32), so that a replayed step is bit-identical to
        an eager one."""
        import numpy as np
        b1, b2 = group["betas"]
        return [float(group["lr"]), float(np.float32(1.0 - b1 ** t)), float(np.sqrt(np.float32(1.0 - b2 ** t)))]

    def slot(self, gi: int, p, state, optimizer_state, group=None) -> torch.Tensor:
        step = int(state.get("step", 0))
        i = next((k for k, (g, rep) in enumerate(self.slots) if g == gi and int(optimizer_state[rep].get("step", 0)) == step), None)
        if i is None:
            if 3 * (len(self.slots) + 1) > self.host.numel():
                raise RuntimeError("DeviceHyper: more (group, step count) classes than slots")
            self.slots.append((gi, p))
            i = len(self.slots) - 1
            if group is not None:
                # A slot is born inside the optimizer step that first needs it -- AFTER this step's fill() / upload.  It gets its
                # values here (host entry + a synchronous upload from pageable memory, ordered before the launch that reads it on the
                # current stream), so that no step ever runs on an unset {lr, bias corrections} = 0 (0 / 0: every parameter NaN).
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("DeviceHyper: a new (group, step count) class appeared during graph capture")
                vals = self._entry(group, step + self.taken + 1)
                for k, v in enumerate(vals):
                    self.host[3 * i + k] = v
                self.dev[3 * i:3 * i + 3].copy_(torch.tensor(vals, dtype=torch.float32))
        if not any(q is p for q in self.active):
            self.active.append(p)
        return self.dev[3 * i:3 * i + 3]

    def fill(self, optimizer) -> None:
        """Host values for the NEXT step (uploaded before the step's launches)."""
        for i, (gi, rep) in enumerate(self.slots):
            t = int(optimizer.state[rep].get("step", 0)) + self.taken + 1
            for k, v in enumerate(self._entry(optimizer.param_groups[gi], t)):
                self.host[3 * i + k] = v

    def advance(self) -> None:
        self.taken += 1

    def commit(self, optimizer) -> None:
        """Write the step counters of the active parameters back (`taken` steps each) and start counting from zero again."""
        for p in self.active:
            st = optimizer.state[p]
            st["step"] = int(st.get("step", 0)) + self.taken
        self.taken = 0

    def switch(self, optimizer, slots: List[tuple], active: List) -> None:
        """Another captured graph takes over: counters written back, then ITS slot table and active set."""
        self.commit(optimizer)
        self.slots, self.active = list(slots), list(active)


class _SlabState:
    """`state_dict()` / `load_state_dict()` in torch's own layout for optimisers that keep their moments in slabs beside the flat
    parameter slabs: per parameter (indexed in param-group order, as torch packs them) `exp_avg` / `exp_avg_sq` / `step`
    (torch.optim.Adam[W]) or `momentum_buffer` (torch.optim.SGD, the reference's LARS) -- what Lightning writes into its checkpoints
    and what a checkpoint of the reference's optimiser holds, so a run can be resumed either way.  (With GraphedTrainStep the step
    counters live on the device: `close()` it before saving.)"""

    def _slab_views(self, f, n):   # -> [(torch state key, view of the slab shaped like the parameter)]
        raise NotImplementedError

    def _after_load(self, st):     # optimiser-specific flags implied by a loaded state
        pass

    def state_dict(self):
        self._index()
        added = []
        for group in self.param_groups:
            for p in group["params"]:
                loc, st = self._where.get(id(p)), self.state.get(p)
                if loc is None or not st:
                    continue
                for key, view in self._slab_views(*loc):
                    st[key] = view.detach().clone()
                    added.append((st, key))
        try:
            sd = super().state_dict()
            sd["state"] = {k: dict(v) for k, v in sd["state"].items()}   # (torch hands out the live per-parameter dicts)
            return sd
        finally:
            for st, key in added:
                del st[key]

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._index()
        for p, st in list(self.state.items()):
            if isinstance(st.get("step"), torch.Tensor):
                st["step"] = int(st["step"].item())
            self._after_load(st)
            loc = self._where.get(id(p))
            if loc is None:
                continue
            for key, view in self._slab_views(*loc):
                if key in st:
                    view.copy_(st.pop(key))


class FusedAdamW(_SlabState, torch.optim.Optimizer):
    decoupled = True   # AdamW; FusedAdam below: torch.optim.Adam's L2 form

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, modules: Iterable = ()):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._modules = list(modules)
        self._slabs: Dict[int, dict] = {}  # id(FlatParams) -> {"m": tensor, "v": tensor}
        self._where: Dict[int, tuple] = {}
        # device_hyper: a DeviceHyper.  When set, step() launches the kernels that read {lr, bias corrections} from device memory and
        # leaves the per-parameter step counters to that object (chadavit_amd.graphed.GraphedTrainStep: the launches are captured
        # once and replayed with new values)
        self.device_hyper: Optional[DeviceHyper] = None

    def _slab_views(self, f, n):
        sl = self._slabs[id(f)]
        return [("exp_avg", f.view(sl["m"], n)), ("exp_avg_sq", f.view(sl["v"], n))]

    def _index(self):
        """param id -> (FlatParams, name); rebuilt if a module re-created its slab."""
        flats = [m.flat_params() for m in self._modules]
        if self._where and all(id(f) in self._slabs for f in flats):
            return
        self._where = {}
        for f in flats:
            if id(f) not in self._slabs:
                self._slabs[id(f)] = {"m": torch.zeros_like(f.flat), "v": torch.zeros_like(f.flat)}
            for n, p in zip(f.names, f.params):
                self._where[id(p)] = (f, n)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._index()
        touched = set()
        for gi, group in enumerate(self.param_groups):
            lr, wd, eps = group["lr"], group["weight_decay"], group["eps"]
            b1, b2 = group["betas"]
            dh = self.device_hyper
            if dh is not None and not self.decoupled:
                raise RuntimeError("device-resident hyper-parameters (GraphedTrainStep) are built for AdamW")
            step_fn = ops.adamw_step if self.decoupled else ops.adam_step
            runs: List[list] = []  # [flat, begin, end, step | device slot]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                hyper = None
                if dh is None:
                    st["step"] = st.get("step", 0) + 1
                else:
                    hyper = dh.slot(gi, p, st, self.state, group)
                loc = self._where.get(id(p))
                if loc is None:  # parameter outside the flat slabs (e.g. online classifier if it ever gets a gradient)
                    if "exp_avg" not in st:
                        st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32)
                        st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32)
                    if hyper is not None:
                        ops.adamw_step_dev(p.data.view(-1), p.grad.contiguous().view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1),
                                           hyper, b1, b2, eps, wd)
                    else:
                        step_fn(p.data.view(-1), p.grad.contiguous().view(-1), st["exp_avg"].view(-1),
                                st["exp_avg_sq"].view(-1), lr, b1, b2, eps, wd, st["step"])
                    continue
                f, n = loc
                gv = f.g(n)
                if p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
                beg = f.offsets[n]
                end = beg + (p.numel() + ALIGN - 1) // ALIGN * ALIGN
                tag = st["step"] if hyper is None else hyper
                same = (runs[-1][3] == tag) if (runs and hyper is None) else (runs and runs[-1][3].data_ptr() == tag.data_ptr())
                if runs and runs[-1][0] is f and runs[-1][2] == beg and same:
                    runs[-1][2] = end
                else:
                    runs.append([f, beg, end, tag])
            for f, beg, end, step in runs:
                sl = self._slabs[id(f)]
                if dh is not None:
                    ops.adamw_step_dev(f.flat[beg:end], f.grad[beg:end], sl["m"][beg:end], sl["v"][beg:end], step, b1, b2, eps, wd)
                else:
                    step_fn(f.flat[beg:end], f.grad[beg:end], sl["m"][beg:end], sl["v"][beg:end], lr, b1, b2, eps, wd, step)
                touched.add(id(f))
        for m in self._modules:
            f = m.flat_params()
            if id(f) in touched:
                f.mark_dirty()
        return loss


class FusedAdam(FusedAdamW):
    """torch.optim.Adam (the reference's "adam", base.py:67-72): weight decay as an L2 term of the gradient; defaults as torch's."""
    decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, modules: Iterable = ()):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, modules=modules)


class FusedSGD(_SlabState, torch.optim.Optimizer):
    """torch.optim.SGD (the reference's "sgd", base.py:67-72) over the flat slabs: one launch per run of consecutive parameters that
    share a step count.  Momentum buffers live in a slab of the parameters' layout."""

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, modules: Iterable = ()):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov))
        self._modules = list(modules)
        self._bufs: Dict[int, torch.Tensor] = {}
        self._where: Dict[int, tuple] = {}

    def _slab_views(self, f, n):
        return [("momentum_buffer", f.view(self._bufs[id(f)], n))] if self.defaults["momentum"] != 0 else []

    def _after_load(self, st):
        if "momentum_buffer" in st:
            st["seen"] = True

    def _index(self):
        flats = [m.flat_params() for m in self._modules]
        if self._where and all(id(f) in self._bufs for f in flats):
            return
        self._where = {}
        for f in flats:
            self._bufs.setdefault(id(f), torch.zeros_like(f.flat))
            for n, p in zip(f.names, f.params):
                self._where[id(p)] = (f, n)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._index()
        touched = set()
        for group in self.param_groups:
            lr, mu, damp, wd, nest = group["lr"], group["momentum"], group["dampening"], group["weight_decay"], group["nesterov"]
            runs: List[list] = []  # [flat, begin, end, first]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                first = "seen" not in st
                st["seen"] = True
                loc = self._where.get(id(p))
                if loc is None:
                    if mu != 0 and "momentum_buffer" not in st:
                        st["momentum_buffer"] = torch.zeros_like(p, dtype=torch.float32)
                    ops.sgd_step(p.data.view(-1), p.grad.contiguous().view(-1), st["momentum_buffer"].view(-1) if mu != 0 else None, lr, mu, damp,
                                 wd, nest, first)
                    continue
                f, n = loc
                gv = f.g(n)
                if p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
                beg = f.offsets[n]
                end = beg + (p.numel() + ALIGN - 1) // ALIGN * ALIGN
                if runs and runs[-1][0] is f and runs[-1][2] == beg and runs[-1][3] == first:
                    runs[-1][2] = end
                else:
                    runs.append([f, beg, end, first])
            for f, beg, end, first in runs:
                ops.sgd_step(f.flat[beg:end], f.grad[beg:end], self._bufs[id(f)][beg:end] if mu != 0 else None, lr, mu, damp, wd, nest, first)
                touched.add(id(f))
        for m in self._modules:
            f = m.flat_params()
            if id(f) in touched:
                f.mark_dirty()
        return loss


class FusedLARS(_SlabState, torch.optim.Optimizer):
    """LARS with the reference's semantics (src/utils/lars.py:112-167): layer-wise trust ratio eta*|p|/(|g| + wd*|p| + eps) and
    weight decay only where scaling applies; SGD momentum (PyTorch convention), optional Nesterov, clip_lr and
    exclude_bias_n_norm.  One HIP launch per (param group, flat slab): a block per tensor computes both norms and applies
    the update."""

    def __init__(self, params, lr, momentum=0, dampening=0, weight_decay=0, nesterov=False, eta=1e-3, eps=1e-8, clip_lr=False,
                 exclude_bias_n_norm=False, modules: Iterable = ()):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid LARS hyper-parameter")
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov, eta=eta,
                        eps=eps, clip_lr=clip_lr, exclude_bias_n_norm=exclude_bias_n_norm)
        super().__init__(params, defaults)
        self._modules = list(modules)
        self._bufs: Dict[int, torch.Tensor] = {}
        self._tables: Dict[tuple, tuple] = {}
        self._where: Dict[int, tuple] = {}

    def _slab_views(self, f, n):
        return [("momentum_buffer", f.view(self._bufs[id(f)], n))]

    def _after_load(self, st):
        if "momentum_buffer" in st:
            st["init"] = True

    def _index(self):
        flats = [m.flat_params() for m in self._modules]
        if self._where and all(id(f) in self._bufs for f in flats):
            return
        self._where = {}
        for f in flats:
            self._bufs.setdefault(id(f), torch.zeros_like(f.flat))
            for n, p in zip(f.names, f.params):
                self._where[id(p)] = (f, n)

    def _launch(self, flat, grad, buf, offs, sizes, flags, group):
        key = (flat.device, offs, sizes, flags)
        tabs = self._tables.get(key)
        if tabs is None:  # at most two flag patterns per slab (first step / later steps): cached, no per-step upload
            tabs = (torch.tensor(offs, dtype=torch.int64, device=flat.device), torch.tensor(sizes, dtype=torch.int64, device=flat.device),
                    torch.tensor(flags, dtype=torch.int32, device=flat.device))
            self._tables[key] = tabs
        ops.lars_step(flat, grad, buf, tabs[0], tabs[1], tabs[2], group["lr"], group["momentum"], group["dampening"],
                      group["weight_decay"], group["eta"], group["eps"], group["clip_lr"], group["nesterov"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._index()
        touched = {}
        for group in self.param_groups:
            per_flat: Dict[int, list] = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                flag = (1 if (p.ndim != 1 or not group["exclude_bias_n_norm"]) else 0) | (2 if st.get("init") else 0)
                st["init"] = True
                loc = self._where.get(id(p))
                if loc is None:  # parameter outside the flat slabs (the online classifier, if it ever gets a gradient)
                    if "momentum_buffer" not in st:
                        st["momentum_buffer"] = torch.zeros_like(p, dtype=torch.float32)
                    self._launch(p.data.view(-1), p.grad.contiguous().view(-1), st["momentum_buffer"].view(-1), (0,), (p.numel(),),
                                 (flag,), group)
                    continue
                f, n = loc
                gv = f.g(n)
                if p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
                e = per_flat.setdefault(id(f), [f, [], [], []])
                e[1].append(f.offsets[n]); e[2].append(p.numel()); e[3].append(flag)
            for f, offs, sizes, flags in per_flat.values():
                self._launch(f.flat, f.grad, self._bufs[id(f)], tuple(offs), tuple(sizes), tuple(flags), group)
                touched[id(f)] = f
        for f in touched.values():
            f.mark_dirty()
        return loss


def remove_bias_and_norm_from_weight_decay(parameter_groups):
    """Per group, move parameters with ndim <= 1 into a `<name>_no_decay` group with weight_decay 0 (decay group first)
    -- reference src/utils/misc.py:425-454."""
    out = []
    for group in parameter_groups:
        decay = {k: v for k, v in group.items() if k != "params"}
        no_decay = {k: v for k, v in group.items() if k != "params"}
        no_decay["weight_decay"] = 0
        if group.get("name", None):
            no_decay["name"] = group["name"] + "_no_decay"
        dp, ndp = [], []
        for p in group["params"]:
            (ndp if p.ndim <= 1 else dp).append(p)
        if dp:
            decay["params"] = dp
            out.append(decay)
        if ndp:
            no_decay["params"] = ndp
            out.append(no_decay)
    return out


class WarmupCosineLR(LRScheduler):
    """Closed form of the reference LinearWarmupCosineAnnealingLR (lr_scheduler.py:127-149); stepping it
    once per optimiser step reproduces the chainable sequence of :76-125."""

    def __init__(self, optimizer, warmup_epochs, max_epochs, warmup_start_lr=0.0, eta_min=0.0, last_epoch=-1):
        self.warmup_epochs = warmup_epochs
        self.max_epochs = max_epochs
        self.warmup_start_lr = warmup_start_lr
        self.eta_min = eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        t = self.last_epoch
        if t < self.warmup_epochs:
            return [self.warmup_start_lr + t * (b - self.warmup_start_lr) / (self.warmup_epochs - 1) for b in self.base_lrs]
        return [self.eta_min + 0.5 * (b - self.eta_min) * (1 + math.cos(math.pi * (t - self.warmup_epochs) / (self.max_epochs - self.warmup_epochs)))
                for b in self.base_lrs]
