"""EMA of the teacher (reference: src/utils/momentum.py:26-87), one fused pass over the flat slab."""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import ops


def _flat_of(module: nn.Module):
    """FlatParams of a chadavit_amd module (ChAdaViT / DINOHead), or None for foreign modules."""
    fp = getattr(module, "flat_params", None)
    if fp is None:
        return None
    try:
        return fp()
    except RuntimeError:
        return None


def mark_params_dirty(module: nn.Module):
    for m in module.modules():
        f = getattr(m, "_flat", None)
        if f is not None:
            f.mark_dirty()


@torch.no_grad()
def initialize_momentum_params(online_net: nn.Module, momentum_net: nn.Module):
    """Copy online -> momentum and freeze it (momentum.py:26-40)."""
    for po, pm in zip(online_net.parameters(), momentum_net.parameters()):
        pm.data.copy_(po.data)
        pm.requires_grad = False
    mark_params_dirty(momentum_net)


class MomentumUpdater:
    def __init__(self, base_tau: float = 0.996, final_tau: float = 1.0):
        assert 0 <= base_tau <= 1
        assert 0 <= final_tau <= 1 and base_tau <= final_tau
        self.base_tau = base_tau
        self.cur_tau = base_tau
        self.final_tau = final_tau
        self.tau_dev = None   # device float32[1]: when set, the EMA kernel reads tau from it (graph-captured step)

    @torch.no_grad()
    def update(self, online_net: nn.Module, momentum_net: nn.Module):
        """theta_t <- tau*theta_t + (1-tau)*theta_s for every parameter pair (momentum.py:63-74)."""
        fo, fm = _flat_of(online_net), _flat_of(momentum_net)
        if fo is not None and fm is not None and fo.numel == fm.numel and fo.names == fm.names:
            if self.tau_dev is not None:
                ops.ema_update_dev(fm.flat, fo.flat, self.tau_dev)
            else:
                ops.ema_update(fm.flat, fo.flat, float(self.cur_tau))
            fm.mark_dirty()
            return
        for op, mp in zip(online_net.parameters(), momentum_net.parameters()):
            if mp.device.type != "cuda":
                raise RuntimeError("chadavit_amd EMA runs on the GPU only")
            ops.ema_update(mp.data.view(-1), op.data.contiguous().view(-1), float(self.cur_tau))
        mark_params_dirty(momentum_net)

    def update_tau(self, cur_step: int, max_steps: int):
        """Cosine schedule (momentum.py:76-87)."""
        self.cur_tau = self.final_tau - (self.final_tau - self.base_tau) * (math.cos(math.pi * cur_step / max_steps) + 1) / 2
