"""Regression / fine-tune evaluation on the HIP engine: the reference's `RegressionModel` surface (src/methods/regression.py:40-516,
driver main_regression.py) for the ChAda-ViT path -- `LinearModel` (chadavit_amd.methods.linear) with ONE output node named
`regressor`, `nn.MSELoss` by default, targets unsqueezed to (B, 1) (regression.py:420-427), and R^2 / MSE / MAE / Pearson r of the batch
in place of the accuracies (the torchmetrics objects' per-batch values, computed directly)."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Tuple

import torch
import torch.nn as nn

from .linear import LinearModel, weighted_mean


def regression_metrics(out: torch.Tensor, target: torch.Tensor) -> Dict[str, torch.Tensor]:
    """R2Score / MeanSquaredError / MeanAbsoluteError / PearsonCorrCoef of one batch (torchmetrics' definitions)."""
    with torch.no_grad():
        o, t = out.detach().float().view(-1), target.detach().float().view(-1)
        err = o - t
        mse, mae = (err * err).mean(), err.abs().mean()
        ss_tot = ((t - t.mean()) ** 2).sum()
        r2 = 1.0 - (err * err).sum() / ss_tot
        oc, tc = o - o.mean(), t - t.mean()
        pcc = (oc * tc).sum() / (oc.norm() * tc.norm())
        return {"r2": r2, "mse": mse, "mae": mae, "pcc": pcc}


class RegressionModel(LinearModel):
    def __init__(self, backbone: nn.Module, cfg, loss_func: Optional[Callable] = None, mixup_func: Optional[Callable] = None):
        cfg.data.num_classes = 1                       # "for simple regression" (regression.py:134): one target node
        super().__init__(backbone, cfg, loss_func=loss_func if loss_func is not None else nn.MSELoss(), mixup_func=mixup_func)
        self.num_target_nodes = 1
        self.regressor = self.classifier               # the reference's parameter names: regressor.{weight,bias}
        del self.classifier

    @property
    def out_layer(self) -> nn.Linear:
        return self.regressor

    def shared_step(self, batch: Tuple, batch_idx: int, index: int) -> Dict[str, Any]:
        X, target, list_num_channels = batch
        self.list_num_channels = [list_num_channels] if isinstance(list_num_channels[0], int) else list_num_channels
        metrics: Dict[str, Any] = {"batch_size": X.size(0)}
        if self.training and self.mixup_func is not None:
            X, target = self.mixup_func(X, target)
        out = self(X, index)["logits"]
        target = target.unsqueeze(1).to(out.dtype)
        metrics["loss"] = self.loss_func(out, target)
        if not (self.training and self.mixup_func is not None):
            metrics.update(regression_metrics(out, target))
        return metrics

    def training_step(self, batch, batch_idx: int) -> torch.Tensor:
        if not self.finetune:
            self.backbone.eval()
        out = self.shared_step(batch, batch_idx, index=0)
        log = {"train_loss": out["loss"]}
        if self.mixup_func is None:
            log.update({"train_" + k: out[k] for k in ("r2", "mse", "mae", "pcc")})
        self.log_dict(log, on_epoch=True, sync_dist=True)
        return out["loss"]

    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int) -> Dict[str, Any]:
        out = self.shared_step(batch, batch_idx, index=0)
        metrics = {"batch_size": out["batch_size"], "val_loss": out["loss"], **{"val_" + k: out[k] for k in ("r2", "mse", "mae", "pcc")}}
        self.validation_step_metrics.append(metrics)
        return metrics

    def on_validation_epoch_end(self):
        log = {k: weighted_mean(self.validation_step_metrics, k, "batch_size") for k in ("val_loss", "val_r2", "val_mse", "val_mae", "val_pcc")}
        self.validation_step_metrics.clear()
        self.log_dict(log, sync_dist=True)
