"""Linear / fine-tune evaluation on the HIP engine (`main_linear.py`-style evaluation on the inference path).

Mirrors the LightningModule surface of the reference's `LinearModel` (src/methods/linear.py:50-628) for the ChAda-ViT path
(`channels_strategy: multi_channels`): `LinearModel(backbone, cfg, loss_func=None, mixup_func=None)`, static
`add_and_assert_specific_cfg`, `configure_optimizers`, `forward(X, index) -> {"logits", "feats"}`, `shared_step`,
`training_step`, `validation_step`, `on_validation_epoch_end`; batch format `(X (sum C,1,S,S), targets (B,), [[C_i]])`
(one_channel_collate_fn with one crop, channels_strategies.py:31-85).

What runs where:
  * the backbone is `chadavit_amd...ChAdaViT` (HIP; frozen and gradient-free unless `finetune`), CLS features `(B, D)` or, with
    `return_all_tokens`, every valid patch token flattened per image `(B, C*p*D)` (linear.py:399-427);
  * the classifier `nn.Linear(features_dim, num_classes)` runs on the bf16 MFMA GEMMs (`chadavit_gemm_nt` forward and dX,
    `chadavit_gemm_tn` for dW / db) with its rows padded to the kernels' 64-column granule;
  * the loss on the (B, num_classes) logits -- `F.cross_entropy`, or the caller's `loss_func` with `mixup_func` -- and the
    accuracies are a few hundred floats of torch arithmetic.

Reference behaviour kept on purpose: `metrics["batch_size"]` is `X.size(0)`, the number of CHANNEL images, not of images
(linear.py:454); with `mixed_channels: False` and `return_all_tokens` an unequal channel count per image fails (the reference's
`torch.stack`, linear.py:421).  Not built: `layer_decay` (timm's `param_groups_layer_decay`), the torchmetrics macro metrics
(recall / precision / AUROC / F1 objects) and the seaborn confusion-matrix figure -- the confusion matrix itself is logged.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..backbones.vit.chada_vit import ChAdaViT
from ..utils.misc import omegaconf_select
from .dino import _Base


def accuracy_at_k(outputs: torch.Tensor, targets: torch.Tensor, top_k: Sequence[int] = (1, 5)) -> List[torch.Tensor]:
    """Percentage of rows whose target is among the k largest outputs (src/utils/metrics.py:26-52): 1-element tensors."""
    with torch.no_grad():
        maxk = min(max(top_k), outputs.shape[1])   # (the reference's topk(5) fails below five classes)
        pred = outputs.topk(maxk, 1, True, True)[1]
        hit = pred.eq(targets.view(-1, 1))
        n = targets.size(0)
        return [hit[:, :k].any(1).float().sum().view(1) * (100.0 / n) for k in top_k]


def weighted_mean(outputs: List[Dict], key: str, batch_size_key: str):
    """Mean of `key` weighted by `batch_size_key` over a list of step outputs (src/utils/metrics.py:55-73)."""
    value, n = 0, 0
    for out in outputs:
        value = value + out[batch_size_key] * out[key]
        n += out[batch_size_key]
    value = value / n
    return value.squeeze(0) if isinstance(value, torch.Tensor) else value


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


class _ClassifierFn(torch.autograd.Function):
    """logits = feats @ W^T + b on the MFMA GEMMs.  bf16 operands, fp32 accumulate and fp32 logits; W's rows padded with zero
    rows to a multiple of 64 (the GEMM's column granule), the padding sliced off the result and its gradient rows never read."""

    @staticmethod
    def forward(ctx, feats, weight, bias, owner):
        N, K = weight.shape
        wb, bp = owner._operands()
        xb = feats.to(torch.bfloat16).contiguous()
        out = ops.gemm_nt(xb, wb, bias=bp, out_fp32=True)
        ctx.owner, ctx.N = owner, N
        ctx.save_for_backward(xb, wb)
        ctx.feat_dtype = feats.dtype
        return out[:, :N].contiguous()

    @staticmethod
    def backward(ctx, dlogits):
        xb, wb = ctx.saved_tensors
        owner, N = ctx.owner, ctx.N
        Np, K = wb.shape
        B = xb.shape[0]
        dl = torch.zeros((B, Np), device=xb.device, dtype=torch.bfloat16)
        dl[:, :N] = dlogits
        dw = db = dx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dwp = torch.empty((Np, K), device=xb.device, dtype=torch.float32)
            dbp = torch.empty((Np,), device=xb.device, dtype=torch.float32)
            ops.gemm_tn(dl, xb, dwp, colsum=dbp, accumulate=False, workspace=owner._workspace(Np, K))
            dw, db = dwp[:N], dbp[:N]
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dl, wb.t().contiguous(), out_fp32=True).to(ctx.feat_dtype)
        return dx, dw, db, None


class LinearModel(_Base):
    _OPTIMIZERS = ("sgd", "lars", "adam", "adamw")                               # linear.py:51-56
    _SCHEDULERS = ["reduce", "warmup_cosine", "step", "exponential", "none"]      # linear.py:57-63

    def __init__(self, backbone: nn.Module, cfg, loss_func: Optional[Callable] = None, mixup_func: Optional[Callable] = None):
        super().__init__()
        cfg = self.add_and_assert_specific_cfg(cfg)
        if cfg.channels_strategy != "multi_channels" or not isinstance(backbone, ChAdaViT):
            raise RuntimeError("chadavit_amd LinearModel: the ChAdaViT backbone with channels_strategy 'multi_channels' only "
                               "(linear.py:386-392); the one_channel / timm-ViT branches are outside the ChAda path")
        self.backbone = backbone
        features_dim = backbone.num_features
        self.return_all_tokens = bool(cfg.backbone.kwargs.return_all_tokens)
        if self.return_all_tokens:   # every patch token of every channel, flattened per image (linear.py:133-138)
            features_dim = cfg.data.img_channels * backbone.token_learner.num_patches * backbone.embed_dim
        self.features_dim = int(features_dim)
        self.num_classes = int(cfg.data.num_classes)
        self.classifier = nn.Linear(self.features_dim, self.num_classes)
        self.mixup_func, self.loss_func = mixup_func, (loss_func if loss_func is not None else nn.CrossEntropyLoss())
        self.max_epochs = cfg.max_epochs
        self.accumulate_grad_batches = cfg.accumulate_grad_batches
        self.optimizer, self.batch_size, self.lr = cfg.optimizer.name, cfg.optimizer.batch_size, cfg.optimizer.lr
        self.weight_decay = cfg.optimizer.weight_decay
        self.extra_optimizer_args = dict(cfg.optimizer.kwargs)
        self.exclude_bias_n_norm_wd = cfg.optimizer.exclude_bias_n_norm_wd
        self.layer_decay = cfg.optimizer.layer_decay
        if self.layer_decay > 0:
            raise RuntimeError("chadavit_amd LinearModel: optimizer.layer_decay needs timm's param_groups_layer_decay; not built")
        self.scheduler = cfg.scheduler.name
        self.lr_decay_steps, self.min_lr = cfg.scheduler.lr_decay_steps, cfg.scheduler.min_lr
        self.warmup_start_lr, self.warmup_epochs = cfg.scheduler.warmup_start_lr, cfg.scheduler.warmup_epochs
        self.scheduler_interval = cfg.scheduler.interval
        assert self.scheduler_interval in ["step", "epoch"]
        self.finetune = bool(cfg.finetune)
        if not self.finetune:
            for p in self.backbone.parameters():
                p.requires_grad = False
        self.validation_step_metrics: List[Dict] = []
        self.validation_step_targets: List[torch.Tensor] = []
        self.validation_step_preds: List[torch.Tensor] = []
        self.channels_strategy, self.mixed_channels = cfg.channels_strategy, cfg.mixed_channels
        self.list_num_channels: List[List[int]] = []
        self.confusion_matrix = None
        self._ws = None

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def add_and_assert_specific_cfg(cfg):
        """Defaults of linear.py:232-281."""
        from ..utils.misc import ensure_node
        for node in ("optimizer", "scheduler", "performance"):
            ensure_node(cfg, node)
        cfg.optimizer.exclude_bias_n_norm_wd = omegaconf_select(cfg, "optimizer.exclude_bias_n_norm_wd", False)
        cfg.optimizer.kwargs = omegaconf_select(cfg, "optimizer.kwargs", {})
        cfg.optimizer.layer_decay = omegaconf_select(cfg, "optimizer.layer_decay", 0.0)
        cfg.finetune = omegaconf_select(cfg, "finetune", False)
        cfg.accumulate_grad_batches = omegaconf_select(cfg, "accumulate_grad_batches", 1)
        cfg.scheduler.lr_decay_steps = omegaconf_select(cfg, "scheduler.lr_decay_steps", None)
        cfg.scheduler.min_lr = omegaconf_select(cfg, "scheduler.min_lr", 0.0)
        cfg.scheduler.warmup_start_lr = omegaconf_select(cfg, "scheduler.warmup_start_lr", 3e-5)
        cfg.scheduler.warmup_epochs = omegaconf_select(cfg, "scheduler.warmup_epochs", 10)
        cfg.scheduler.interval = omegaconf_select(cfg, "scheduler.interval", "step")
        cfg.performance.disable_channel_last = omegaconf_select(cfg, "performance.disable_channel_last", False)
        cfg.channels_strategy = omegaconf_select(cfg, "channels_strategy", None)
        cfg.mixed_channels = omegaconf_select(cfg, "mixed_channels", False)
        return cfg

    def configure_optimizers(self):
        """linear.py:283-371: the classifier alone, or {backbone, classifier} groups when fine-tuning; the fused optimisers of
        chadavit_amd.optim; schedulers warmup_cosine (per step or per epoch) | reduce | step | exponential | none."""
        from torch.optim.lr_scheduler import ExponentialLR, MultiStepLR, ReduceLROnPlateau
        from ..optim import FusedAdam, FusedAdamW, FusedLARS, FusedSGD, WarmupCosineLR, remove_bias_and_norm_from_weight_decay
        lin_name = "classifier" if self.out_layer is getattr(self, "classifier", None) else "regressor"
        if not self.finetune:
            groups: List[Dict[str, Any]] = [{"params": list(self.out_layer.parameters())}]
        else:
            groups = [{"name": "backbone", "params": list(self.backbone.parameters())},
                      {"name": lin_name, "params": list(self.out_layer.parameters())}]
        if self.exclude_bias_n_norm_wd:
            groups = remove_bias_and_norm_from_weight_decay(groups)
        assert self.optimizer in self._OPTIMIZERS
        kw = dict(self.extra_optimizer_args)
        if "betas" in kw:
            kw["betas"] = tuple(kw["betas"])
        cls = {"sgd": FusedSGD, "lars": FusedLARS, "adam": FusedAdam, "adamw": FusedAdamW}[self.optimizer]
        opt = cls(groups, lr=self.lr, weight_decay=self.weight_decay, modules=[self.backbone] if self.finetune else [], **kw)
        if self.scheduler == "none":
            return opt
        if self.scheduler == "warmup_cosine":
            total = self.trainer.estimated_stepping_batches
            warm = self.warmup_epochs * (total / self.max_epochs) if self.scheduler_interval == "step" else self.warmup_epochs
            steps = total if self.scheduler_interval == "step" else self.max_epochs
            sched: Any = {"scheduler": WarmupCosineLR(opt, warmup_epochs=warm, max_epochs=steps,
                                                      warmup_start_lr=self.warmup_start_lr if self.warmup_epochs > 0 else self.lr,
                                                      eta_min=self.min_lr),
                          "interval": self.scheduler_interval, "frequency": 1}
        elif self.scheduler == "reduce":
            sched = ReduceLROnPlateau(opt)
        elif self.scheduler == "step":
            sched = MultiStepLR(opt, self.lr_decay_steps, gamma=0.1)
        elif self.scheduler == "exponential":
            sched = ExponentialLR(opt, self.weight_decay)
        else:
            raise ValueError(f"{self.scheduler} not in (warmup_cosine, cosine, reduce, step, exponential)")
        return [opt], [sched]

    # ------------------------------------------------------------------------------------------
    def _operands(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """bf16 copy of the classifier weight with zero rows up to a multiple of 64, and the padded fp32 bias.  Rebuilt at every
        call (num_classes x features_dim elements): the fused optimisers update parameters in place through raw pointers, which
        no version counter sees."""
        w, b = self.out_layer.weight, self.out_layer.bias
        Np = _pad64(w.shape[0])
        wb = torch.zeros((Np, w.shape[1]), device=w.device, dtype=torch.bfloat16)
        wb[: w.shape[0]] = w.detach()
        bp = torch.zeros((Np,), device=w.device, dtype=torch.float32)
        bp[: w.shape[0]] = b.detach()
        return wb, bp

    @property
    def out_layer(self) -> nn.Linear:
        """The trained linear layer (`classifier` here, `regressor` in the regression subclass)."""
        return self.classifier

    def _workspace(self, Np: int, K: int) -> torch.Tensor:
        need = 2 * (Np * K + Np)
        dev = self.out_layer.weight.device
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, device=dev, dtype=torch.float32)
        return self._ws

    def forward(self, X: torch.Tensor, index: int) -> Dict[str, Any]:
        """Backbone (gradient-free unless fine-tuning) + linear layer (linear.py:373-432)."""
        if X.device.type != "cuda":
            raise RuntimeError("chadavit_amd has no CPU path")
        lnc = self.list_num_channels
        with torch.set_grad_enabled(self.finetune and torch.is_grad_enabled()):
            feats = self.backbone(X, index, lnc)
            if not self.mixed_channels and self.return_all_tokens:
                counts = lnc[index]
                if len(set(counts)) != 1:
                    raise RuntimeError("LinearModel: return_all_tokens without mixed_channels needs the same channel count in "
                                       f"every image (the reference stacks per-image chunks, linear.py:414-421); got {sorted(set(counts))}")
                feats = feats.reshape(len(counts), -1)
        if feats.shape[1] != self.features_dim:
            raise RuntimeError(f"LinearModel: features of width {feats.shape[1]} for a classifier of {self.features_dim} inputs "
                               "(data.img_channels / return_all_tokens / mixed_channels do not describe this batch)")
        lin = self.out_layer
        if torch.is_grad_enabled() and (feats.requires_grad or lin.weight.requires_grad):
            logits = _ClassifierFn.apply(feats, lin.weight, lin.bias, self)
        else:
            wb, bp = self._operands()
            logits = ops.gemm_nt(feats.to(torch.bfloat16).contiguous(), wb, bias=bp, out_fp32=True)[:, : self.num_classes].contiguous()
        return {"logits": logits, "feats": feats}

    def shared_step(self, batch: Tuple, batch_idx: int, index: int) -> Dict[str, Any]:
        """linear.py:434-511: loss + accuracies (or the mixup loss alone while training with a mixup function)."""
        X, target, list_num_channels = batch
        self.list_num_channels = [list_num_channels] if isinstance(list_num_channels[0], int) else list_num_channels
        metrics: Dict[str, Any] = {"batch_size": X.size(0)}   # channel images, as the reference counts them
        if self.training and self.mixup_func is not None:
            X, target = self.mixup_func(X, target)
            out = self(X, index)["logits"]
            metrics.update({"loss": self.loss_func(out, target)})
        else:
            out = self(X, index)["logits"]
            loss = F.cross_entropy(out, target)
            acc1, acc5 = accuracy_at_k(out, target, top_k=(1, 5))
            metrics.update({"loss": loss, "acc1": acc1, "acc5": acc5})
        self.validation_step_targets.append(target if target.dim() == 1 else target.argmax(1))
        self.validation_step_preds.append(out.detach().argmax(1))
        return metrics

    def training_step(self, batch, batch_idx: int) -> torch.Tensor:
        if not self.finetune:
            self.backbone.eval()   # linear.py:525-526
        out = self.shared_step(batch, batch_idx, index=0)
        log = {"train_loss": out["loss"]}
        if self.mixup_func is None:
            log.update({"train_acc1": out["acc1"], "train_acc5": out["acc5"]})
        self.log_dict(log, on_epoch=True, sync_dist=True)
        return out["loss"]

    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int) -> Dict[str, Any]:
        out = self.shared_step(batch, batch_idx, index=0)
        metrics = {"batch_size": out["batch_size"], "val_loss": out["loss"], "val_acc1": out["acc1"], "val_acc5": out["acc5"]}
        self.validation_step_metrics.append(metrics)
        return metrics

    def on_validation_epoch_end(self):
        """Batch-size weighted means of the step metrics + the confusion matrix of every prediction since the last call
        (training steps included, as the reference collects them: linear.py:505-509, 577-628)."""
        log = {k: weighted_mean(self.validation_step_metrics, k, "batch_size") for k in ("val_loss", "val_acc1", "val_acc5")}
        pred, tgt = torch.cat(self.validation_step_preds), torch.cat(self.validation_step_targets)
        cm = torch.zeros((self.num_classes, self.num_classes), dtype=torch.long, device=pred.device)
        cm.view(-1).index_add_(0, tgt * self.num_classes + pred, torch.ones_like(pred))
        self.confusion_matrix = cm.cpu().numpy().astype(int)   # rows: target, columns: prediction
        self.validation_step_metrics.clear()
        self.validation_step_targets.clear()
        self.validation_step_preds.clear()
        self.log_dict(log, sync_dist=True)
