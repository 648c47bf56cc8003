"""DINO method on the HIP engine: student/teacher heads, EMA, centring + sharpened cross-entropy.

Mirrors the LightningModule hook surface of the reference (src/methods/dino.py:114-391 on top of
src/methods/base.py BaseMethod / BaseMomentumMethod hot-path pieces) so a
`main_pretrain.py`-style driver (or chadavit_amd.trainer) can call the same hooks in the same order:
`training_step -> backward -> on_after_backward -> optimizer.step -> optimizer_zero_grad -> on_train_batch_end`.
pytorch_lightning is optional: if importable the class subclasses LightningModule, otherwise a
minimal stand-in with `log` / `log_dict` / `current_epoch` / `trainer`.

MI355X-first differences (results identical):
  * the two global crops are packed into ONE ragged batch per network (student, teacher) and all local
    crops into one more, instead of one backbone call per crop (base.py:695-707, 1213-1218);
  * heads run on the bf16 MFMA GEMMs with fused GELU / GELU' epilogues; prototypes are weight-normalised
    once per parameter version.
Reference-parity crop semantics are kept: local crops run the student BACKBONE only and never reach the
loss (DINO does not override multicrop_forward).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..backbones import vit_channels
from ..backbones.vit.chada_vit import ChAdaViT, trunc_normal_
from ..data.channels_strategies import adjacent_view
from ..flat import FlatParams
from ..losses.dino import DINOLoss
from ..ragged import ragged_batch
from ..utils.misc import AttrDict, ensure_node, is_missing, omegaconf_select
from ..utils.momentum import MomentumUpdater, initialize_momentum_params

try:  # pragma: no cover - not installed in this image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    class _Base(nn.Module):
        """Minimal LightningModule stand-in (hooks are called by chadavit_amd.trainer)."""
        current_epoch: int = 0
        trainer: Any = None

        def log(self, name, value, *a, sync_dist: bool = False, **k):
            """Lightning's `self.log`.  sync_dist=True (dino.py:319: the training / validation loss) asks for the MEAN over
            ranks; a collective per logged value and step would put a rank rendezvous into the step, so the value is kept as a
            detached tensor and the reduction happens when the metrics are READ (`logged_metrics`): one all-reduce of all
            sync_dist values, off the training path."""
            self._logged = getattr(self, "_logged", {})
            self._logged_sync = getattr(self, "_logged_sync", set())
            self._logged[name] = value.detach() if isinstance(value, torch.Tensor) else value
            (self._logged_sync.add if sync_dist else self._logged_sync.discard)(name)

        def log_dict(self, d, *a, **k):
            for n, v in d.items():
                self.log(n, v, **k)

        def logged_metrics(self) -> Dict[str, float]:
            """Last logged value per name as floats; names logged with sync_dist=True are averaged over the ranks
            (collective: every rank has to call it, as with Lightning's own reduction)."""
            import torch.distributed as dist
            vals = dict(getattr(self, "_logged", {}))
            names = sorted(n for n in getattr(self, "_logged_sync", ()) if n in vals)
            if names and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dev = next((vals[n].device for n in names if isinstance(vals[n], torch.Tensor)), torch.device("cpu"))
                t = torch.stack([torch.as_tensor(vals[n], dtype=torch.float64, device=dev).reshape(()) for n in names])
                dist.all_reduce(t)
                t /= dist.get_world_size()
                for n, v in zip(names, t.tolist()):
                    vals[n] = v
            return {n: (float(v) if isinstance(v, (torch.Tensor, int, float)) else v) for n, v in vals.items()}


# ==============================================================================================
# DINO head
# ==============================================================================================
class DINOHead(nn.Module):
    """MLP(D -> hidden -> hidden -> bottleneck, GELU; optional BatchNorm1d after the first two Linears) -> L2 normalise ->
    weight-normed prototypes (reference dino.py:32-111).  Parameter names as the reference's nn.Sequential gives them:
    mlp.{0,2,4}.{weight,bias} without BatchNorm, mlp.{0,3,6} + BatchNorm at mlp.{1,4} (with their running_* buffers) with
    `use_bn` (cfg `method_kwargs.use_bn_in_head`); last_layer.weight_{g,v}."""

    def __init__(self, in_dim: int, num_prototypes: int, use_bn: bool = True, norm_last_layer: bool = True,
                 num_layers: int = 3, hidden_dim: int = 2048, bottleneck_dim: int = 256):
        super().__init__()
        if max(num_layers, 1) != 3:
            raise RuntimeError("chadavit_amd DINOHead: only the 3-layer projector of the reference configs is supported")
        self.use_bn = bool(use_bn)
        if self.use_bn:
            if hidden_dim % 4 != 0:
                raise RuntimeError("chadavit_amd DINOHead: use_bn needs hidden_dim % 4 == 0")
            self.mlp = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.BatchNorm1d(hidden_dim), nn.GELU(),
                                     nn.Linear(hidden_dim, hidden_dim), nn.BatchNorm1d(hidden_dim), nn.GELU(),
                                     nn.Linear(hidden_dim, bottleneck_dim))
            self._lin, self._bn = ("mlp.0", "mlp.3", "mlp.6"), ("mlp.1", "mlp.4")
        else:
            self.mlp = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, hidden_dim), nn.GELU(),
                                     nn.Linear(hidden_dim, bottleneck_dim))
            self._lin, self._bn = ("mlp.0", "mlp.2", "mlp.4"), ()
        self.apply(self._init_weights)
        self.last_layer = nn.utils.weight_norm(nn.Linear(bottleneck_dim, num_prototypes, bias=False))
        self.last_layer.weight_g.data.fill_(1)
        if norm_last_layer:
            self.last_layer.weight_g.requires_grad = False
        self._flat: Optional[FlatParams] = None
        self._proto = None  # (version, w bf16, w^T bf16, inv_norm)
        self._tn_ws = None
        self.skip_last_layer_grad = False  # set by DINO while epoch < freeze_last_layer (grads would be dropped)
        self.grad_ready_hook = None
        self._pending_backwards = 0

    @staticmethod
    def _init_weights(m: nn.Module):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def flat_params(self) -> FlatParams:
        dev = self.mlp[0].weight.device
        if dev.type != "cuda":
            raise RuntimeError("DINOHead (chadavit_amd) runs on the GPU only")
        if self._flat is None or self._flat.device != dev or not self._flat.attached():
            self._flat = FlatParams(list(self.named_parameters()), dev, transpose_names=[n + ".weight" for n in self._lin])
            self._proto = None
        return self._flat

    def _prototypes(self, flat: FlatParams):
        ver = flat._version()
        if self._proto is None or self._proto[0] != ver:
            w, wt, inv = ops.weightnorm_fwd(flat.f("last_layer.weight_v"), flat.f("last_layer.weight_g").view(-1))
            self._proto = (ver, w, wt, inv)
        return self._proto[1:]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.device.type != "cuda":
            raise RuntimeError("chadavit_amd has no CPU path")
        flat = self.flat_params()
        need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in flat.params))
        params = list(flat.params) if need_grad else []
        return _HeadFn.apply(self, x, need_grad, *params)


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head: DINOHead, x, need_grad, *params):
        flat = head.flat_params()
        flat.refresh(need_transposes=need_grad)
        w, wt, winv = head._prototypes(flat)
        xb = x.to(torch.bfloat16).contiguous()
        M = xb.shape[0]
        dev = x.device
        L0, L1, L2 = head._lin
        hid = flat.shapes[L0 + ".weight"][0]
        bn_saved = None
        if head.use_bn:
            # Linear -> BatchNorm1d (statistics over the rows of this call, dino.py:66-72) -> GELU, twice
            if need_grad and not head.training:
                raise RuntimeError("chadavit_amd DINOHead: gradients through BatchNorm in eval mode are not supported")
            bn_saved = []
            act = xb
            for li, bi in ((L0, 0), (L1, 1)):
                bn = head.mlp[int(head._bn[bi].split(".")[1])]
                z = ops.gemm_nt(act, flat.w(li + ".weight"), bias=flat.f(li + ".bias"), out_fp32=True)   # fp32: the statistics of a crop's few rows
                if head.training:
                    mean, rstd = ops.bn_stats(z, bn.eps, bn.running_mean, bn.running_var, bn.momentum if bn.momentum is not None else 0.1)
                    bn.num_batches_tracked += 1
                else:
                    mean, rstd = bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
                pre, nxt = ops.bn_apply_gelu(z, mean, rstd, flat.f(head._bn[bi] + ".weight"), flat.f(head._bn[bi] + ".bias"))
                bn_saved.append((z, mean, rstd))
                if bi == 0:
                    pre1, h1 = pre, nxt
                else:
                    pre2, h2 = pre, nxt
                act = nxt
        else:
            pre1 = torch.empty((M, hid), device=dev, dtype=torch.bfloat16)
            pre2 = torch.empty((M, hid), device=dev, dtype=torch.bfloat16)
            h1 = ops.gemm_nt(xb, flat.w(L0 + ".weight"), bias=flat.f(L0 + ".bias"), epilogue=ops.EPI_GELU, aux_out=pre1)
            h2 = ops.gemm_nt(h1, flat.w(L1 + ".weight"), bias=flat.f(L1 + ".bias"), epilogue=ops.EPI_GELU, aux_out=pre2)
        t = ops.gemm_nt(h2, flat.w(L2 + ".weight"), bias=flat.f(L2 + ".bias"), out_fp32=True)
        tn, tinv = ops.l2norm_fwd(t)
        logits = ops.gemm_nt(tn, w, out_fp32=True)
        ctx.head, ctx.need_grad, ctx.n_params = head, need_grad, len(params)
        ctx.x_needs_grad = x.requires_grad
        if need_grad:
            ctx.saved = (xb, pre1, h1, pre2, h2, t, tn, tinv, wt, winv)
            ctx.bn_saved = bn_saved
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        head: DINOHead = ctx.head
        if not ctx.need_grad:
            return (None,) * (3 + ctx.n_params)
        flat = head.flat_params()
        if ctx.saved is None:
            raise RuntimeError("second backward through the same DINOHead pass (retain_graph is not supported)")
        xb, pre1, h1, pre2, h2, t, tn, tinv, wt, winv = ctx.saved
        dev = dlogits.device
        G = flat.g
        if head._tn_ws is None or head._tn_ws.device != dev:
            P, K = flat.shapes["last_layer.weight_v"]
            hid = flat.shapes[head._lin[1] + ".weight"][0]
            # eight T-splits of the widest gradient: one per XCD (chadavit_gemm_tn sends split s to XCD s % 8; with room for two, the
            # prototype layer's gradient ran on two XCDs)
            head._tn_ws = torch.empty(8 * max(P * K + P, hid * hid + hid), device=dev, dtype=torch.float32)
        ws = head._tn_ws
        L0, L1, L2 = head._lin
        w0 = head.mlp[0].weight
        acc = w0.grad is not None and w0.grad.data_ptr() == G(L0 + ".weight").data_ptr()
        if w0.grad is not None and not acc:
            for n, p in zip(flat.names, flat.params):
                if p.grad is not None:
                    G(n).copy_(p.grad)
            acc = True
        dl = dlogits.to(torch.bfloat16).contiguous()
        dtn = ops.gemm_nt(dl, wt, out_fp32=True)
        vname = "last_layer.weight_v"
        if flat.params[flat.names.index(vname)].requires_grad and not head.skip_last_layer_grad:
            P, K = flat.shapes[vname]
            dw = torch.empty((P, K), device=dev, dtype=torch.float32)
            ops.gemm_tn(dl, tn, dw, accumulate=False, workspace=ws)
            gname = "last_layer.weight_g"   # trainable only with method_kwargs.norm_last_layer = False (dino.py:83-84)
            dg = G(gname).view(-1) if flat.params[flat.names.index(gname)].requires_grad else None
            ops.weightnorm_bwd(dw, flat.f(vname), flat.f(gname).view(-1), winv, G(vname), accumulate=acc, dg=dg)
        elif not acc:
            G(vname).zero_()
            G("last_layer.weight_g").zero_()
        dt = ops.l2norm_bwd(dtn, t, tinv)
        dh2 = ops.gemm_nt(dt, flat.wt(L2 + ".weight"), epilogue=ops.EPI_GELUBWD, aux=pre2)   # gradient w.r.t. the second GELU's input
        ops.gemm_tn(dt, h2, G(L2 + ".weight"), colsum=G(L2 + ".bias"), accumulate=acc, workspace=ws)
        if head.use_bn:   # ... which is the second BatchNorm's output
            (z1, m1, r1), (z2, m2, r2) = ctx.bn_saved
            B0, B1 = head._bn
            dh2 = ops.bn_bwd(dh2, z2, m2, r2, flat.f(B1 + ".weight"), G(B1 + ".weight"), G(B1 + ".bias"), accumulate=acc)
        dh1 = ops.gemm_nt(dh2, flat.wt(L1 + ".weight"), epilogue=ops.EPI_GELUBWD, aux=pre1)
        ops.gemm_tn(dh2, h1, G(L1 + ".weight"), colsum=G(L1 + ".bias"), accumulate=acc, workspace=ws)
        if head.use_bn:
            dh1 = ops.bn_bwd(dh1, z1, m1, r1, flat.f(B0 + ".weight"), G(B0 + ".weight"), G(B0 + ".bias"), accumulate=acc)
        dx = ops.gemm_nt(dh1, flat.wt(L0 + ".weight"), out_fp32=True) if ctx.x_needs_grad else None
        ops.gemm_tn(dh1, xb, G(L0 + ".weight"), colsum=G(L0 + ".bias"), accumulate=acc, workspace=ws)
        for n, p in zip(flat.names, flat.params):
            if p.requires_grad and not (n.startswith("last_layer.") and head.skip_last_layer_grad):   # (dino.py:374-376 drops both)
                p.grad = G(n)
        pending = getattr(head, "_pending_backwards", 0)
        if pending > 1:   # (BatchNorm: one head call per crop -- the gradient slab is complete after the last of their backwards)
            head._pending_backwards = pending - 1
        elif head.grad_ready_hook is not None:
            head.grad_ready_hook(flat, 0, flat.numel)
        ctx.saved = ctx.bn_saved = None   # (released with the backward, as autograd releases saved tensors)
        return (None, dx, None) + (None,) * ctx.n_params


# ==============================================================================================
# DINO method
# ==============================================================================================
class DINO(_Base):
    _BACKBONES = {"vit_channels": vit_channels}

    def __init__(self, cfg):
        super().__init__()
        cfg = self.add_and_assert_specific_cfg(cfg)
        self.cfg = cfg
        self.method_name = cfg.method
        # ---- backbone (base.py:157-187)
        self.backbone_args = cfg.backbone.kwargs
        if cfg.backbone.name not in self._BACKBONES:
            raise RuntimeError(f"chadavit_amd supports backbone 'vit_channels' only, got {cfg.backbone.name}")
        self.base_model = self._BACKBONES[cfg.backbone.name]
        self.backbone_name = cfg.backbone.name
        kwargs = dict(self.backbone_args)
        if cfg.channels_strategy == "multi_channels":
            kwargs["max_number_channels"] = cfg.data.max_img_channels
        else:
            raise RuntimeError("chadavit_amd implements channels_strategy='multi_channels' (the ChAda-ViT path) only")
        self.backbone: nn.Module = self.base_model(cfg.method, pretrained=False, **kwargs)
        self.features_dim = self.backbone.num_features
        self.num_classes = cfg.data.num_classes
        self.channels_strategy = cfg.channels_strategy
        self.mixed_channels = cfg.mixed_channels
        self.list_num_channels: List[List[int]] = []
        self.return_all_tokens = cfg.backbone.kwargs.return_all_tokens
        if not self.mixed_channels:
            raise RuntimeError("chadavit_amd implements mixed_channels=True (variable channel counts) only")
        self.classifier = nn.Linear(self.features_dim, self.num_classes)  # online probe: dead weight in DINO
        # ---- optimisation settings (base.py:236-275)
        self.max_epochs = cfg.max_epochs
        self.accumulate_grad_batches = cfg.accumulate_grad_batches
        self.optimizer = cfg.optimizer.name
        self.batch_size = cfg.optimizer.batch_size
        self.lr = cfg.optimizer.lr
        self.weight_decay = cfg.optimizer.weight_decay
        self.classifier_lr = cfg.optimizer.classifier_lr
        self.extra_optimizer_args = dict(cfg.optimizer.kwargs)
        self.exclude_bias_n_norm_wd = cfg.optimizer.exclude_bias_n_norm_wd
        self.scheduler = cfg.scheduler.name
        self.min_lr = cfg.scheduler.min_lr
        self.lr_decay_steps = cfg.scheduler.lr_decay_steps
        self.warmup_start_lr = cfg.scheduler.warmup_start_lr
        self.warmup_epochs = cfg.scheduler.warmup_epochs
        self.scheduler_interval = cfg.scheduler.interval
        assert self.scheduler_interval in ["step", "epoch"]
        if self.accumulate_grad_batches:
            self.lr = self.lr * self.accumulate_grad_batches
            self.classifier_lr = self.classifier_lr * self.accumulate_grad_batches if self.classifier_lr else None
            self.min_lr = self.min_lr * self.accumulate_grad_batches
            self.warmup_start_lr = self.warmup_start_lr * self.accumulate_grad_batches
        self.num_large_crops = cfg.data.num_large_crops
        self.num_small_crops = cfg.data.num_small_crops
        self.num_crops = self.num_large_crops + self.num_small_crops
        self.multicrop = self.num_small_crops != 0
        # ---- momentum pieces (base.py:1005-1044)
        self.momentum_backbone: nn.Module = self.base_model(cfg.method, pretrained=False, **kwargs)
        initialize_momentum_params(self.backbone, self.momentum_backbone)
        # build-side option (not a reference key): backbone.kwargs.weight_dtype = "fp8" runs the encoder's nn.Linear forwards of both
        # networks on the MX-scaled fp8 MFMA (BASELINE.json configs[4]); the factory ignores unknown kwargs as the reference's does
        wdt = omegaconf_select(cfg, "backbone.kwargs.weight_dtype", "bf16")
        if wdt not in ("bf16", "fp8"):
            raise RuntimeError(f"backbone.kwargs.weight_dtype must be 'bf16' or 'fp8', got {wdt}")
        self.backbone.weight_dtype = self.momentum_backbone.weight_dtype = wdt
        self.momentum_classifier = None
        self.momentum_updater = MomentumUpdater(cfg.momentum.base_tau, cfg.momentum.final_tau)
        # ---- DINO pieces (dino.py:133-178)
        mk = cfg.method_kwargs
        self.clip_grad = mk.clip_grad
        self.freeze_last_layer = mk.freeze_last_layer
        head_kw = dict(in_dim=self.features_dim, hidden_dim=mk.proj_hidden_dim, use_bn=mk.use_bn_in_head,
                       bottleneck_dim=mk.proj_output_dim, num_prototypes=mk.num_prototypes, norm_last_layer=mk.norm_last_layer)
        self.head = DINOHead(**head_kw)
        self.momentum_head = DINOHead(**head_kw)
        initialize_momentum_params(self.head, self.momentum_head)
        # Build-side option, OFF by default (not a reference key; SURVEY 7 / 8(d) "standard-DINO"): the multi-crop loss of the DINO paper --
        # the local crops go through the head, reach the loss as extra student views (2 teacher x (2 + n_local) student pairs) and are
        # trained through.  The reference computes the local crops' backbone features and drops them (dino.py:300-325; base.py:566-620
        # returns no "z"); parity mode reproduces exactly that.
        self.standard_multicrop_loss = bool(mk.standard_multicrop_loss) and self.multicrop
        self.dino_loss_func = DINOLoss(num_prototypes=mk.num_prototypes, student_temp=mk.student_temperature,
                                       warmup_teacher_temp=mk.warmup_teacher_temperature, teacher_temp=mk.teacher_temperature,
                                       warmup_teacher_temp_epochs=mk.warmup_teacher_temperature_epochs, num_epochs=self.max_epochs,
                                       num_large_crops=self.num_crops if self.standard_multicrop_loss else 2)
        self.last_step = 0
        # ---- validation pieces (base.py:277-296): online k-NN bank, per-batch metric list, SSL validation loss switch
        self.knn_eval = cfg.knn_eval.enabled
        self.knn_k = cfg.knn_eval.k
        if self.knn_eval:
            from ..utils.knn import WeightedKNNClassifier
            self.knn = WeightedKNNClassifier(k=self.knn_k, distance_fx=cfg.knn_eval.distance_func)
        self.validation_step_outputs: List[Dict[str, Any]] = []
        self.compute_ssl_val_loss = cfg.ssl_val_loss
        self.batch_crops = True  # pack same-size crops into one ragged batch per network
        # The reference's training_step runs the student backbone on the local crops and never reads the result (no head, no loss,
        # no gradient: SURVEY A7).  True (default) = run that pass all the same, as the reference does and as bench.py times it;
        # False = skip it -- same loss, gradients and weights, about a tenth of the step less (out["feats"] then holds the global
        # crops only).
        self.compute_unused_local_pass = True
        # teacher / local-crop passes on side HIP streams (see training_step); default as for ChAdaViT.dw_side_stream: it pays
        # where the backbone runs the GEMM chain (Base), not where the fused block kernels already fill the chip
        self.overlap_streams = getattr(self.backbone, "embed_dim", 0) >= 768
        self._streams = None
        self._local_pending = False
        self._clip_index = None

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def add_and_assert_specific_cfg(cfg):
        """Defaults of BaseMethod / BaseMomentumMethod / DINO (base.py:304-369, 1077-1096; dino.py:180-225)."""
        if not isinstance(cfg, dict) and not hasattr(cfg, "__getitem__"):
            raise RuntimeError("cfg must be a mapping with attribute access (omegaconf.DictConfig or chadavit_amd AttrDict)")
        if isinstance(cfg, dict) and not isinstance(cfg, AttrDict):
            cfg = AttrDict(cfg)
        for node in ("backbone", "optimizer", "scheduler", "momentum", "method_kwargs", "data"):
            ensure_node(cfg, node)
        cfg.backbone.kwargs = omegaconf_select(cfg, "backbone.kwargs", {})
        cfg.optimizer.exclude_bias_n_norm_wd = omegaconf_select(cfg, "optimizer.exclude_bias_n_norm_wd", False)
        cfg.optimizer.kwargs = omegaconf_select(cfg, "optimizer.kwargs", {})
        cfg.optimizer.classifier_lr = omegaconf_select(cfg, "optimizer.classifier_lr", None)
        cfg.accumulate_grad_batches = omegaconf_select(cfg, "accumulate_grad_batches", 1)
        cfg.scheduler.min_lr = omegaconf_select(cfg, "scheduler.min_lr", 0.0)
        cfg.scheduler.lr_decay_steps = omegaconf_select(cfg, "scheduler.lr_decay_steps", None)
        cfg.scheduler.warmup_start_lr = omegaconf_select(cfg, "scheduler.warmup_start_lr", 3e-5)
        cfg.scheduler.warmup_epochs = omegaconf_select(cfg, "scheduler.warmup_epochs", 10)
        cfg.scheduler.interval = omegaconf_select(cfg, "scheduler.interval", "step")
        cfg.mixed_channels = omegaconf_select(cfg, "mixed_channels", False)
        cfg.ssl_val_loss = omegaconf_select(cfg, "ssl_val_loss", False)
        ensure_node(cfg, "knn_eval")
        cfg.knn_eval.enabled = omegaconf_select(cfg, "knn_eval.enabled", False)
        cfg.knn_eval.k = omegaconf_select(cfg, "knn_eval.k", 20)
        cfg.knn_eval.distance_func = omegaconf_select(cfg, "knn_eval.distance_func", "euclidean")
        cfg.momentum.base_tau = omegaconf_select(cfg, "momentum.base_tau", 0.99)
        cfg.momentum.final_tau = omegaconf_select(cfg, "momentum.final_tau", 1.0)
        cfg.momentum.classifier = omegaconf_select(cfg, "momentum.classifier", False)
        if cfg.momentum.classifier:
            raise RuntimeError("chadavit_amd: momentum.classifier=True is outside the DINO hot path")
        assert not is_missing(cfg, "method_kwargs.proj_hidden_dim")
        assert not is_missing(cfg, "method_kwargs.proj_output_dim")
        assert not is_missing(cfg, "method_kwargs.num_prototypes")
        mk = cfg.method_kwargs
        mk.clip_grad = omegaconf_select(cfg, "method_kwargs.clip_grad", 0)
        mk.freeze_last_layer = omegaconf_select(cfg, "method_kwargs.freeze_last_layer", 1)
        mk.norm_last_layer = omegaconf_select(cfg, "method_kwargs.norm_last_layer", True)
        mk.use_bn_in_head = omegaconf_select(cfg, "method_kwargs.use_bn_in_head", False)
        mk.standard_multicrop_loss = omegaconf_select(cfg, "method_kwargs.standard_multicrop_loss", False)
        mk.student_temperature = omegaconf_select(cfg, "method_kwargs.student_temperature", 0.1)
        mk.teacher_temperature = omegaconf_select(cfg, "method_kwargs.teacher_temperature", 0.07)
        mk.warmup_teacher_temperature = omegaconf_select(cfg, "method_kwargs.warmup_teacher_temperature", 0.04)
        mk.warmup_teacher_temperature_epochs = omegaconf_select(cfg, "method_kwargs.warmup_teacher_temperature_epochs", 0)
        return cfg

    # ------------------------------------------------------------------------------------------
    @property
    def learnable_params(self) -> List[Dict[str, Any]]:
        """Param groups in the reference order: backbone, classifier, head (base.py:405-414, dino.py:227-236)."""
        return [
            {"name": "backbone", "params": self.backbone.parameters()},
            {"name": "classifier", "params": self.classifier.parameters(), "lr": self.classifier_lr, "weight_decay": 0},
            {"name": "head", "params": self.head.parameters()},
        ]

    @property
    def momentum_pairs(self) -> List[Tuple[Any, Any]]:
        return [(self.backbone, self.momentum_backbone), (self.head, self.momentum_head)]

    def configure_optimizers(self):
        """AdamW on the flat slabs + per-step warmup-cosine LR (base.py:416-492; lr_scheduler.py:76-125)."""
        from ..optim import FusedAdam, FusedAdamW, FusedLARS, FusedSGD, WarmupCosineLR, remove_bias_and_norm_from_weight_decay
        if self.optimizer not in ("adamw", "lars", "adam", "sgd"):   # base.py:67-72 _OPTIMIZERS
            raise RuntimeError(f"optimizer {self.optimizer} not in (sgd, lars, adam, adamw)")
        groups = []
        for g in self.learnable_params:
            g = dict(g)
            g["params"] = list(g["params"])
            groups.append(g)
        if self.exclude_bias_n_norm_wd:
            groups = remove_bias_and_norm_from_weight_decay(groups)
        kw = dict(self.extra_optimizer_args)
        if self.optimizer in ("adamw", "adam"):
            if "betas" in kw:
                kw["betas"] = tuple(kw["betas"])
            cls = FusedAdamW if self.optimizer == "adamw" else FusedAdam
            opt = cls(groups, lr=self.lr, weight_decay=self.weight_decay, modules=[self.backbone, self.head], **kw)
        elif self.optimizer == "sgd":
            opt = FusedSGD(groups, lr=self.lr, weight_decay=self.weight_decay, modules=[self.backbone, self.head], **kw)
        else:
            opt = FusedLARS(groups, lr=self.lr, weight_decay=self.weight_decay, modules=[self.backbone, self.head], **kw)
        if str(self.scheduler).lower() == "none":
            return opt
        if self.scheduler == "step":   # base.py:472-473
            from torch.optim.lr_scheduler import MultiStepLR
            return [opt], [MultiStepLR(opt, self.lr_decay_steps)]
        if self.scheduler != "warmup_cosine":
            raise ValueError(f"{self.scheduler} not in (warmup_cosine, cosine, step)")   # (the reference's message, base.py:474-475)
        total = self.trainer.estimated_stepping_batches
        warm = self.warmup_epochs * (total / self.max_epochs) if self.scheduler_interval == "step" else self.warmup_epochs
        steps = total if self.scheduler_interval == "step" else self.max_epochs
        sched = WarmupCosineLR(opt, warmup_epochs=warm, max_epochs=steps,
                               warmup_start_lr=self.warmup_start_lr if self.warmup_epochs > 0 else self.lr, eta_min=self.min_lr)
        return [opt], [{"scheduler": sched, "interval": self.scheduler_interval, "frequency": 1}]

    def optimizer_zero_grad(self, epoch, batch_idx, optimizer, *_):
        optimizer.zero_grad(set_to_none=True)

    # ------------------------------------------------------------------------------------------
    def forward(self, X: torch.Tensor, index: int) -> Dict[str, Any]:
        """Student backbone + online probe + head on one crop (base.py:508-564, dino.py:267-281)."""
        assert isinstance(self.backbone, ChAdaViT)
        feats = self.backbone(X, index, self.list_num_channels)
        logits = self.classifier(feats.detach())
        return {"logits": logits, "feats": feats, "z": self.head(feats)}

    def multicrop_forward(self, X: torch.Tensor, index: int) -> Dict[str, Any]:
        return {"feats": self.backbone(X, index, self.list_num_channels)}  # base.py:566-620 (no head)

    @torch.no_grad()
    def momentum_forward(self, X: torch.Tensor, index: int) -> Dict[str, Any]:
        feats = self.momentum_backbone(X, index, self.list_num_channels)
        return {"feats": feats, "z": self.momentum_head(feats)}

    def on_train_start(self):
        self.last_step = 0

    def on_train_epoch_start(self):
        self.dino_loss_func.epoch = self.current_epoch

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _head_per_crop(head: "DINOHead", feats: torch.Tensor, n_crops: int) -> torch.Tensor:
        """The head on the features of `n_crops` crops stacked row-wise.  Without BatchNorm one call; with it one call per crop, as the
        reference makes them (base.py:690-700 runs `self(x)` crop by crop): BatchNorm1d's statistics are those of one crop's rows and
        its running estimates move once per crop."""
        if not head.use_bn or n_crops == 1:
            head._pending_backwards = 0
            return head(feats)
        head._pending_backwards = n_crops if torch.is_grad_enabled() else 0
        return torch.cat([head(f) for f in feats.chunk(n_crops)])

    def training_step(self, batch: Sequence[Any], batch_idx: int) -> torch.Tensor:
        """Reference flow: base.py:668-733 (student), :1186-1248 (teacher), dino.py:300-325 (loss)."""
        X, targets, list_num_channels = batch
        self.list_num_channels = list_num_channels
        X = [X] if isinstance(X, torch.Tensor) else X
        if isinstance(list_num_channels[0], int):
            self.list_num_channels = list_num_channels = [list_num_channels]
        assert len(X) == self.num_crops
        self.head.skip_last_layer_grad = self.current_epoch < self.freeze_last_layer
        self.backbone._pending_backwards = 0   # (2 only in the standard multi-crop option below; a step that raised must not leave it set)
        nl = self.num_large_crops
        same_size = all(x.shape[-1] == X[0].shape[-1] for x in X[:nl])
        # Independent passes run on side HIP streams so their kernels fill each other's grid tails (a 1178-tile grid is
        # 2.3 waves of the 512 resident blocks): teacher fwd || student fwd; local-crop fwd (result unused) ||
        # loss + student backward.
        dev = X[0].device
        main = torch.cuda.current_stream(dev)
        use_streams = self.overlap_streams and self.batch_crops and same_size
        # parameter slabs are (re)built on the main stream, never lazily inside a side-stream pass; the student's bf16 shadows /
        # packed weights are refreshed here too, BEFORE the side streams fork: the local-crop pass reads them on its own stream
        for mod in (self.backbone, self.head, self.momentum_backbone, self.momentum_head):
            mod.flat_params()
        self.backbone.flat_params().refresh(need_transposes=torch.is_grad_enabled())
        if use_streams:
            if self._streams is None:
                self._streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            s_teacher, s_local = self._streams
        if self.batch_crops and same_size:
            xg = adjacent_view(list(X[:nl]))   # crops written back to back by the collate / augmentation: no copy
            if xg is None:
                xg = torch.cat(list(X[:nl]), dim=0)
            nch = [c for k in range(nl) for c in list_num_channels[k]]
            # ONE description of the global-crop batch, uploaded on the main stream before the fork; both passes take it (a
            # side-stream consumer orders itself behind the upload: RaggedBatch.use_on_current_stream)
            rbg = ragged_batch(nch, (xg.shape[-1] // self.backbone.token_learner.patch_size) ** 2, xg.device)
            if use_streams:
                # after the cat above (and after the previous step's optimiser / EMA, all enqueued on `main`)
                s_teacher.wait_stream(main)
                s_local.wait_stream(main)
                xg.record_stream(s_teacher)
                with torch.cuda.stream(s_teacher), torch.no_grad():
                    momentum_feats = self.momentum_backbone.forward_ragged(xg, nch, rb=rbg)
                    momentum_p = self._head_per_crop(self.momentum_head, momentum_feats, nl)
            feats = self.backbone.forward_ragged(xg, nch, rb=rbg)
            p = self._head_per_crop(self.head, feats, nl)
            feats_list = list(feats.chunk(nl))
            if use_streams:
                main.wait_stream(s_teacher)
                momentum_p.record_stream(main)
                momentum_feats.record_stream(main)
            else:
                with torch.no_grad():
                    momentum_feats = self.momentum_backbone.forward_ragged(xg, nch, rb=rbg)
                    momentum_p = self._head_per_crop(self.momentum_head, momentum_feats, nl)
        else:
            self.head._pending_backwards = nl if (self.head.use_bn and torch.is_grad_enabled()) else 0
            outs = [self(x, k) for k, x in enumerate(X[:nl])]
            p = torch.cat([o["z"] for o in outs])
            feats_list = [o["feats"] for o in outs]
            mouts = [self.momentum_forward(x, k) for k, x in enumerate(X[:nl])]
            momentum_p = torch.cat([o["z"] for o in mouts])
            momentum_feats = torch.cat([o["feats"] for o in mouts])
        if self.multicrop and self.standard_multicrop_loss:
            # standard-DINO option: the local crops are student views like the global ones -- backbone WITH gradient, head, loss
            small = list(X[nl:])
            if not (self.batch_crops and all(x.shape[-1] == small[0].shape[-1] for x in small)) or self.head.use_bn:
                raise RuntimeError("standard_multicrop_loss: local crops of one size, batch_crops=True and no BatchNorm in the head")
            xs = adjacent_view(small)
            if xs is None:
                xs = torch.cat(small, dim=0)
            nchs = [c for k in range(len(small)) for c in list_num_channels[nl + k]]
            if torch.is_grad_enabled():   # two backward passes per network this step: the gradient spans are final after the second
                self.backbone._pending_backwards = 2
                self.head._pending_backwards = 2
            feats_l = self.backbone.forward_ragged(xs, nchs)
            p = torch.cat([p, self.head(feats_l)])
            feats_list += list(feats_l.chunk(len(small)))
            self._local_pending = False
        elif self.multicrop and self.compute_unused_local_pass:
            # local crops: student backbone only, no head, no loss, no gradient reaches them
            small = list(X[nl:])
            ctx = torch.cuda.stream(s_local) if use_streams else torch.no_grad()
            with ctx, torch.no_grad():
                if self.batch_crops and all(x.shape[-1] == small[0].shape[-1] for x in small):
                    xs = adjacent_view(small)
                    if xs is None:
                        xs = torch.cat(small, dim=0)
                    nchs = [c for k in range(len(small)) for c in list_num_channels[nl + k]]
                    feats_list += list(self.backbone.forward_ragged(xs, nchs).chunk(len(small)))
                else:
                    feats_list += [self.backbone(x, nl + k, list_num_channels) for k, x in enumerate(small)]
            self._local_pending = use_streams
        # what the passes produced (base.py:1186-1248's `outs`): student CLS features per crop -- global crops, then the local crops the
        # reference computes and drops --, teacher CLS features, both heads' logits.  The step tests hold every one of them against the
        # reference (tests/golden_util.py::step_outputs_vs_golden): the loss alone barely moves when a teacher row is wrong.
        self._last_outs = {"feats": feats_list, "z": p, "momentum_z": momentum_p, "momentum_feats": momentum_feats}
        if self.knn_eval:  # online k-NN bank: CLS features of the global crops with a label (base.py:723-731)
            t_rep = targets.repeat(nl)
            mask = t_rep != -1
            self.knn.update(train_features=torch.cat(feats_list[:nl])[mask].detach(), train_targets=t_rep[mask])
        dino_loss = self.dino_loss_func(p, momentum_p)
        self.log("dino_loss_train", dino_loss, on_step=True, on_epoch=True, sync_dist=True)
        return dino_loss

    # ------------------------------------------------------------------------------------------
    def dino_clip_gradients(self, clip: float):
        """Per-PARAMETER L2 clip on the backbone (dino.py:249-261) as one launch over the flat grad slab."""
        flat = self.backbone.flat_params()
        if self._clip_index is None or self._clip_index[0] is not flat:
            names = [n for n, p in zip(flat.names, flat.params) if p.requires_grad]
            offs = torch.tensor([flat.offsets[n] for n in names], dtype=torch.int64, device=flat.device)
            sizes = torch.tensor([flat.view(flat.grad, n).numel() for n in names], dtype=torch.int64, device=flat.device)
            self._clip_index = (flat, offs, sizes)
        if any(p.grad is not None and p.grad.data_ptr() != flat.g(n).data_ptr() for n, p in zip(flat.names, flat.params)):
            raise RuntimeError("backbone gradients are not views of the flat slab")
        ops.clip_tensors(flat.grad, self._clip_index[1], self._clip_index[2], float(clip))

    def on_after_backward(self):
        if self._local_pending:  # the side-stream local-crop pass still reads the student's bf16 weights
            torch.cuda.current_stream().wait_stream(self._streams[1])
            self._local_pending = False
        if self.clip_grad:
            self.dino_clip_gradients(self.clip_grad)
        if self.current_epoch < self.freeze_last_layer:
            for p in self.head.last_layer.parameters():
                p.grad = None

    def on_train_batch_end(self, outputs, batch, batch_idx):
        """EMA of (backbone, head) pairs then cosine tau (base.py:1250-1276)."""
        if self.trainer.global_step > self.last_step:
            for mp in self.momentum_pairs:
                self.momentum_updater.update(*mp)
            self.log("tau", self.momentum_updater.cur_tau)
            self.momentum_updater.update_tau(cur_step=self.trainer.global_step,
                                             max_steps=self.trainer.estimated_stepping_batches)
        self.last_step = self.trainer.global_step

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def validation_step(self, batch: Sequence[Any], batch_idx: int, dataloader_idx: int = None,
                        update_validation_step_outputs: bool = True) -> Dict[str, Any]:
        """Reference flow for mixed-channel batches: BaseMethod.validation_step (base.py:753-870), BaseMomentumMethod's
        momentum pass on top (base.py:1278-1375), DINO's SSL validation loss (dino.py:327-365).
          ssl_val_loss off: the batch is ONE tensor of collated channel images; student forward (feats / logits / z), teacher
            forward (momentum feats / z, returned only through the k-NN / outs of the student as the reference does), metrics =
            {"batch_size": len(targets)}.
          ssl_val_loss on: the batch carries num_crops crops; student head pass on the large crops, backbone-only pass on the
            small ones, teacher pass on the large crops, `dino_loss_val` = the training loss on them -- which also moves the
            centre, exactly as calling `dino_loss_func` does in the reference (losses/dino.py:98).
        With knn_eval (and not in Lightning's sanity check) the CLS features feed the online k-NN as test samples."""
        X, targets, list_num_channels = batch
        self.list_num_channels = list_num_channels
        if isinstance(list_num_channels[0], int):
            self.list_num_channels = list_num_channels = [list_num_channels]
        sanity = bool(getattr(self.trainer, "sanity_checking", False)) if self.trainer is not None else False
        nl = self.num_large_crops
        if self.compute_ssl_val_loss:
            X = [X] if isinstance(X, torch.Tensor) else X
            assert len(X) == self.num_crops
            batch_size = len(list_num_channels[0])
            per = [self(x, k) for k, x in enumerate(X[:nl])]
            outs: Dict[str, Any] = {k: [o[k] for o in per] for k in per[0]}
            if self.multicrop:
                for k, x in enumerate(X[nl:]):  # index restarts at 0 as in the reference (base.py:803-806)
                    outs["feats"] = outs["feats"] + [self.multicrop_forward(x, k)["feats"]]
            if self.knn_eval and not sanity:
                # the reference hands the LIST of per-crop features to the k-NN here (base.py:814-818, `.detach()` on a list
                # fails there); the usable reading -- global-crop features as test samples -- is what is built
                self.knn.update(test_features=torch.cat(outs.pop("feats")[:nl]).detach(), test_targets=targets.repeat(nl).detach())
            metrics = {"batch_size": batch_size}
            outs.update(metrics)
            mom = [self.momentum_forward(x, k) for k, x in enumerate(X[:nl])]
            outs.update({"momentum_" + k: [o[k] for o in mom] for k in mom[0]})
            dino_loss = self.dino_loss_func(torch.cat(outs["z"]), torch.cat(outs["momentum_z"]))
            self.log("dino_loss_val", dino_loss, on_step=True, on_epoch=True, sync_dist=True)
            if update_validation_step_outputs:
                outs.update({"dino_loss_val": dino_loss})
                self.validation_step_outputs.append(outs)
            return outs
        if not isinstance(X, torch.Tensor):
            X = X[0]
        outs = self(X, 0)
        if self.knn_eval and not sanity:
            self.knn.update(test_features=outs.pop("feats").detach(), test_targets=targets.detach())
        metrics = {"batch_size": targets.size(0)}
        outs.update(metrics)
        self.momentum_forward(X, 0)  # executed and dropped by the reference when there is no momentum classifier
        if update_validation_step_outputs:
            self.validation_step_outputs.append(outs)
        return outs

    def on_validation_epoch_end(self):
        """base.py:1377-1436 for mixed-channel runs: the class-probe metrics do not exist there (no labels-vs-logits pass), so
        the epoch log holds the online k-NN accuracies only -- the reference computes them inside its `not mixed_channels`
        branch and therefore never reports them on this path; reporting them is the one deliberate addition."""
        log = {}
        sanity = bool(getattr(self.trainer, "sanity_checking", False)) if self.trainer is not None else False
        if self.knn_eval and not sanity:
            acc1, acc5 = self.knn.compute()
            log.update({"val_knn_acc1": acc1, "val_knn_acc5": acc5})
        if len(log) > 0:
            self.log_dict(log, sync_dist=True)
        self.validation_step_outputs.clear()

    @torch.no_grad()
    def extract_features(self, batch):
        """CLS features of the student backbone for evaluation (base.py:901-981)."""
        X, targets, list_num_channels = batch
        self.list_num_channels = list_num_channels
        X = X[0] if isinstance(X, (list, tuple)) else X
        return self.backbone(X, 0, list_num_channels if isinstance(list_num_channels[0], list) else [list_num_channels]), targets
