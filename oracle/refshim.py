"""Import the UNMODIFIED reference modules from /root/reference (build container only).

TEST INFRASTRUCTURE.  Used by tests/golden/make_golden.py to generate golden vectors and by
tests/test_oracle_vs_reference.py (skipped when /root/reference is absent, i.e. on the GPU box).
Nothing of the reference is copied: its files are loaded by path with importlib, behind stub
modules for the third-party packages this image lacks (pytorch_lightning, omegaconf, timm, cv2,
tifffile, torchmetrics) -- recipe from SURVEY.md section 8(c).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch.nn as nn

REF_ROOT = os.environ.get("CHADAVIT_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "src", "backbones", "vit", "chada_vit.py"))


class _AttrDict(dict):
    """30-line omegaconf.DictConfig stand-in: attribute access, nested, `.copy()`."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            v = _AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return _AttrDict({k: (v.copy() if isinstance(v, _AttrDict) else v) for k, v in self.items()})


_MISSING = object()


def _select(cfg, key, default=None):
    cur = cfg
    for part in key.split("."):
        if isinstance(cur, dict) and part in cur:
            cur = cur[part]
        else:
            return default
    return cur


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    if "omegaconf" not in sys.modules:
        OmegaConf = type("OmegaConf", (), {
            "select": staticmethod(_select),
            "is_missing": staticmethod(lambda cfg, key: _select(cfg, key, _MISSING) is _MISSING),
            "create": staticmethod(lambda d=None: _AttrDict(d or {})),
        })
        _stub("omegaconf", OmegaConf=OmegaConf, DictConfig=_AttrDict, ListConfig=list)
    if "pytorch_lightning" not in sys.modules:
        class LightningModule(nn.Module):
            current_epoch = 0
            trainer = None

            def log(self, *a, **k):
                pass

            def log_dict(self, *a, **k):
                pass
        _stub("pytorch_lightning", LightningModule=LightningModule)
    for name in ("cv2", "tifffile"):
        if name not in sys.modules:
            _stub(name)
    if "timm" not in sys.modules:
        _stub("timm")
        _stub("timm.models")
        _stub("timm.models.helpers", group_parameters=lambda *a, **k: None)
        _stub("timm.optim")
        _stub("timm.optim.optim_factory", _layer_map=lambda *a, **k: None)
        _stub("timm.models.vision_transformer", PatchEmbed=nn.Module, _create_vision_transformer=lambda *a, **k: None)
        _stub("timm.models.registry", register_model=lambda f: f)
    if "torchmetrics" not in sys.modules:
        _stub("torchmetrics")
        _stub("torchmetrics.metric", Metric=_MetricStub)


class _MetricStub(nn.Module):
    """Just enough of torchmetrics.Metric (absent here) for the reference's WeightedKNNClassifier to run: list states."""

    def __init__(self, dist_sync_on_step: bool = False):
        super().__init__()
        self._defaults = {}

    def add_state(self, name, default, persistent=False, **_):
        self._defaults[name] = default
        setattr(self, name, list(default) if isinstance(default, list) else default)

    def reset(self):
        for name, default in self._defaults.items():
            setattr(self, name, list(default) if isinstance(default, list) else default)


def _load(modname: str, relpath: str, is_pkg: bool = False):
    path = os.path.join(REF_ROOT, relpath)
    kw = {"submodule_search_locations": [os.path.dirname(path)]} if is_pkg else {}
    spec = importlib.util.spec_from_file_location(modname, path, **kw)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


_LOADED = None


def load():
    """Returns a namespace with the reference classes: ChAdaViT, chada_vit, vit_channels, DINO,
    DINOHead, DINOLoss, MomentumUpdater, initialize_momentum_params, one_channel_collate_fn,
    LinearWarmupCosineAnnealingLR, LARS, AttrDict."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not available():
        raise RuntimeError(f"reference not found under {REF_ROOT}")
    _install_stubs()
    for pkg in ("src", "src.utils", "src.backbones", "src.backbones.vit", "src.losses", "src.methods", "src.data"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    misc = _load("src.utils.misc", "src/utils/misc.py")
    mom = _load("src.utils.momentum", "src/utils/momentum.py")
    lars = _load("src.utils.lars", "src/utils/lars.py")
    sched = _load("src.utils.lr_scheduler", "src/utils/lr_scheduler.py")
    _load("src.utils.metrics", "src/utils/metrics.py")
    knn = _load("src.utils.knn", "src/utils/knn.py")
    _load("src.backbones.vit.vit", "src/backbones/vit/vit.py")
    cv = _load("src.backbones.vit.chada_vit", "src/backbones/vit/chada_vit.py")
    vitpkg = _load("src.backbones.vit", "src/backbones/vit/__init__.py", is_pkg=True)
    bbpkg = _load("src.backbones", "src/backbones/__init__.py", is_pkg=True)
    cs = _load("src.data.channels_strategies", "src/data/channels_strategies.py")
    loss = _load("src.losses.dino", "src/losses/dino.py")
    _load("src.methods.base", "src/methods/base.py")
    dino = _load("src.methods.dino", "src/methods/dino.py")
    ns = types.SimpleNamespace(
        ChAdaViT=cv.ChAdaViT, chada_vit=cv.chada_vit, vit_channels=vitpkg.vit_channels,
        DINO=dino.DINO, DINOHead=dino.DINOHead, DINOLoss=loss.DINOLoss,
        MomentumUpdater=mom.MomentumUpdater, initialize_momentum_params=mom.initialize_momentum_params,
        one_channel_collate_fn=cs.one_channel_collate_fn,
        LinearWarmupCosineAnnealingLR=sched.LinearWarmupCosineAnnealingLR, LARS=lars.LARS,
        AttrDict=_AttrDict, misc=misc, WeightedKNNClassifier=knn.WeightedKNNClassifier,
    )
    _LOADED = ns
    return ns


def load_linear():
    """The reference's linear / fine-tune evaluation module (src/methods/linear.py), for the golden vectors of
    chadavit_amd.methods.linear.  Third-party names it imports and this image lacks are stubbed: wandb, seaborn,
    pytorch_lightning.loggers.WandbLogger, and the torchmetrics.classification metric classes (stand-ins that return 0 -- the
    goldens hold only what the reference computes itself: logits, F.cross_entropy, accuracy_at_k, gradients); the SLURM logger
    module (lightning_fabric imports) is replaced by an empty class, it is only named in an isinstance at epoch end."""
    ns = load()
    import torch

    class _ZeroMetric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            return torch.zeros(())
    for name in ("wandb", "seaborn"):
        if name not in sys.modules:
            _stub(name)
    pl = sys.modules["pytorch_lightning"]
    if "pytorch_lightning.loggers" not in sys.modules:
        pl.loggers = _stub("pytorch_lightning.loggers", WandbLogger=type("WandbLogger", (), {}))
    tm = sys.modules["torchmetrics"]
    if "torchmetrics.classification" not in sys.modules:
        tm.classification = _stub("torchmetrics.classification", **{n: _ZeroMetric for n in (
            "MulticlassAccuracy", "MulticlassRecall", "MulticlassPrecision", "MulticlassAUROC", "MulticlassF1Score",
            "MulticlassConfusionMatrix")})
    if "src.utils.slurm_logger" not in sys.modules:
        _stub("src.utils.slurm_logger", SLURMLogger=type("SLURMLogger", (), {}))
    if "src.methods.linear" not in sys.modules:
        _load("src.methods.linear", "src/methods/linear.py")
    ns.LinearModel = sys.modules["src.methods.linear"].LinearModel
    return ns


def load_regression():
    """The reference's regression evaluation module (src/methods/regression.py); the torchmetrics regression metric objects it
    constructs (absent here) are stand-ins that return 0 -- the goldens hold logits, the MSE loss and gradients only."""
    ns = load_linear()
    import torch

    class _ZeroMetric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            return torch.zeros(())
    tm = sys.modules["torchmetrics"]
    for n in ("R2Score", "MeanSquaredError", "MeanAbsoluteError", "PearsonCorrCoef"):
        if not hasattr(tm, n):
            setattr(tm, n, _ZeroMetric)
    if "src.methods.regression" not in sys.modules:
        _load("src.methods.regression", "src/methods/regression.py")
    ns.RegressionModel = sys.modules["src.methods.regression"].RegressionModel
    return ns


def linear_cfg(embed_dim=192, return_all_tokens=False, img_channels=3, mixed_channels=False, num_classes=7, finetune=False,
               optimizer="sgd", lr=0.1, weight_decay=0.0, scheduler="none", max_epochs=10, batch_size=4):
    """Minimal cfg for the reference `LinearModel(backbone, cfg)` (linear.py:65-232 reads these keys)."""
    return _AttrDict({
        "backbone": {"name": "vit_channels",
                     "kwargs": {"embed_dim": embed_dim, "patch_size": 16, "return_all_tokens": return_all_tokens,
                                "max_number_channels": 10}},
        "data": {"dataset": "synthetic", "num_classes": num_classes, "img_channels": img_channels, "max_img_channels": 10},
        "channels_strategy": "multi_channels", "mixed_channels": mixed_channels, "max_epochs": max_epochs,
        "finetune": finetune,
        "optimizer": {"name": optimizer, "batch_size": batch_size, "lr": lr, "weight_decay": weight_decay},
        "scheduler": {"name": scheduler}, "slurm": {"enabled": False}, "wandb": {"enabled": False},
    })


def load_custom_transforms():
    """The reference's own augmentation arithmetic (src/data/custom_transforms.py), for golden vectors of CustomColorJitter.
    Its third-party imports are absent here and only provide base classes / names at import time: albumentations
    (A.ImageOnlyTransform) and torchvision.transforms are stubbed; nothing of them is executed by CustomColorJitter.apply."""
    if "albumentations" not in sys.modules:
        class ImageOnlyTransform:
            def __init__(self, always_apply=False, p=0.5):
                self.always_apply, self.p = always_apply, p
        _stub("albumentations", ImageOnlyTransform=ImageOnlyTransform)
    if "torchvision" not in sys.modules:
        _stub("torchvision")
        _stub("torchvision.transforms", RandomApply=object, InterpolationMode=object)
        _stub("torchvision.transforms.functional")
    for pkg in ("src", "src.data"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    return _load("src.data.custom_transforms", "src/data/custom_transforms.py")


def dino_cfg(embed_dim=192, num_prototypes=4096, num_large_crops=2, num_small_crops=0, max_epochs=10,
             proj_hidden_dim=2048, proj_output_dim=256, batch_size=4, lr=5e-4, weight_decay=1e-4,
             base_tau=0.9995, final_tau=1.0, warmup_teacher_temperature_epochs=3, clip_grad=0, freeze_last_layer=1,
             ssl_val_loss=False, knn_eval=False, knn_k=20, knn_distance="euclidean", use_bn_in_head=False, norm_last_layer=True):
    """Minimal cfg for the reference `DINO(cfg)` (SURVEY.md section 8(c) key list)."""
    return _AttrDict({
        "method": "dino",
        "backbone": {"name": "vit_channels",
                     "kwargs": {"embed_dim": embed_dim, "patch_size": 16, "return_all_tokens": False,
                                "max_number_channels": 10}},
        "data": {"dataset": "synthetic", "num_classes": 7, "max_img_channels": 10, "img_channels": 1,
                 "num_large_crops": num_large_crops, "num_small_crops": num_small_crops},
        "channels_strategy": "multi_channels", "mixed_channels": True, "weights_init": "random",
        "max_epochs": max_epochs,
        "optimizer": {"name": "adamw", "batch_size": batch_size, "lr": lr, "weight_decay": weight_decay,
                      "classifier_lr": 0.1, "token_learner_lr": None},
        "scheduler": {"name": "warmup_cosine"},
        "momentum": {"base_tau": base_tau, "final_tau": final_tau},
        "method_kwargs": {"proj_hidden_dim": proj_hidden_dim, "proj_output_dim": proj_output_dim,
                          "num_prototypes": num_prototypes, "clip_grad": clip_grad, "use_bn_in_head": use_bn_in_head,
                          "norm_last_layer": norm_last_layer,
                          "freeze_last_layer": freeze_last_layer,
                          "warmup_teacher_temperature_epochs": warmup_teacher_temperature_epochs},
        "ssl_val_loss": ssl_val_loss, "slurm": {"enabled": False}, "wandb": {"enabled": False},
        "knn_eval": {"enabled": knn_eval, "k": knn_k, "distance_func": knn_distance},
    })
