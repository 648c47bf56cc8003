"""Linear / fine-tune evaluation (chadavit_amd.methods.linear.LinearModel) on MI355X against the golden vectors of the
unmodified reference `LinearModel` (src/methods/linear.py; tests/golden/make_golden.py linear) and the CPU oracle.

Tolerances (bf16 operands, fp32 accumulate): logits abs <= 3e-2 + 2e-2 * |ref|, loss abs <= 2e-2, classifier-gradient cosine
>= 0.999 and norm rel <= 2e-2, backbone gradient norms rel <= 6e-2 (cosine >= 0.99 on the tensors compared element-wise);
accuracies exact unless two logits of a row are closer than the logit tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle import chada_ref as R
from oracle import procedural as P

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _cfg(D, return_all_tokens, img_channels, mixed, n_cls, finetune, lr, wd, optimizer="sgd", kwargs=None, scheduler="none"):
    from chadavit_amd.utils.misc import AttrDict
    return AttrDict({
        "backbone": {"name": "vit_channels", "kwargs": {"embed_dim": D, "patch_size": 16, "return_all_tokens": return_all_tokens,
                                                        "max_number_channels": 10}},
        "data": {"dataset": "synthetic", "num_classes": n_cls, "img_channels": img_channels, "max_img_channels": 10},
        "channels_strategy": "multi_channels", "mixed_channels": mixed, "max_epochs": 10, "finetune": finetune,
        "optimizer": {"name": optimizer, "batch_size": 4, "lr": lr, "weight_decay": wd, "kwargs": kwargs or {}},
        "scheduler": {"name": scheduler},
    })


def _model(g, dev, **over):
    from chadavit_amd.backbones import vit_channels
    from chadavit_amd.methods.linear import LinearModel
    D, nch = int(g["D"]), [int(c) for c in g["nch"]]
    rat, ft, mixed = bool(g["return_all_tokens"]), bool(g["finetune"]), bool(g["mixed"])
    bb = vit_channels("dino", patch_size=16, embed_dim=D, return_all_tokens=rat, max_number_channels=10)
    bb.load_state_dict(P.fill_state_dict(P.backbone_shapes(D), seed=1))
    cfg = _cfg(D, rat, nch[0], mixed, int(g["n_cls"]), ft, float(g["lr"]), float(g["wd"]), kwargs={"momentum": float(g["momentum"])}, **over)
    m = LinearModel(bb, cfg)
    assert m.classifier.in_features == int(g["K"])
    m.classifier.load_state_dict(P.fill_state_dict({"weight": (int(g["n_cls"]), int(g["K"])), "bias": (int(g["n_cls"]),)}, seed=21))
    return m.to(dev), nch


def _batch(nch, S, seed, dev):
    x, labels, ncl = R.collate(P.make_images(nch, [S], seed=seed))
    return (x.to(dev), labels.to(dev), ncl)


def _acc_ok(got, want, logits_ref, targets, k, tol):
    """Exact, or explained by a near-tie at the k-th place of some row."""
    if abs(got - want) < 1e-4:
        return True
    top = np.sort(logits_ref, axis=1)[:, ::-1]
    t = logits_ref[np.arange(len(targets)), targets]
    near = np.abs(t - top[:, min(k, top.shape[1]) - 1]) <= tol
    if k < top.shape[1]:
        near |= np.abs(t - top[:, k]) <= tol
    return abs(got - want) <= 100.0 * near.sum() / len(targets) + 1e-4


@pytest.mark.parametrize("name", ["linear_tiny_cls", "linear_tiny_all_tokens_finetune"])
def test_linear_model_step_vs_golden_and_oracle(name):
    from chadavit_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    m, nch = _model(g, dev)
    S, ft = int(g["S"]), bool(g["finetune"])
    batch = _batch(nch, S, 9, dev)
    tr = Trainer(max_epochs=10, steps_per_epoch=4).attach(m)
    m.train()
    met = m.shared_step(batch, 0, 0)
    assert met["batch_size"] == int(g["batch_size"]) == batch[0].shape[0]
    met["loss"].backward()
    with torch.no_grad():
        fw = m(batch[0], 0)
    logits, ref = fw["logits"].float().cpu().numpy(), g["logits"]
    assert list(fw["feats"].shape) == [int(v) for v in g["feats_shape"]]
    tol = 3e-2 + 2e-2 * np.abs(ref)
    assert (np.abs(logits - ref) <= tol).all(), np.abs(logits - ref).max()
    assert _cos(fw["feats"][:, :64].float(), torch.from_numpy(g["feats_head"])) >= 0.999
    assert abs(float(met["loss"].detach()) - float(g["loss"])) <= 2e-2
    tg = g["targets"]
    assert _acc_ok(float(met["acc1"]), float(g["acc1"][0]), ref, tg, 1, 6e-2)
    assert _acc_ok(float(met["acc5"]), float(g["acc5"][0]), ref, tg, 5, 6e-2)
    dW, db = m.classifier.weight.grad, m.classifier.bias.grad
    assert _cos(dW[:, :64], torch.from_numpy(g["dW_head"])) >= 0.999
    assert abs(float(dW.double().norm()) - float(g["dW_norm"])) <= 2e-2 * float(g["dW_norm"])
    np.testing.assert_allclose(db.cpu().numpy(), g["db"], atol=2e-3, rtol=2e-2)
    # ... and the oracle on the same inputs (every classifier-gradient element, not only the stored slice)
    bbp = P.fill_state_dict(P.backbone_shapes(int(g["D"])), seed=1)
    cl = P.fill_state_dict({"weight": (int(g["n_cls"]), int(g["K"])), "bias": (int(g["n_cls"]),)}, seed=21)
    _, _, _, _, _, og = R.linear_step(bbp, cl["weight"], cl["bias"], batch[0].cpu(), nch, batch[1].cpu(), bool(g["return_all_tokens"]),
                                      bool(g["mixed"]), ft)
    assert _cos(dW, og["classifier.weight"]) >= 0.999
    if ft:
        grads = {n: p.grad for n, p in m.backbone.named_parameters() if p.grad is not None}
        assert sorted(grads) == sorted(str(n) for n in g["bb_grad_names"])
        for n, gn in zip(g["bb_grad_names"], g["bb_grad_norms"]):
            got = float(grads[str(n)].double().norm())
            assert abs(got - float(gn)) <= 6e-2 * float(gn) + 1e-5, (str(n), got, float(gn))
            if float(gn) > 1e-3:
                assert _cos(grads[str(n)], og["backbone." + str(n)]) >= 0.99, str(n)
        assert _cos(grads["norm.weight"], torch.from_numpy(g["grad::norm.weight"])) >= 0.99
    else:
        assert all(p.grad is None for p in m.backbone.parameters())
    # the optimiser step configure_optimizers built (fused SGD with momentum and weight decay), then validation on a second batch
    tr.optimizer.step()
    np.testing.assert_allclose(m.classifier.weight[:, :64].detach().cpu().numpy(), g["post_W_head"], atol=3e-5 + 2e-2 * float(g["lr"]), rtol=1e-3)
    np.testing.assert_allclose(m.classifier.bias.detach().cpu().numpy(), g["post_b"], atol=1e-3 * max(1.0, 100 * float(g["lr"])), rtol=1e-3)
    if ft:
        np.testing.assert_allclose(m.backbone.norm.weight.detach().cpu().numpy(), g["post::norm.weight"], atol=2e-4, rtol=1e-3)
    tr.optimizer.zero_grad(set_to_none=True)
    m.eval()
    v = m.validation_step(_batch(nch, S, 10, dev), 0)
    assert v["batch_size"] == int(g["val_batch_size"])
    assert abs(float(v["val_loss"]) - float(g["val_loss"])) <= 3e-2 * max(1.0, float(g["val_loss"]))
    m.on_validation_epoch_end()
    assert m.confusion_matrix.shape == (int(g["n_cls"]),) * 2 and m.confusion_matrix.sum() == 2 * len(nch)
    assert not m.validation_step_metrics and not m.validation_step_preds
    logged = m.logged_metrics()
    assert abs(logged["val_loss"] - float(v["val_loss"])) < 1e-6 and "train_loss" not in logged   # (shared_step was called directly)


def test_linear_model_trains_a_separable_problem_and_freezes_the_backbone():
    """training_step through the Trainer (warmup-cosine schedule, fused AdamW): the loss on one batch falls, the frozen backbone
    does not move, mixup-style soft targets go through the caller's loss function."""
    from chadavit_amd.backbones import vit_channels
    from chadavit_amd.methods.linear import LinearModel
    from chadavit_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    bb = vit_channels("dino", patch_size=16, embed_dim=192, return_all_tokens=False, max_number_channels=10)
    bb.load_state_dict(P.fill_state_dict(P.backbone_shapes(192), seed=1))
    cfg = _cfg(192, False, 3, True, 5, False, 5e-2, 0.0, optimizer="adamw", scheduler="warmup_cosine")
    cfg.scheduler.warmup_epochs = 1
    m = LinearModel(bb, cfg).to(dev)
    before = {n: p.detach().clone() for n, p in m.backbone.named_parameters()}
    nch = [1, 3, 2, 4, 1, 2, 3, 1]
    x, _, ncl = R.collate(P.make_images(nch, [96], seed=3))
    batch = (x.to(dev), (torch.arange(len(nch)) % 5).to(dev), ncl)
    tr = Trainer(max_epochs=4, steps_per_epoch=8).attach(m)
    m.train()
    losses = [float(tr.train_step(batch, i).detach()) for i in range(24)]
    assert losses[-1] < 0.5 * losses[0], losses
    assert not m.backbone.training                     # linear.py:525-526
    assert all(torch.equal(p, before[n]) for n, p in m.backbone.named_parameters())
    assert m.logged_metrics()["train_acc1"] >= 60.0
    # soft targets: loss_func + mixup_func supplied by the caller (main_linear.py:120-152)
    soft = lambda out, t: torch.sum(-t * torch.log_softmax(out, dim=-1), dim=-1).mean()  # noqa: E731
    mix = lambda X, t: (X, torch.nn.functional.one_hot(t, 5).float() * 0.9 + 0.02)        # noqa: E731
    m2 = LinearModel(bb, _cfg(192, False, 3, True, 5, False, 5e-2, 0.0), loss_func=soft, mixup_func=mix).to(dev)
    m2.train()
    out = m2.shared_step(batch, 0, 0)
    assert set(out) == {"batch_size", "loss"} and out["loss"].requires_grad
    out["loss"].backward()
    assert m2.classifier.weight.grad is not None and torch.isfinite(m2.classifier.weight.grad).all()


def test_linear_model_rejects_what_the_reference_cannot_run():
    from chadavit_amd.backbones import vit_channels
    from chadavit_amd.methods.linear import LinearModel
    dev = torch.device("cuda:0")
    bb = vit_channels("dino", patch_size=16, embed_dim=192, return_all_tokens=True, max_number_channels=10)
    m = LinearModel(bb, _cfg(192, True, 2, False, 7, False, 0.1, 0.0)).to(dev)
    x, labels, ncl = R.collate(P.make_images([2, 3], [224], seed=4))
    with pytest.raises(RuntimeError):      # unequal channel counts cannot be stacked (linear.py:414-421)
        m.shared_step((x.to(dev), labels.to(dev), ncl), 0, 0)
    x, labels, ncl = R.collate(P.make_images([2, 2], [96], seed=4))
    with pytest.raises(RuntimeError):      # 36 patches per channel against a classifier built for 196 (linear.py:133-138)
        m.shared_step((x.to(dev), labels.to(dev), ncl), 0, 0)
    bad = _cfg(192, True, 2, False, 7, False, 0.1, 0.0)
    bad.channels_strategy = "one_channel"
    with pytest.raises(RuntimeError):
        LinearModel(bb, bad)


def test_regression_model_step_vs_golden_and_oracle():
    """chadavit_amd.methods.regression.RegressionModel (one `regressor` node, MSE on float targets, fine-tuning the backbone) against
    the golden of the reference's RegressionModel (src/methods/regression.py) and the oracle; then the fused SGD step and a
    validation step with R^2 / MSE / MAE / Pearson r against their definitions."""
    from chadavit_amd.backbones import vit_channels
    from chadavit_amd.methods.regression import RegressionModel
    from chadavit_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(GOLDEN, "regression_tiny_finetune.npz"), allow_pickle=False)
    D, S, nch = int(g["D"]), int(g["S"]), [int(c) for c in g["nch"]]
    bb = vit_channels("dino", patch_size=16, embed_dim=D, return_all_tokens=False, max_number_channels=10)
    bbsd = P.fill_state_dict(P.backbone_shapes(D), seed=1)
    bb.load_state_dict(bbsd)
    m = RegressionModel(bb, _cfg(D, False, nch[0], True, 1, True, float(g["lr"]), float(g["wd"]), kwargs={"momentum": float(g["momentum"])}))
    rg = P.fill_state_dict({"weight": (1, D), "bias": (1,)}, seed=23)
    m.regressor.load_state_dict(rg)
    m = m.to(dev)
    assert sorted(k for k in m.state_dict() if not k.startswith("backbone.")) == ["regressor.bias", "regressor.weight"]

    def batch_of(seed):
        x, labels, ncl = R.collate(P.make_images(nch, [S], seed=seed))
        return x.to(dev), (labels.float() * 0.37 - 1.0).to(dev), ncl
    batch = batch_of(9)
    tr = Trainer(max_epochs=10, steps_per_epoch=4).attach(m)
    m.train()
    met = m.shared_step(batch, 0, 0)
    assert met["batch_size"] == int(g["batch_size"])
    met["loss"].backward()
    assert abs(float(met["loss"].detach()) - float(g["loss"])) <= 2e-2
    with torch.no_grad():
        out = m(batch[0], 0)["logits"]
    np.testing.assert_allclose(out.float().cpu().numpy(), g["logits"], atol=3e-2, rtol=2e-2)
    assert _cos(m.regressor.weight.grad, torch.from_numpy(g["dW"])) >= 0.999
    np.testing.assert_allclose(m.regressor.bias.grad.cpu().numpy(), g["db"], atol=3e-3, rtol=2e-2)
    _, _, og = R.regression_step(bbsd, rg["weight"], rg["bias"], batch[0].cpu(), nch, batch[1].cpu(), True)
    grads = {n: p.grad for n, p in m.backbone.named_parameters() if p.grad is not None}
    assert sorted(grads) == sorted(str(n) for n in g["bb_grad_names"])
    for n, gn in zip(g["bb_grad_names"], g["bb_grad_norms"]):
        got = float(grads[str(n)].double().norm())
        assert abs(got - float(gn)) <= 6e-2 * float(gn) + 1e-5, (str(n), got, float(gn))
        if float(gn) > 1e-3:   # (0.98 for the single D-vectors -- cls_token: 192 numbers of bf16 noise against a small gradient; measured 0.986)
            assert _cos(grads[str(n)], og["backbone." + str(n)]) >= (0.98 if grads[str(n)].numel() <= D else 0.99), str(n)
    tr.optimizer.step()
    np.testing.assert_allclose(m.regressor.weight.detach().cpu().numpy(), g["post_W"], atol=1e-4, rtol=1e-3)
    np.testing.assert_allclose(m.regressor.bias.detach().cpu().numpy(), g["post_b"], atol=1e-4, rtol=1e-3)
    tr.optimizer.zero_grad(set_to_none=True)
    m.eval()
    vb = batch_of(10)
    v = m.validation_step(vb, 0)
    assert abs(float(v["val_loss"]) - float(g["val_loss"])) <= 3e-2 * max(1.0, float(g["val_loss"]))
    with torch.no_grad():
        o = m(vb[0], 0)["logits"].float().view(-1).cpu()
    t = vb[1].cpu()
    assert abs(float(v["val_mse"]) - float(((o - t) ** 2).mean())) < 1e-5 and abs(float(v["val_mae"]) - float((o - t).abs().mean())) < 1e-5
    assert abs(float(v["val_r2"]) - float(1 - ((o - t) ** 2).sum() / ((t - t.mean()) ** 2).sum())) < 1e-4
    assert abs(float(v["val_pcc"]) - float(np.corrcoef(o.numpy(), t.numpy())[0, 1])) < 1e-4
    m.on_validation_epoch_end()
    assert abs(m.logged_metrics()["val_mse"] - float(v["val_mse"])) < 1e-6 and not m.validation_step_metrics
