"""The oracle (oracle/chada_ref.py) against golden vectors produced by the unmodified reference
(tests/golden/make_golden.py).  CPU only.  Tolerances: SURVEY.md section 8(c) -- abs <= 1e-5 on
CLS / tokens, <= 1e-6 on loss (relative), rel <= 1e-4 on grad norms."""
import os

import numpy as np
import pytest
import torch

from oracle import chada_ref as R
from oracle import procedural as P

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def _backbone_case(g):
    D = int(g["D"])
    nch = [int(c) for c in g["nch"]]
    sizes = [int(s) for s in g["sizes"]]
    sd = P.fill_state_dict(P.backbone_shapes(D), seed=int(g["seed_w"]))
    imgs = P.make_images(nch, sizes, seed=int(g["seed_x"]))
    crops, _, ncl = R.collate(imgs)
    if not isinstance(crops, list):
        crops = [crops]
    return sd, crops, ncl, nch


@pytest.mark.parametrize("name", ["backbone_tiny", "backbone_small", "backbone_base", "backbone_notebook12h", "backbone_tiny_sizes"])
def test_backbone_matches_reference(name):
    g = _load(name)
    sd, crops, ncl, nch = _backbone_case(g)
    nheads = int(g["nheads"])
    eps = float(g["final_eps"])
    with torch.no_grad():
        for k, x in enumerate(crops):
            tok, cu = R.tokenize_ragged(sd, x, ncl[k])
            assert [int(v) for v in (cu[1:] - cu[:-1])] == [int(v) for v in g[f"mask{k}_valid_per_img"]]
            rows = g[f"tok{k}_rows"]
            np.testing.assert_allclose(tok[rows].numpy(), g[f"tok{k}_vals"], atol=1e-5, rtol=0)
            assert abs(tok.double().sum().item() - float(g[f"tok{k}_sum"])) < 1e-2
            collect = []
            cls = R.backbone_ragged(sd, x, ncl[k], nheads=nheads, final_eps=eps, collect=collect)
            np.testing.assert_allclose(collect[1][rows].numpy(), g[f"blk0_{k}_vals"], atol=2e-5, rtol=0)
            np.testing.assert_allclose(collect[-1][rows].numpy(), g[f"blk{len(collect) - 2}_{k}_vals"], atol=2e-5, rtol=0)
            np.testing.assert_allclose(cls.numpy(), g[f"cls{k}"], atol=1e-5, rtol=0)
            allt = R.backbone_ragged(sd, x, ncl[k], nheads=nheads, final_eps=eps, return_all_tokens=True)
            assert list(allt.shape) == [int(v) for v in g[f"all{k}_shape"]]
            np.testing.assert_allclose(allt[g[f"all{k}_rows"]].numpy(), g[f"all{k}_vals"], atol=2e-5, rtol=0)


def test_padded_equals_ragged():
    g = _load("backbone_tiny")
    sd, crops, ncl, nch = _backbone_case(g)
    with torch.no_grad():
        a = R.backbone_padded(sd, crops[1], ncl[1])
        b = R.backbone_ragged(sd, crops[1], ncl[1])
    np.testing.assert_allclose(a.numpy(), b.numpy(), atol=5e-6, rtol=0)
    np.testing.assert_allclose(a.numpy(), g["cls1"], atol=1e-5, rtol=0)


@pytest.mark.parametrize("name", ["loss_p4096", "loss_p65536"])
def test_loss_matches_reference(name):
    g = _load(name)
    B, PR = int(g["B"]), int(g["P"])
    center = P.tensor((1, PR), "loss.center", 0.05, seed=11)
    s = P.tensor((2 * B, PR), "loss.student", 1.0, seed=12).requires_grad_(True)
    t = P.tensor((2 * B, PR), "loss.teacher", 1.0, seed=13)
    sched = R.teacher_temp_schedule(0.04, 0.07, 3, 10)
    np.testing.assert_allclose(sched, g["schedule"], rtol=0, atol=0)
    loss = R.dino_loss(s, t, center, float(sched[int(g["epoch"])]))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-6 * abs(float(g["loss"])) + 1e-6
    np.testing.assert_allclose(s.grad[[0, B - 1, B, 2 * B - 1], :128].numpy(), g["dstudent_rows"], atol=1e-8, rtol=1e-5)
    newc = R.center_update(center, t)
    np.testing.assert_allclose(newc[0, :128].numpy(), g["center_new"], atol=1e-7, rtol=0)


def test_schedules_match_reference():
    g = _load("schedules")
    for step, tau in zip(g["tau_steps"], g["taus"]):
        assert abs(R.tau_schedule(int(step), 100, 0.9995, 1.0) - float(tau)) < 1e-12
    lrs = [R.warmup_cosine_lr(s, float(g["base_lr"]), float(g["warmup"]), float(g["max_steps"]),
                              float(g["warmup_start_lr"]), float(g["eta_min"])) for s in range(100)]
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-9, atol=1e-12)


def _step_case(g):
    from tests.golden_util import build_sd
    D, PR = int(g["D"]), int(g["P"])
    sd = build_sd(D, PR, use_bn=bool(int(g["use_bn"])) if "use_bn" in g.files else False)
    imgs = P.make_images([int(c) for c in g["nch"]], [int(s) for s in g["sizes"]], seed=7)
    crops, _, ncl = R.collate(imgs)
    return sd, crops, ncl


def test_validation_step_matches_reference():
    """oracle validation_step vs the reference's DINO.validation_step, both cfg.ssl_val_loss settings (golden val_tiny)."""
    g = _load("val_tiny")
    sd, crops, ncl = _step_case(g)
    nl = int(g["n_large"])
    temp = float(R.teacher_temp_schedule(0.04, 0.07, 3, 10)[1])
    o = R.validation_step(sd, crops, ncl, nl, temp, True)
    assert abs(o["dino_loss_val"].item() - float(g["ssl::dino_loss_val"])) < 2e-6 * abs(float(g["ssl::dino_loss_val"]))
    np.testing.assert_allclose(torch.cat(o["z"])[:, :64].numpy(), g["ssl::z"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(torch.cat(o["momentum_z"])[:, :64].numpy(), g["ssl::momentum_z"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(torch.cat(o["feats"][:nl]).numpy(), g["ssl::feats"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(torch.cat(o["logits"]).numpy(), g["ssl::logits"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(o["center"][0, :256].numpy(), g["ssl::center"], atol=1e-6, rtol=0)
    assert len(o["feats"]) == int(g["ssl::n_feats"]) and o["batch_size"] == int(g["ssl::batch_size"])
    o = R.validation_step(sd, crops[0], [ncl[0]], nl, temp, False)
    np.testing.assert_allclose(o["z"][:, :64].numpy(), g["plain::z"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(o["feats"].numpy(), g["plain::feats"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(o["logits"].numpy(), g["plain::logits"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(o["center"][0, :256].numpy(), g["plain::center"], atol=0, rtol=0)  # untouched without the loss
    assert o["batch_size"] == int(g["plain::batch_size"])


_STEP_GOLDENS = ["step_tiny_multicrop", "step_tiny_c1_clip", "step_small_mixed", "step_base_c10", "step_tiny_bn_head",
                 "step_tiny_trained_prototype_norms", "step_tiny_trained_prototype_norms_epoch0",
                 # the standard-DINO multi-crop loss (flagged option; golden from a subclass of the reference's DINO, make_golden.py)
                 "step_tiny_standard_multicrop"]
if os.environ.get("CHADAVIT_SLOW_TESTS"):  # 26282-row Tiny step: 85 s of oracle on 8 cores (checked when the golden was made)
    _STEP_GOLDENS.append("step_tiny_fused_rows")


@pytest.mark.parametrize("name", _STEP_GOLDENS)
def test_training_step_matches_reference(name):
    g = _load(name)
    sd, crops, ncl = _step_case(g)
    loss, grads, newc, aux = R.training_step(sd, crops, ncl, int(g["n_large"]), float(g["teacher_temp"]),
                                             freeze_last_layer=int(g["epoch"]) < 1, clip_grad=float(g["clip_grad"]),
                                             norm_last_layer=bool(int(g["norm_last_layer"])) if "norm_last_layer" in g.files else True,
                                             standard_multicrop=bool(int(g["standard_multicrop"])) if "standard_multicrop" in g.files else False)
    assert abs(loss.item() - float(g["loss"])) < 2e-6 * abs(float(g["loss"]))
    none_names = set(str(n) for n in g["none_grad_names"])
    for n, gn in zip(g["grad_names"], g["grad_norms"]):
        n = str(n)
        assert grads[n] is not None, n
        # (BatchNorm right behind the head's first Linear cancels any per-column shift of the features: backbone.norm.bias gets a
        #  gradient that is zero up to rounding, ~1e-7, in the reference and here)
        assert abs(grads[n].double().norm().item() - gn) <= 2e-4 * gn + (1e-7 if "use_bn" in g.files and int(g["use_bn"]) else 1e-9), n
    for n in none_names:
        if n.startswith("classifier."):
            continue
        assert grads[n] is None, n
    for key in g.files:
        if key.startswith("grad::") and not key.endswith("]"):
            np.testing.assert_allclose(grads[key[6:]].numpy(), g[key], rtol=2e-3, atol=3e-6 if "use_bn" in g.files and int(g["use_bn"]) else 3e-7)
    np.testing.assert_allclose(newc[0, :256].numpy(), g["center_new"], atol=1e-7, rtol=0)
    # round 4: what the passes produced (features of every crop incl. the local ones, teacher features, both heads' logits) and a
    # spread of <= 1024 elements of EVERY gradient tensor, against the reference's -- this is what pins the oracle's `aux`, the
    # checker of the HIP path's per-pass outputs in tests/test_model_gpu.py
    from tests.golden_util import golden_grad_subsets, grad_subset_index, step_outputs_vs_golden
    step_outputs_vs_golden({"feats": aux["feats"], "momentum_feats": torch.cat(aux["teacher_feats"]), "z": aux["student_logits"],
                            "momentum_z": aux["teacher_logits"]}, g, cos_min=1 - 1e-7, rel_max=1e-4, what="oracle")
    bn = "use_bn" in g.files and int(g["use_bn"])
    for n, ref in golden_grad_subsets(g).items():
        got = grads[n].flatten()[grad_subset_index(grads[n].numel())].double().numpy()
        scale = float(np.abs(ref).max())
        assert float(np.abs(got - ref).max()) <= 2e-3 * scale + (3e-6 if bn else 3e-7), (n, float(np.abs(got - ref).max()), scale)
    for key in g.files:   # use_bn_in_head: the heads' BatchNorm running estimates after one update per global crop
        if key.startswith("bn::"):
            which, rest = key[4:].split(".", 1)
            np.testing.assert_allclose(aux[which + "_bn"][rest].numpy(), g[key], rtol=1e-5, atol=1e-6)
    # AdamW + EMA (base.py:1263-1273, momentum.py:63-87)
    lr, wd, tau = float(g["lr"]), float(g["wd"]), float(g["tau_used"])
    # (with BatchNorm in the head a handful of backbone gradient entries are zero up to rounding, and AdamW's first step moves a
    #  parameter by lr * sign(gradient): allow a few dozen such entries to land on the other side)
    flips = 60 * lr if "use_bn" in g.files and int(g["use_bn"]) else 0.0
    post = dict(zip([str(n) for n in g["post_names"]], g["post_sums"]))
    new_student = {}
    for n, gr in grads.items():
        p = sd[n]
        if gr is None:
            new_student[n] = p
            continue
        new_student[n], _, _ = R.adamw_step(p, gr, torch.zeros_like(p), torch.zeros_like(p), 1, lr, wd)
    for n in ["backbone.norm.weight"]:
        d = np.abs(new_student[n].numpy() - g["post::" + n])
        if flips:   # (an entry whose gradient is noise may move by lr in the other direction)
            assert (d > 1e-6).sum() <= 3 and d.max() <= 2.1 * lr, (d.max(), (d > 1e-6).sum())
        else:
            np.testing.assert_allclose(new_student[n].numpy(), g["post::" + n], atol=1e-6, rtol=0)
    for n, v in new_student.items():
        assert abs(v.double().sum().item() - post[n]) <= 1e-5 * (abs(post[n]) + v.numel() ** 0.5) + flips, n
        tn = n.replace("backbone.", "momentum_backbone.", 1) if n.startswith("backbone.") else n.replace("head.", "momentum_head.", 1)
        tv = tau * sd[tn] + (1 - tau) * v
        assert abs(tv.double().sum().item() - post[tn]) <= 1e-5 * (abs(post[tn]) + v.numel() ** 0.5) + flips, tn
    assert abs(R.tau_schedule(1, int(g["max_steps"]), float(g["base_tau"]), 1.0) - float(g["tau_next"])) < 1e-12


@pytest.mark.parametrize("name", ["traj_tiny_c1", "traj_tiny_mixed_multicrop"])
def test_five_step_trajectory_matches_reference(name):
    """Round 6: the state CARRIED between steps (golden traj_tiny_c1: five consecutive steps of the unmodified reference on BASELINE
    configs[0]'s shape -- Tiny, four one-channel images, two global crops -- a new batch every step, the epoch boundary after step 3).  The
    oracle must reproduce every step to fp32 round-off: the centre of step k enters the loss of k + 1 (losses/dino.py:103-118), the EMA
    teacher of step k the teacher pass of k + 1 (momentum.py:63-87), AdamW's moments and per-parameter step counts run on (the last layer
    thaws at epoch 1: dino.py:374-376), tau follows its cosine (base.py:1270-1273)."""
    from tests.golden_util import oracle_trajectory
    g = _load(name)   # (traj_tiny_mixed_multicrop: a different 1-10 channel mix every step, two global + two local crops)
    recs, sd = oracle_trajectory(g)
    assert len(recs) == int(g["steps"]) == 5
    for k, r in enumerate(recs):
        assert abs(r["loss"] - float(g["loss"][k])) <= 2e-6 * abs(float(g["loss"][k])), (k, r["loss"], float(g["loss"][k]))
        assert r["teacher_temp"] == float(g["teacher_temp"][k])
        np.testing.assert_allclose(r["center"], g["center"][k], atol=2e-6, rtol=0)
        assert abs(r["center_sum"] - float(g["center_sum"][k])) <= 5e-5, k
        assert abs(r["tau_used"] - float(g["tau_used"][k])) < 1e-12 and abs(r["tau_next"] - float(g["tau_next"][k])) < 1e-12, k
        assert abs(r["grad_norm_total"] - float(g["grad_norm_total"][k])) <= 1e-4 * float(g["grad_norm_total"][k]), k
        for key in ("z", "momentum_z"):   # both heads' logits, all 4096 columns: fp64 row sums and sums of squares
            np.testing.assert_allclose(r[key + "_rowsum"], g[key + "_rowsum"][k], atol=1e-3, rtol=0)
            np.testing.assert_allclose(r[key + "_rowsq"], g[key + "_rowsq"][k], rtol=1e-5)
        # parameters: AdamW moves an entry whose gradient is rounding noise by lr in either direction -- sums of the student's tensors carry a
        # few such flips (observed 4e-4), the EMA teacher 1 % of them
        np.testing.assert_allclose(r["student_sums"], g["student_sums"][k], atol=4e-3, rtol=1e-6)
        np.testing.assert_allclose(r["teacher_sums"], g["teacher_sums"][k], atol=2e-4, rtol=1e-6)
        np.testing.assert_allclose(r["student_sq"], g["student_sq"][k], rtol=5e-6)
        np.testing.assert_allclose(r["teacher_sq"], g["teacher_sq"][k], rtol=1e-6)
    # the trajectory is not a fixed point: the loss moves by more than any tolerance above between every two steps
    assert min(abs(float(a) - float(b)) for a, b in zip(g["loss"][:-1], g["loss"][1:])) > 0.03
    np.testing.assert_allclose(sd["backbone.norm.weight"].numpy(), g["post::backbone.norm.weight"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(sd["momentum_backbone.norm.weight"].numpy(), g["post::momentum_backbone.norm.weight"], atol=1e-6, rtol=0)


def test_lars_and_wd_split_match_reference():
    g = _load("lars")
    shapes = {"w0": (64, 48), "b0": (64,), "w1": (16, 64, 3), "g1": (16,), "z": (8, 8)}
    combos = {"plain": dict(), "excl": dict(exclude_bias_n_norm=True), "clip_nest": dict(clip_lr=True, nesterov=True),
              "wd0": dict(weight_decay=0.0)}
    for cname, kw in combos.items():
        args = dict(lr=0.3, momentum=0.9, weight_decay=1e-2, eta=1e-3)
        args.update(kw)
        for n, s in shapes.items():
            p = P.tensor(s, "lars." + n, 0.5, seed=61)
            if n == "z":
                p = torch.zeros_like(p)
            buf = None
            for step in range(2):
                gr = P.tensor(s, f"lars.g{step}." + n, 0.2, seed=62)
                p, buf = R.lars_step(p, gr, buf, **args)
            np.testing.assert_allclose(p.numpy(), g[f"{cname}::{n}"], rtol=2e-6, atol=1e-7, err_msg=f"{cname} {n}")
    groups = [{"name": "backbone", "params": [torch.zeros(s) for s in shapes.values()], "lr": 0.1},
              {"name": "head", "params": [torch.zeros(3)], "weight_decay": 0.5}]
    split = R.split_bias_and_norm_groups(groups)
    assert [x["name"] for x in split] == [str(n) for n in g["split_names"]]
    assert [len(x["params"]) for x in split] == [int(c) for c in g["split_counts"]]
    assert [float(x.get("weight_decay", -1)) for x in split] == [float(w) for w in g["split_wd"]]


@pytest.mark.parametrize("name", ["attnmap_tiny", "attnmap_tiny96"])
def test_last_selfattention_matches_reference(name):
    """oracle.last_selfattention vs the reference's get_last_selfattention (chada_vit.py:313-320)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    D, B, S = int(g["D"]), int(g["B"]), int(g["S"])
    sd = P.fill_state_dict(P.backbone_shapes(D), seed=int(g["seed_w"]))
    imgs = P.make_images([1] * B, [S], seed=int(g["seed_x"]))
    x = torch.stack([c[0] for c, _ in imgs])
    att = R.last_selfattention(sd, x)
    assert list(att.shape) == [int(v) for v in g["shape"]]
    np.testing.assert_allclose(att[:, :, 0, :].numpy(), g["cls_rows"], atol=2e-6)
    np.testing.assert_allclose(att[:, :, torch.from_numpy(g["rows"]), :].numpy(), g["sel_rows"], atol=2e-6)
    np.testing.assert_allclose(att.sum(-2).numpy(), g["col_sums"], atol=5e-5)
    assert abs(float((att.double() ** 2).sum()) - float(g["sq_sum"])) <= 1e-5 * float(g["sq_sum"])
    assert float((att.sum(-1) - 1).abs().max()) <= 1e-5


def _knn_data(seed, n_train, n_test, dim, n_cls, noise):
    centres = P.tensor((n_cls, dim), "knn.centres", 1.0, seed=seed)
    ytr = torch.arange(n_train) % n_cls
    yte = (torch.arange(n_test) * 3 + 1) % n_cls
    xtr = centres[ytr] + noise * P.tensor((n_train, dim), "knn.train", 1.0, seed=seed)
    xte = centres[yte] + noise * P.tensor((n_test, dim), "knn.test", 1.0, seed=seed)
    return xtr, ytr, xte, yte


KNN_CASES = [("cos_k20", 20, 0.07, "cosine"), ("cos_k200", 200, 0.07, "cosine"), ("cos_k5_T1", 5, 1.0, "cosine"),
             ("euc_k20", 20, 0.07, "euclidean")]


def test_knn_oracle_matches_reference_classifier():
    """oracle.knn_accuracy vs WeightedKNNClassifier.compute of the reference (src/utils/knn.py) on procedural features."""
    g = np.load(os.path.join(GOLDEN, "eval_knn_ckpt.npz"))
    xtr, ytr, xte, yte = _knn_data(71, 1500, 400, 64, 10, 4.5)
    for tag, k, T, fx in KNN_CASES:
        top1, top5 = R.knn_accuracy(xtr, ytr, xte, yte, k=k, T=T, distance_fx=fx)
        assert abs(top1 - float(g[f"{tag}::acc"][0])) < 1e-9 and abs(top5 - float(g[f"{tag}::acc"][1])) < 1e-9, (tag, top1, top5)


def test_custom_color_jitter_matches_reference():
    """oracle.custom_color_jitter vs the reference's CustomColorJitter.apply (src/data/custom_transforms.py:301-351)."""
    g = np.load(os.path.join(GOLDEN, "jitter.npz"))
    img = (P.tensor((int(g["H"]), int(g["W"]), int(g["C"])), "jitter.img", 0.5, seed=int(g["seed_img"])).numpy() * 0.5 + 0.5).astype(np.float32)
    for k in range(2):
        out = R.custom_color_jitter(img, g[f"shifts{k}"], g[f"gammas{k}"])
        np.testing.assert_allclose(out, g[f"out{k}"], atol=1e-6)
        assert out.min() >= 0.0 and out.max() <= 1.0


def _linear_case(g, seed_x):
    D, S = int(g["D"]), int(g["S"])
    nch = [int(c) for c in g["nch"]]
    bb = P.fill_state_dict(P.backbone_shapes(D), seed=1)
    cl = P.fill_state_dict({"weight": (int(g["n_cls"]), int(g["K"])), "bias": (int(g["n_cls"]),)}, seed=21)
    imgs = P.make_images(nch, [S], seed=seed_x)
    x, labels, ncl = R.collate(imgs)
    return bb, cl["weight"], cl["bias"], x, labels, ncl[0], nch


@pytest.mark.parametrize("name", ["linear_tiny_cls", "linear_tiny_all_tokens_finetune"])
def test_linear_eval_step_matches_reference(name):
    """The oracle's restatement of LinearModel.shared_step + backward + one SGD step + a validation step, against the reference
    (src/methods/linear.py run through oracle/refshim.load_linear by tests/golden/make_golden.py linear)."""
    g = _load(name)
    rat, ft, mixed = bool(g["return_all_tokens"]), bool(g["finetune"]), bool(g["mixed"])
    bb, W, b, x, labels, nch, _ = _linear_case(g, 9)
    assert [int(t) for t in g["targets"]] == [int(t) for t in labels]
    loss, logits, feats, acc1, acc5, grads = R.linear_step(bb, W, b, x, nch, labels, rat, mixed, ft)
    assert list(feats.shape) == [int(v) for v in g["feats_shape"]]
    assert int(g["batch_size"]) == x.shape[0]          # channel images, not images (linear.py:454)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(feats[:, :64].numpy(), g["feats_head"], atol=1e-5, rtol=1e-5)
    assert abs(float(feats.double().sum()) - float(g["feats_sum"])) <= 1e-5 * max(1.0, abs(float(g["feats_sum"]))) + 1e-2
    assert abs(float(loss) - float(g["loss"])) <= 1e-6 * 10
    assert acc1 == pytest.approx(float(g["acc1"][0])) and acc5 == pytest.approx(float(g["acc5"][0]))
    np.testing.assert_allclose(grads["classifier.weight"][:, :64].numpy(), g["dW_head"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(grads["classifier.bias"].numpy(), g["db"], atol=1e-6, rtol=1e-4)
    assert float(grads["classifier.weight"].double().norm()) == pytest.approx(float(g["dW_norm"]), rel=1e-4)
    lr, mom, wd = float(g["lr"]), float(g["momentum"]), float(g["wd"])
    W1, _ = R.sgd_step(W, grads["classifier.weight"], None, lr, mom, wd)
    b1, _ = R.sgd_step(b, grads["classifier.bias"], None, lr, mom, wd)
    np.testing.assert_allclose(W1[:, :64].numpy(), g["post_W_head"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(b1.numpy(), g["post_b"], atol=1e-6, rtol=1e-5)
    bb1 = dict(bb)
    if ft:
        for n, gn in zip(g["bb_grad_names"], g["bb_grad_norms"]):
            got = float(grads["backbone." + str(n)].double().norm())
            assert abs(got - float(gn)) <= 1e-3 * float(gn) + 1e-6, (n, got, float(gn))
        np.testing.assert_allclose(grads["backbone.norm.weight"].numpy(), g["grad::norm.weight"], atol=2e-5, rtol=1e-3)
        for k in bb:
            if "backbone." + k in grads:
                bb1[k], _ = R.sgd_step(bb[k], grads["backbone." + k], None, lr, mom, wd)
        np.testing.assert_allclose(bb1["norm.weight"].numpy(), g["post::norm.weight"], atol=1e-6, rtol=1e-5)
    # validation step on the second batch with the updated weights
    _, _, _, x2, labels2, nch2, _ = _linear_case(g, 10)
    with torch.no_grad():
        f2 = R.linear_features(bb1, x2, nch2, rat, mixed)
        lg2 = f2 @ W1.t() + b1
    vloss = float(torch.nn.functional.cross_entropy(lg2, labels2))
    assert abs(vloss - float(g["val_loss"])) <= 2e-4 * max(1.0, float(g["val_loss"]))
    a1, a5 = R.accuracy_at_k(lg2, labels2)
    assert a1 == pytest.approx(float(g["val_acc1"][0])) and a5 == pytest.approx(float(g["val_acc5"][0]))
    assert int(g["val_batch_size"]) == x2.shape[0]


def test_backbone_with_other_constructor_arguments_matches_reference():
    """patch 8 / a 64- or 96-pixel position grid / depth 2-3 / six heads / max_number_channels = 5 (channel tokens skipped by the
    reference: chada_vit.py:219, 248): the oracle's generic code against the reference run with those constructor arguments."""
    from tests.golden_util import CTOR_CASES, ctor_case_state
    g = _load("backbone_ctor_args")
    assert int(g["n_cases"]) == len(CTOR_CASES)
    for ci, (kw, nch, sizes, seed) in enumerate(CTOR_CASES):
        sd = ctor_case_state(kw, seed)
        crops, _, ncl = R.collate(P.make_images(nch, sizes, seed=seed + 100))
        crops = crops if isinstance(crops, list) else [crops]
        add_chan = kw["max_number_channels"] == 10
        for k, x in enumerate(crops):
            with torch.no_grad():
                cls = R.backbone_ragged(sd, x, ncl[k], kw["num_heads"], final_eps=1e-5, patch=kw["patch_size"], add_channel_token=add_chan)
                allt = R.backbone_ragged(sd, x, ncl[k], kw["num_heads"], final_eps=1e-5, return_all_tokens=True, patch=kw["patch_size"],
                                         add_channel_token=add_chan)
            np.testing.assert_allclose(cls.numpy(), g[f"c{ci}_cls{k}"], atol=2e-5, rtol=1e-4, err_msg=f"case {ci} crop {k}")
            assert list(allt.shape) == [int(v) for v in g[f"c{ci}_all{k}_shape"]]
            np.testing.assert_allclose(allt[g[f"c{ci}_all{k}_rows"]].numpy(), g[f"c{ci}_all{k}_vals"], atol=2e-5, rtol=1e-4)


def test_regression_step_matches_reference():
    """The oracle's restatement of RegressionModel.shared_step + backward + one SGD step + validation, against the reference
    (src/methods/regression.py through oracle/refshim.load_regression; tests/golden/make_golden.py regression)."""
    g = _load("regression_tiny_finetune")
    D, S, nch = int(g["D"]), int(g["S"]), [int(c) for c in g["nch"]]
    bb = P.fill_state_dict(P.backbone_shapes(D), seed=1)
    rg = P.fill_state_dict({"weight": (1, D), "bias": (1,)}, seed=23)

    def batch_of(seed):
        x, labels, ncl = R.collate(P.make_images(nch, [S], seed=seed))
        return x, labels.float() * 0.37 - 1.0, ncl[0]
    x, t, ncl = batch_of(9)
    np.testing.assert_allclose(t.numpy(), g["targets"], atol=0)
    loss, out, grads = R.regression_step(bb, rg["weight"], rg["bias"], x, ncl, t, bool(g["finetune"]))
    assert int(g["batch_size"]) == x.shape[0]
    np.testing.assert_allclose(out.numpy(), g["logits"], atol=2e-5, rtol=1e-5)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5
    np.testing.assert_allclose(grads["regressor.weight"].numpy(), g["dW"], atol=5e-6, rtol=1e-4)   # (fp32 noise of the features, ~1e-6)
    np.testing.assert_allclose(grads["regressor.bias"].numpy(), g["db"], atol=5e-6, rtol=1e-4)
    for n, gn in zip(g["bb_grad_names"], g["bb_grad_norms"]):
        got = float(grads["backbone." + str(n)].double().norm())
        assert abs(got - float(gn)) <= 1e-3 * float(gn) + 1e-6, (n, got, float(gn))
    lr, mom, wd = float(g["lr"]), float(g["momentum"]), float(g["wd"])
    W1, _ = R.sgd_step(rg["weight"], grads["regressor.weight"], None, lr, mom, wd)
    b1, _ = R.sgd_step(rg["bias"], grads["regressor.bias"], None, lr, mom, wd)
    np.testing.assert_allclose(W1.numpy(), g["post_W"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(b1.numpy(), g["post_b"], atol=1e-6, rtol=1e-5)
    bb1 = {k: (R.sgd_step(v, grads["backbone." + k], None, lr, mom, wd)[0] if "backbone." + k in grads else v) for k, v in bb.items()}
    x2, t2, ncl2 = batch_of(10)
    with torch.no_grad():
        o2 = R.backbone_ragged(bb1, x2, ncl2, 2) @ W1.t() + b1
    assert abs(float(torch.nn.functional.mse_loss(o2, t2.unsqueeze(1))) - float(g["val_loss"])) <= 2e-4 * max(1.0, float(g["val_loss"]))
