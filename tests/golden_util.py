"""Shared procedural state builder (identical to the one tests/golden/make_golden.py used)."""
from oracle import procedural as P


def build_sd(D, PR, seeds=(1, 2, 3, 4, 5), hidden=2048, bott=256, use_bn=False):
    sd = {}
    sd.update({"backbone." + k: v for k, v in P.fill_state_dict(P.backbone_shapes(D), seed=seeds[0]).items()})
    sd.update({"momentum_backbone." + k: v for k, v in P.fill_state_dict(P.backbone_shapes(D), seed=seeds[1]).items()})
    sd.update({"head." + k: v for k, v in P.fill_state_dict(P.head_shapes(D, hidden, bott, PR, use_bn=use_bn), seed=seeds[2]).items()})
    sd.update({"momentum_head." + k: v for k, v in P.fill_state_dict(P.head_shapes(D, hidden, bott, PR, use_bn=use_bn), seed=seeds[3]).items()})
    if use_bn:  # BatchNorm1d buffers at torch's initial values (use_bn_in_head: dino.py:66-72)
        import torch
        for h in ("head.", "momentum_head."):
            for bn in ("mlp.1.", "mlp.4."):
                sd[h + bn + "running_mean"] = torch.zeros(hidden)
                sd[h + bn + "running_var"] = torch.ones(hidden)
                sd[h + bn + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    sd.update(P.fill_state_dict({"classifier.weight": (7, D), "classifier.bias": (7,),
                                 "dino_loss_func.center": (1, PR)}, seed=seeds[4]))
    return sd


# constructor-argument cases of tests/golden/backbone_ctor_args.npz (kept in step with CTOR_CASES of tests/golden/make_golden.py)
CTOR_CASES = [
    (dict(embed_dim=192, patch_size=8, img_size=[64], depth=3, num_heads=2, max_number_channels=10), [2, 1], [64, 32], 71),
    (dict(embed_dim=192, patch_size=16, img_size=[224], depth=2, num_heads=6, max_number_channels=5), [3, 5, 1], [224], 72),
    (dict(embed_dim=128, patch_size=16, img_size=[96], depth=2, num_heads=2, max_number_channels=10), [1, 4], [96, 224], 73),
]


def ctor_case_state(kw, seed):
    from oracle import procedural as P
    return P.fill_state_dict(P.backbone_shapes(kw["embed_dim"], depth=kw["depth"], patch=kw["patch_size"], img=kw["img_size"][0],
                                               max_channels=kw["max_number_channels"]), seed=seed)


def grad_subset_index(numel, k=1024):
    """The spread of <= k flat indices whose gradient values the step goldens hold for every tensor (`gsub_vals`; same rule as
    tests/golden/make_golden.py::grad_subset_index)."""
    import numpy as np
    k = min(int(numel), k)
    return (np.arange(k, dtype=np.int64) * int(numel)) // k


def golden_grad_subsets(g):
    """{name: the reference's gradient at grad_subset_index(numel)} from a step golden."""
    import numpy as np
    offs = np.concatenate([[0], np.cumsum(g["gsub_counts"])])
    return {str(n): g["gsub_vals"][offs[i]:offs[i + 1]] for i, n in enumerate(g["grad_names"])}


def step_outputs_vs_golden(outs, g, copies=1, cos_min=0.999, rel_max=2e-2, what="HIP", logit_cos_min=None, logit_rel_max=None):
    """What a DINO step's passes produced against the reference's (`outs::*` of a step golden, recorded by forward hooks on the
    unmodified reference): the student's CLS features of EVERY crop -- the local crops included --, the teacher's CLS features, the
    student's and the teacher's logits.  `outs` = {"feats": [per crop], "momentum_feats": tensor or None, "z", "momentum_z"}.
    `copies`: the batch is that many copies of the golden's images, crop by crop (bench-scale replicas): every copy must match.
    `logit_*`: a separate bar for the two logit matrices (BatchNorm in the head over a handful of rows amplifies the features' bf16
    noise: the columns' batch deviations it divides by are a small fraction of the features themselves).
    Returns the worst (cosine, rel-L2, name) seen, for the log."""
    import numpy as np
    import torch

    def stack(t, rows_per_crop):   # (copies * rows) per crop -> [copy][golden row order]
        t = t.detach().float().cpu()
        parts, o = [], 0
        for r in rows_per_crop:
            parts.append(t[o:o + r * copies].reshape(copies, r, -1))
            o += r * copies
        assert o == t.shape[0], (o, t.shape)
        return torch.cat(parts, dim=1)

    rpc = [int(v) for v in g["outs::feats_rows_per_crop"]]
    nl = int(g["n_large"])
    zr = rpc if int(g["outs::z_shape"][0]) == sum(rpc) else rpc[:nl]   # (standard multi-crop option: every crop has student logits)
    got = {"feats": stack(torch.cat([f.detach().float().cpu() for f in outs["feats"]]), rpc),
           "z": stack(outs["z"], zr), "momentum_z": stack(outs["momentum_z"], rpc[:nl])}
    if outs.get("momentum_feats") is not None:
        got["momentum_feats"] = stack(outs["momentum_feats"], rpc[:nl])
    worst = (1.0, 0.0, None)
    feat_bar = (cos_min, rel_max)
    for key, t in got.items():
        cos_min, rel_max = feat_bar
        if "feats" not in key:
            cos_min, rel_max = (logit_cos_min or cos_min), (logit_rel_max or rel_max)
        ref = torch.from_numpy(g["outs::" + key]).double()
        assert list(t.shape[1:]) == [int(v) for v in g["outs::" + key + "_shape"]], (key, t.shape)
        rs, rq = torch.from_numpy(g["outs::" + key + "_rowsum"]), torch.from_numpy(g["outs::" + key + "_rowsq"])
        for c in range(copies):
            h = t[c].double()
            hh = h[:, :ref.shape[1]]
            cos = float((hh.flatten() @ ref.flatten()) / (hh.norm() * ref.norm() + 1e-30))
            rel = float((hh - ref).norm() / (ref.norm() + 1e-30))
            assert cos >= cos_min and rel <= rel_max, (what, key, "copy", c, "cos", cos, "rel", rel)
            # every row on its own (one wrong row among many does not move the matrix-level figures much)
            rrel = (hh - ref).norm(dim=1) / (ref.norm(dim=1) + 1e-30)
            assert float(rrel.max()) <= 4 * rel_max, (what, key, "copy", c, "row", int(rrel.argmax()), float(rrel.max()))
            # columns the golden does not hold (logits past the first 256 prototypes): fp64 row sums / sums of squares over all of them
            # (an element-wise error of relative size r moves a row's sum by ~ r * |row| with random signs, its sum of squares by 2r)
            assert bool(((h.sum(1) - rs).abs() <= 8 * rel_max * rq.sqrt()).all()), (what, key, "copy", c, "row sums")
            assert bool((((h ** 2).sum(1) - rq).abs() <= 4 * rel_max * rq).all()), (what, key, "copy", c, "row sums of squares")
            if cos < worst[0] or rel > worst[1]:
                worst = (min(cos, worst[0]), max(rel, worst[1]), key)
    return worst


def step_outputs_vs_oracle(outs, aux, copies=1, cos_min=0.999, rel_max=2e-2, logit_cos_min=None, logit_rel_max=None):
    """The same four outputs against the CPU oracle's `aux` (oracle/chada_ref.py::training_step), on ALL columns (the golden holds the
    first 256 prototypes of the logits); `aux` is pinned to the reference by tests/test_oracle_golden.py."""
    import torch
    nz = aux["student_logits"].shape[0] // aux["teacher_feats"][0].shape[0]    # student views (2, or all crops with the standard option)
    refs = {"feats": list(aux["feats"]), "momentum_feats": list(aux["teacher_feats"]),
            "z": list(aux["student_logits"].chunk(nz)), "momentum_z": list(aux["teacher_logits"].chunk(len(aux["teacher_feats"])))}
    worst = (1.0, 0.0, None)
    feat_bar = (cos_min, rel_max)
    for key, per_crop in refs.items():
        cos_min, rel_max = feat_bar
        if "feats" not in key:
            cos_min, rel_max = (logit_cos_min or cos_min), (logit_rel_max or rel_max)
        got = outs[key]
        got = torch.cat([f.detach().float().cpu() for f in got]) if isinstance(got, (list, tuple)) else got.detach().float().cpu()
        ref = torch.cat([r.unsqueeze(0).expand(copies, *r.shape).reshape(-1, r.shape[-1]) for r in per_crop]).double()
        assert got.shape == ref.shape, (key, got.shape, ref.shape)
        got = got.double()
        cos = float((got.flatten() @ ref.flatten()) / (got.norm() * ref.norm() + 1e-30))
        rel = float((got - ref).norm() / (ref.norm() + 1e-30))
        rrel = (got - ref).norm(dim=1) / (ref.norm(dim=1) + 1e-30)
        assert cos >= cos_min and rel <= rel_max, ("vs oracle", key, "cos", cos, "rel", rel)
        assert float(rrel.max()) <= 4 * rel_max, ("vs oracle", key, "row", int(rrel.argmax()), float(rrel.max()))
        if cos < worst[0] or rel > worst[1]:
            worst = (min(cos, worst[0]), max(rel, worst[1]), key)
    return worst


def grad_subsets_vs_golden(named, g, cos_min=0.99, norm_rel=6e-2):
    """Every gradient tensor of the HIP step against the REFERENCE's own values on the golden's spread of <= 1024 elements per tensor
    (cosine over the spread; tensors whose norm is negligible beside the largest are skipped, as in the oracle comparison).  Returns
    the worst (cosine, name)."""
    import numpy as np
    worst = (1.0, None)
    gmax = float(max(g["grad_norms"]))
    norms = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
    for n, ref in golden_grad_subsets(g).items():
        gh = named[n].grad
        assert gh is not None, n
        if float(norms[n]) <= 1e-3 * gmax or float(np.abs(ref).max()) == 0.0:
            continue
        got = gh.detach().flatten()[grad_subset_index(gh.numel())].double().cpu().numpy()
        c = float((got @ ref.astype(np.float64)) / (np.linalg.norm(got) * np.linalg.norm(ref.astype(np.float64)) + 1e-30))
        if c < worst[0]:
            worst = (c, n)
    assert worst[0] >= cos_min, ("gradient spread vs the reference", worst)
    return worst


def traj_channels(g):
    """step -> the channel counts of that step's images (a trajectory golden holds one mix for all steps or one per step)."""
    per_step = "nch_per_step" in g.files and int(g["nch_per_step"])
    return (lambda k: [int(c) for c in g["nch"][k]]) if per_step else (lambda k: [int(c) for c in g["nch"]])


def oracle_trajectory(g):
    """Consecutive training steps of the CPU oracle on a trajectory golden's schedule (tests/golden/make_golden.py::golden_traj): a new
    procedural batch every step (seed 7 + k), epoch = k // steps_per_epoch (teacher temperature, last layer frozen during epoch 0), AdamW as
    torch runs it (the step count is PER PARAMETER: a tensor whose gradient was None during epoch 0 starts at 1 when it thaws), EMA with the
    current tau, then the cosine tau update (base.py:1250-1276).  Returns one record per step with the golden's keys."""
    import numpy as np
    import torch
    from oracle import chada_ref as R
    D, PR = int(g["D"]), int(g["P"])
    sizes = [int(s) for s in g["sizes"]]
    nch_of = traj_channels(g)
    lr, wd, tau, max_steps, spe = float(g["lr"]), float(g["wd"]), float(g["base_tau"]), int(g["max_steps"]), int(g["steps_per_epoch"])
    names = [str(n) for n in g["param_names"]]
    sd = build_sd(D, PR)
    temps = R.teacher_temp_schedule(0.04, 0.07, 3, 10)
    state = {}   # name -> (m, v, step)
    recs = []
    for k in range(int(g["steps"])):
        epoch = k // spe
        crops, _, ncl = R.collate(P.make_images(nch_of(k), sizes, seed=7 + k))
        loss, grads, newc, aux = R.training_step(sd, crops, ncl, int(g["n_large"]), float(temps[epoch]), freeze_last_layer=epoch < 1)
        tot = sum(float(v.double().norm()) ** 2 for v in grads.values() if v is not None) ** 0.5
        for n, gr in grads.items():
            if gr is None:
                continue
            m, v, st = state.get(n, (torch.zeros_like(sd[n]), torch.zeros_like(sd[n]), 0))
            sd[n], m, v = R.adamw_step(sd[n], gr, m, v, st + 1, lr, wd)
            state[n] = (m, v, st + 1)
        sd["dino_loss_func.center"] = newc
        tau_used = tau
        for n in list(sd):
            if n.startswith(("backbone.", "head.")) and not n.endswith(("running_mean", "running_var", "num_batches_tracked")):
                tn = "momentum_" + n
                sd[tn] = tau * sd[tn] + (1 - tau) * sd[n]
        tau = R.tau_schedule(k + 1, max_steps, float(g["base_tau"]), 1.0)
        z, tz = aux["student_logits"].double(), aux["teacher_logits"].double()
        recs.append({"loss": float(loss), "center": newc[0, :256].numpy().copy(), "center_sum": float(newc.double().sum()), "tau_used": tau_used,
                     "tau_next": tau, "teacher_temp": float(temps[epoch]), "grad_norm_total": tot,
                     "z_rowsum": z.sum(1).numpy(), "z_rowsq": (z ** 2).sum(1).numpy(),
                     "momentum_z_rowsum": tz.sum(1).numpy(), "momentum_z_rowsq": (tz ** 2).sum(1).numpy(),
                     "student_sums": np.asarray([float(sd[n].double().sum()) for n in names]),
                     "teacher_sums": np.asarray([float(sd["momentum_" + n].double().sum()) for n in names]),
                     "student_sq": np.asarray([float((sd[n].double() ** 2).sum()) for n in names]),
                     "teacher_sq": np.asarray([float((sd["momentum_" + n].double() ** 2).sum()) for n in names])})
    return recs, sd
