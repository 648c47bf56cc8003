"""Generate tests/golden/*.npz by running the UNMODIFIED reference (build container only).

    PYTHONPATH=/root/repo python tests/golden/make_golden.py

Weights and inputs come from oracle/procedural.py (RNG-free integer hash), so only *outputs* are
stored.  The reference is imported through oracle/refshim.py; nothing of it is copied.
Each npz also records the procedural seeds so the tests can regenerate the identical inputs.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from oracle import procedural as P  # noqa: E402
from oracle import refshim  # noqa: E402

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ref = refshim.load()
torch.manual_seed(0)


def f32(t):
    return t.detach().to(torch.float32).cpu().numpy()


def row_subset(n, k=24):
    """Deterministic spread of row indices incl. first/last."""
    idx = sorted(set([0, 1, n - 1] + [int(i * (n - 1) / (k - 1)) for i in range(k)]))
    return np.asarray(idx, dtype=np.int64)


# --------------------------------------------------------------------------------------
def golden_backbone(name, D, nch, sizes, seed_w, seed_x, nheads_direct=None):
    if nheads_direct is None:  # training factory: 2 heads, final eps 1e-6 (chada_vit.py:333-339)
        m = ref.vit_channels("dino", patch_size=16, embed_dim=D, return_all_tokens=False, max_number_channels=10)
    else:  # notebook-style direct construction (chada_vit.py:138-139): eps 1e-5, given heads
        m = ref.ChAdaViT(embed_dim=D, patch_size=16, num_heads=nheads_direct, return_all_tokens=True,
                         max_number_channels=10)
    sd = P.fill_state_dict(P.backbone_shapes(D), seed=seed_w)
    m.load_state_dict(sd)
    imgs = P.make_images(nch, sizes, seed=seed_x)
    crops, labels, ncl = ref.one_channel_collate_fn([(i, c, l) for i, (c, l) in enumerate(imgs)])
    if not isinstance(crops, list):
        crops, ncl = [crops], ncl
    out = {"D": D, "nch": np.asarray(nch), "sizes": np.asarray(sizes), "seed_w": seed_w, "seed_x": seed_x,
           "nheads": 2 if nheads_direct is None else nheads_direct,
           "final_eps": 1e-6 if nheads_direct is None else 1e-5}
    with torch.no_grad():
        for k, x in enumerate(crops):
            emb, mask = m.channel_aware_tokenization(x, k, ncl)
            valid = emb[~mask]  # (T, D) image-major valid tokens == ragged packing order
            rs = row_subset(valid.shape[0])
            out[f"tok{k}_rows"] = rs
            out[f"tok{k}_vals"] = f32(valid[rs])
            out[f"tok{k}_sum"] = np.float64(valid.double().sum().item())
            out[f"mask{k}_valid_per_img"] = (~mask).sum(1).numpy()
            # per-block outputs (valid tokens only)
            xx = emb
            for bi, blk in enumerate(m.blocks):
                xx = blk(xx, src_key_padding_mask=mask)
                if bi in (0, len(m.blocks) - 1):
                    vb = xx[~mask]
                    out[f"blk{bi}_{k}_vals"] = f32(vb[rs])
            m.return_all_tokens = False
            out[f"cls{k}"] = f32(m(x, k, ncl))
            m.return_all_tokens = True
            allt = m(x, k, ncl)
            ra = row_subset(allt.shape[0])
            out[f"all{k}_rows"] = ra
            out[f"all{k}_vals"] = f32(allt[ra])
            out[f"all{k}_shape"] = np.asarray(allt.shape)
            out[f"all{k}_sum"] = np.float64(allt.double().sum().item())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def grad_subset_index(numel, k=1024):
    """Deterministic spread of <= k flat indices of a tensor (first element included); tests/golden_util.py holds the same rule."""
    k = min(int(numel), k)
    return (np.arange(k, dtype=np.int64) * int(numel)) // k


# --------------------------------------------------------------------------------------
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


def golden_step(name, D, PR, nch, sizes, n_large, epoch, clip_grad=0.0, lr=5e-4, wd=1e-4, base_tau=0.9995,
                max_steps=100, use_bn=False, norm_last_layer=True, standard_multicrop=False):
    cfg = refshim.dino_cfg(embed_dim=D, num_prototypes=PR, num_large_crops=n_large,
                           num_small_crops=len(sizes) - n_large, clip_grad=clip_grad, lr=lr, weight_decay=wd,
                           base_tau=base_tau, use_bn_in_head=use_bn, norm_last_layer=norm_last_layer)
    if standard_multicrop:
        # NOT the reference's default behaviour.  The standard-DINO multi-crop loss assembled from the reference's own pieces: a subclass
        # whose multicrop_forward (base.py:566-620) also returns the head's output for the local crops -- BaseMethod.training_step then
        # appends it to outs["z"] (base.py:700-706) and DINO.training_step concatenates all of it (dino.py:311) -- and the reference's
        # DINOLoss chunking the student logits into num_crops views (losses/dino.py:82 with num_large_crops = num_crops).
        class StandardMultiCropDINO(ref.DINO):
            def multicrop_forward(self, X, index):
                out = super().multicrop_forward(X, index)
                out["z"] = self.head(out["feats"])
                return out
        model = StandardMultiCropDINO(cfg)
        model.dino_loss_func.num_large_crops = len(sizes)
    else:
        model = ref.DINO(cfg)
    sd = build_sd(D, PR, use_bn=use_bn)
    model.load_state_dict(sd)
    imgs = P.make_images(nch, sizes, seed=7)
    batch = ref.one_channel_collate_fn([(i, c, l) for i, (c, l) in enumerate(imgs)])
    model.current_epoch = epoch
    model.on_train_epoch_start()
    # AdamW exactly as BaseMethod.configure_optimizers builds it (base.py:416-441), scheduler "none"
    params = model.learnable_params
    for g in params:
        g["params"] = list(g["params"])
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=wd)
    # Round 4: what the step's passes PRODUCE, not only what the loss and the gradients make of it -- forward hooks (observers; the
    # reference is not modified) on the four networks record the student's CLS features of every crop (global crops, then the local
    # crops whose features DINO.training_step computes and drops), the teacher's CLS features and both heads' logits.
    seen = {"backbone": [], "momentum_backbone": [], "head": [], "momentum_head": []}
    hooks = [getattr(model, k).register_forward_hook(lambda m_, i_, o_, k=k: seen[k].append(o_.detach().clone())) for k in seen]
    loss = model.training_step(batch, 0)
    for h in hooks:
        h.remove()
    assert len(seen["backbone"]) == len(sizes) and len(seen["momentum_backbone"]) == len(seen["momentum_head"]) == n_large
    assert len(seen["head"]) == (len(sizes) if standard_multicrop else n_large)
    loss.backward()
    model.on_after_backward()
    names, gnorms, none_names, gsub = [], [], [], []
    for n, p in model.named_parameters():
        if not n.startswith(("backbone.", "head.", "classifier.")):
            continue
        if p.grad is None:
            none_names.append(n)
        else:
            names.append(n)
            gnorms.append(p.grad.double().norm().item())
            gsub.append(f32(p.grad.flatten()[grad_subset_index(p.numel())]))
    out = {"D": D, "P": PR, "nch": np.asarray(nch), "sizes": np.asarray(sizes), "n_large": n_large, "epoch": epoch,
           "clip_grad": clip_grad, "lr": lr, "wd": wd, "base_tau": base_tau, "max_steps": max_steps,
           "teacher_temp": float(model.dino_loss_func.teacher_temp_schedule[epoch]),
           "loss": np.float64(loss.item()), "grad_names": np.asarray(names), "grad_norms": np.asarray(gnorms),
           "none_grad_names": np.asarray(none_names),
           "center_new": f32(model.dino_loss_func.center)[0, :256],
           "center_new_sum": np.float64(model.dino_loss_func.center.double().sum().item()), "use_bn": int(use_bn),
           "norm_last_layer": int(norm_last_layer), "standard_multicrop": int(standard_multicrop)}
    # a fixed spread of <= 1024 elements of EVERY gradient tensor of the reference (grad_subset_index), in the order of grad_names
    out["gsub_vals"] = np.concatenate(gsub)
    out["gsub_counts"] = np.asarray([len(v) for v in gsub])
    for key, mods in (("z", "head"), ("momentum_z", "momentum_head"), ("feats", "backbone"), ("momentum_feats", "momentum_backbone")):
        t = torch.cat(seen[mods])                                  # crops stacked row-wise, in call order
        out["outs::" + key] = f32(t if "feats" in key else t[:, :256])   # features: all D columns; logits: the first 256 prototypes
        out["outs::" + key + "_shape"] = np.asarray(t.shape)
        out["outs::" + key + "_rowsum"] = t.double().sum(1).numpy()          # fp64 row sums / sums of squares over ALL columns
        out["outs::" + key + "_rowsq"] = (t.double() ** 2).sum(1).numpy()
    out["outs::feats_rows_per_crop"] = np.asarray([int(t.shape[0]) for t in seen["backbone"]])
    if not norm_last_layer and dict(model.named_parameters())["head.last_layer.weight_g"].grad is not None:
        out["grad::head.last_layer.weight_g"] = f32(dict(model.named_parameters())["head.last_layer.weight_g"].grad)
    if use_bn:  # the heads' BatchNorm running estimates after the step's forward passes (one update per global crop), and two BN gradients
        for n, b in model.named_buffers():
            if n.startswith(("head.", "momentum_head.")) and b.is_floating_point():
                out["bn::" + n] = f32(b)
        for n in ("head.mlp.1.weight", "head.mlp.4.bias"):
            out["grad::" + n] = f32(dict(model.named_parameters())[n].grad)
    # a few full gradient tensors (small ones) for element-wise checks
    for n in ("backbone.cls_token", "backbone.channel_token", "backbone.norm.weight", "backbone.blocks.0.norm1.weight",
              "backbone.blocks.11.norm1.bias", "backbone.blocks.5.self_attn.in_proj_bias",
              "backbone.token_learner.proj.bias", "head.mlp.6.bias" if use_bn else "head.mlp.4.bias"):
        out["grad::" + n] = f32(dict(model.named_parameters())[n].grad)
    pe = dict(model.named_parameters())["backbone.pos_embed"].grad
    out["grad::backbone.pos_embed[:8]"] = f32(pe[0, 0, :8])
    opt.step()
    for mp in model.momentum_pairs:  # base.py:1263-1266
        model.momentum_updater.update(*mp)
    tau0 = model.momentum_updater.cur_tau
    model.momentum_updater.update_tau(cur_step=1, max_steps=max_steps)  # base.py:1270-1273
    out["tau_used"] = tau0
    out["tau_next"] = model.momentum_updater.cur_tau
    post = {n: np.float64(p.double().sum().item()) for n, p in model.named_parameters()
            if n.startswith(("backbone.", "head.", "momentum_backbone.", "momentum_head."))}
    out["post_names"] = np.asarray(list(post))
    out["post_sums"] = np.asarray(list(post.values()))
    out["post::backbone.norm.weight"] = f32(dict(model.named_parameters())["backbone.norm.weight"])
    out["post::momentum_backbone.norm.weight"] = f32(dict(model.named_parameters())["momentum_backbone.norm.weight"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "loss", loss.item())


# --------------------------------------------------------------------------------------
TRAJ_PARAMS = ["backbone.cls_token", "backbone.pos_embed", "backbone.token_learner.proj.weight", "backbone.blocks.0.self_attn.in_proj_weight",
               "backbone.blocks.5.linear1.weight", "backbone.blocks.11.linear2.weight", "backbone.blocks.11.norm2.weight", "backbone.norm.weight",
               "head.mlp.0.weight", "head.mlp.4.weight", "head.last_layer.weight_v"]


def golden_traj(name, D, PR, nch, sizes, n_large, steps=5, lr=5e-4, wd=1e-4, base_tau=0.99, max_steps=20, steps_per_epoch=3):
    """Round 6: the state a training run CARRIES from step to step (every other golden is one step): the centre of step k in the loss of
    step k + 1 (losses/dino.py:103-118), the EMA teacher of step k in the teacher forward of k + 1 (momentum.py:63-87, base.py:1250-1276),
    AdamW's moments from step 2 on, tau's cosine schedule, and the epoch boundary (teacher temperature schedule, last layer frozen during
    epoch 0: dino.py:367-376).  BASELINE configs[0] shape by default: Tiny, one-channel images, two global crops.  `steps` consecutive steps
    of the unmodified reference, a NEW procedural batch every step (seed 7 + k), epoch = k // steps_per_epoch, Lightning's hook order
    (training_step, backward, on_after_backward, optimizer.step, optimizer_zero_grad, on_train_batch_end's two updates)."""
    per_step = isinstance(nch[0], (list, tuple))   # a different channel mix every step (the ragged packer's description changes from step to step)
    nch_of = (lambda k: list(nch[k])) if per_step else (lambda k: list(nch))
    cfg = refshim.dino_cfg(embed_dim=D, num_prototypes=PR, num_large_crops=n_large, num_small_crops=len(sizes) - n_large, lr=lr,
                           weight_decay=wd, base_tau=base_tau)
    model = ref.DINO(cfg)
    model.load_state_dict(build_sd(D, PR))
    params = model.learnable_params
    for g in params:
        g["params"] = list(g["params"])
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=wd)
    named = dict(model.named_parameters())
    out = {"D": D, "P": PR, "nch": np.asarray(nch), "nch_per_step": int(per_step), "sizes": np.asarray(sizes), "n_large": n_large, "steps": steps, "lr": lr, "wd": wd,
           "base_tau": base_tau, "max_steps": max_steps, "steps_per_epoch": steps_per_epoch, "param_names": np.asarray(TRAJ_PARAMS)}
    rec = {k: [] for k in ("loss", "epoch", "teacher_temp", "center", "center_sum", "tau_used", "tau_next", "momentum_z_rowsum", "momentum_z_rowsq",
                           "z_rowsum", "z_rowsq", "student_sums", "teacher_sums", "student_sq", "teacher_sq", "grad_norm_total")}
    for k in range(steps):
        epoch = k // steps_per_epoch
        model.current_epoch = epoch
        if k % steps_per_epoch == 0:
            model.on_train_epoch_start()
        imgs = P.make_images(nch_of(k), sizes, seed=7 + k)
        batch = ref.one_channel_collate_fn([(i, c, l) for i, (c, l) in enumerate(imgs)])
        seen = {"head": [], "momentum_head": []}
        hooks = [getattr(model, m).register_forward_hook(lambda m_, i_, o_, m=m: seen[m].append(o_.detach().clone())) for m in seen]
        loss = model.training_step(batch, k)
        for h in hooks:
            h.remove()
        loss.backward()
        model.on_after_backward()
        tot = 0.0
        for n, p in model.named_parameters():
            if n.startswith(("backbone.", "head.")) and p.grad is not None:
                tot += p.grad.double().norm().item() ** 2
        opt.step()
        model.optimizer_zero_grad(epoch, k, opt)
        tau_used = model.momentum_updater.cur_tau
        for mp in model.momentum_pairs:  # base.py:1263-1266
            model.momentum_updater.update(*mp)
        model.momentum_updater.update_tau(cur_step=k + 1, max_steps=max_steps)  # base.py:1270-1273 (global_step after the optimiser step)
        z, tz = torch.cat(seen["head"]).double(), torch.cat(seen["momentum_head"]).double()
        rec["loss"].append(loss.item()); rec["epoch"].append(epoch)
        rec["teacher_temp"].append(float(model.dino_loss_func.teacher_temp_schedule[epoch]))
        rec["center"].append(f32(model.dino_loss_func.center)[0, :256]); rec["center_sum"].append(model.dino_loss_func.center.double().sum().item())
        rec["tau_used"].append(tau_used); rec["tau_next"].append(model.momentum_updater.cur_tau)
        rec["momentum_z_rowsum"].append(tz.sum(1).numpy()); rec["momentum_z_rowsq"].append((tz ** 2).sum(1).numpy())
        rec["z_rowsum"].append(z.sum(1).numpy()); rec["z_rowsq"].append((z ** 2).sum(1).numpy())
        rec["student_sums"].append([named[n].double().sum().item() for n in TRAJ_PARAMS])
        rec["teacher_sums"].append([named["momentum_" + n].double().sum().item() for n in TRAJ_PARAMS])
        rec["student_sq"].append([(named[n].double() ** 2).sum().item() for n in TRAJ_PARAMS])
        rec["teacher_sq"].append([(named["momentum_" + n].double() ** 2).sum().item() for n in TRAJ_PARAMS])
        rec["grad_norm_total"].append(tot ** 0.5)
        print("traj step", k, "epoch", epoch, "loss", loss.item(), "tau", tau_used, flush=True)
    out.update({k: np.asarray(v) for k, v in rec.items()})
    out["post::backbone.norm.weight"] = f32(named["backbone.norm.weight"])
    out["post::momentum_backbone.norm.weight"] = f32(named["momentum_backbone.norm.weight"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, rec["loss"])


# --------------------------------------------------------------------------------------
def golden_loss(name, B, PR, epoch):
    lf = ref.DINOLoss(num_prototypes=PR, warmup_teacher_temp=0.04, teacher_temp=0.07, warmup_teacher_temp_epochs=3,
                      num_epochs=10)
    lf.center = P.tensor((1, PR), "loss.center", 0.05, seed=11)
    lf.epoch = epoch
    s = P.tensor((2 * B, PR), "loss.student", 1.0, seed=12).requires_grad_(True)
    t = P.tensor((2 * B, PR), "loss.teacher", 1.0, seed=13)
    loss = lf(s, t)
    loss.backward()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), B=B, P=PR, epoch=epoch,
                        temp=float(lf.teacher_temp_schedule[epoch]), loss=np.float64(loss.item()),
                        dstudent_rows=f32(s.grad[[0, B - 1, B, 2 * B - 1], :128]),
                        dstudent_norm=np.float64(s.grad.double().norm().item()),
                        center_new=f32(lf.center)[0, :128], center_new_sum=np.float64(lf.center.double().sum().item()),
                        schedule=lf.teacher_temp_schedule)
    print("wrote", name, loss.item())


def golden_schedules(name):
    mu = ref.MomentumUpdater(0.9995, 1.0)
    taus = []
    for step in range(0, 101, 5):
        mu.update_tau(step, 100)
        taus.append(mu.cur_tau)
    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([w], lr=5e-4)
    sch = ref.LinearWarmupCosineAnnealingLR(opt, warmup_epochs=10, max_epochs=100, warmup_start_lr=3e-5, eta_min=1e-6)
    lrs = []
    for _ in range(100):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), tau_steps=np.arange(0, 101, 5), taus=np.asarray(taus),
                        lrs=np.asarray(lrs), base_lr=5e-4, warmup=10, max_steps=100, warmup_start_lr=3e-5, eta_min=1e-6)
    print("wrote", name)


def golden_lars(name):
    """Two reference LARS steps (src/utils/lars.py) on a small procedural parameter set, four option combinations, plus the
    bias/norm weight-decay split of src/utils/misc.py:425-454."""
    shapes = {"w0": (64, 48), "b0": (64,), "w1": (16, 64, 3), "g1": (16,), "z": (8, 8)}
    out = {"names": np.asarray(list(shapes))}
    combos = {"plain": dict(), "excl": dict(exclude_bias_n_norm=True), "clip_nest": dict(clip_lr=True, nesterov=True),
              "wd0": dict(weight_decay=0.0)}
    for cname, kw in combos.items():
        ps = {n: torch.nn.Parameter(P.tensor(s, "lars." + n, 0.5, seed=61)) for n, s in shapes.items()}
        with torch.no_grad():
            ps["z"].zero_()  # zero-norm parameter: no scaling, no weight decay (lars.py:140)
        args = dict(lr=0.3, momentum=0.9, weight_decay=1e-2, eta=1e-3)
        args.update(kw)
        opt = ref.LARS(list(ps.values()), **args)
        for step in range(2):
            for n, p in ps.items():
                p.grad = P.tensor(shapes[n], f"lars.g{step}." + n, 0.2, seed=62)
            opt.step()
        for n, p in ps.items():
            out[f"{cname}::{n}"] = f32(p)
    groups = [{"name": "backbone", "params": [torch.nn.Parameter(torch.zeros(s)) for s in shapes.values()], "lr": 0.1},
              {"name": "head", "params": [torch.nn.Parameter(torch.zeros(3))], "weight_decay": 0.5}]
    split = ref.misc.remove_bias_and_norm_from_weight_decay(groups)
    out["split_names"] = np.asarray([g_["name"] for g_ in split])
    out["split_wd"] = np.asarray([float(g_.get("weight_decay", -1)) for g_ in split])
    out["split_counts"] = np.asarray([len(g_["params"]) for g_ in split])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def golden_attnmap(name, D, B, S, seed_w, seed_x):
    """get_last_selfattention (chada_vit.py:313-320) on B one-channel S x S images: (B, H, N, N) probabilities; stored are
    the CLS rows the consumer uses (main_attn.py:207), a spread of other rows, and checksums."""
    m = ref.vit_channels("dino", patch_size=16, embed_dim=D, return_all_tokens=False, max_number_channels=10)
    m.load_state_dict(P.fill_state_dict(P.backbone_shapes(D), seed=seed_w))
    imgs = P.make_images([1] * B, [S], seed=seed_x)
    x = torch.stack([c[0] for c, _ in imgs])  # (B, 1, S, S)
    with torch.no_grad():
        att = m.get_last_selfattention(x)
    N = att.shape[-1]
    rs = row_subset(N, 12)
    out = {"D": D, "B": B, "S": S, "seed_w": seed_w, "seed_x": seed_x, "shape": np.asarray(att.shape), "rows": rs,
           "cls_rows": f32(att[:, :, 0, :]), "sel_rows": f32(att[:, :, rs, :]),
           "row_sums_max_dev": np.float64((att.sum(-1) - 1).abs().max().item()),
           "col_sums": f32(att.sum(-2)), "sq_sum": np.float64((att.double() ** 2).sum().item())}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, tuple(att.shape))


def knn_data(seed, n_train, n_test, dim, n_cls, noise):
    """Procedural clustered features: class centres + noise (RNG-free, regenerated identically by the tests)."""
    centres = P.tensor((n_cls, dim), "knn.centres", 1.0, seed=seed)
    ytr = torch.arange(n_train) % n_cls
    yte = (torch.arange(n_test) * 3 + 1) % n_cls
    xtr = centres[ytr] + noise * P.tensor((n_train, dim), "knn.train", 1.0, seed=seed)
    xte = centres[yte] + noise * P.tensor((n_test, dim), "knn.test", 1.0, seed=seed)
    return xtr, ytr, xte, yte


def golden_eval(name):
    """(f)3: weighted k-NN accuracies from the reference's WeightedKNNClassifier (src/utils/knn.py) on procedural features,
    and the reference DINO module's state_dict layout (key -> shape) that checkpoints carry."""
    out = {}
    cases = [("cos_k20", 20, 0.07, "cosine"), ("cos_k200", 200, 0.07, "cosine"), ("cos_k5_T1", 5, 1.0, "cosine"),
             ("euc_k20", 20, 0.07, "euclidean")]
    xtr, ytr, xte, yte = knn_data(71, 1500, 400, 64, 10, 4.5)
    for tag, k, T, fx in cases:
        m = ref.WeightedKNNClassifier(k=k, T=T, distance_fx=fx)
        m.update(train_features=xtr[:700], train_targets=ytr[:700])      # two updates: the memory bank is concatenated
        m.update(train_features=xtr[700:], train_targets=ytr[700:], test_features=xte, test_targets=yte)
        top1, top5 = m.compute()
        out[f"{tag}::acc"] = np.asarray([top1, top5], dtype=np.float64)
        print(tag, top1, top5)
    model = ref.DINO(refshim.dino_cfg(embed_dim=192, num_prototypes=4096, num_large_crops=2, num_small_crops=8))
    sd = model.state_dict()
    out["sd_keys"] = np.asarray(list(sd.keys()))
    out["sd_shapes"] = np.asarray([",".join(str(int(v)) for v in t_.shape) for t_ in sd.values()])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, len(sd))


def golden_jitter(name):
    """(f)2, reference-owned part: CustomColorJitter.apply (src/data/custom_transforms.py:301-351) on a procedural 5-channel
    float image.  The class draws np.random.uniform(shifts) then np.random.uniform(gammas) (:322-325); the same seed
    reproduces the draws, which are stored with the output."""
    ct = refshim.load_custom_transforms()
    img = P.tensor((40, 56, 5), "jitter.img", 0.5, seed=81).numpy() * 0.5 + 0.5   # HWC, roughly [0, 1]
    img = img.astype(np.float32)
    out = {"H": 40, "W": 56, "C": 5, "seed_img": 81}
    for k, (smin, smax, gmin, gmax, seed) in enumerate([(-0.3, 0.3, 0.5, 1.5, 7), (-0.1, 0.6, 0.2, 2.5, 8)]):
        jit = ct.CustomColorJitter(int_min_shift=smin, int_max_shift=smax, gamma_min=gmin, gamma_max=gmax, p=1.0)
        np.random.seed(seed)
        res = jit.apply(img.copy())
        np.random.seed(seed)
        shifts = np.random.uniform(smin, smax, 5)
        gammas = np.random.uniform(gmin, gmax, 5)
        out[f"shifts{k}"], out[f"gammas{k}"], out[f"out{k}"] = shifts, gammas, res.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def golden_val(name, D, PR, nch, sizes, n_large):
    """DINO.validation_step (dino.py:327-365 over base.py:753-870, 1278-1375) on the unmodified reference, both settings of
    cfg.ssl_val_loss: student feats / logits / z, teacher z, dino_loss_val and the centre it leaves behind."""
    out = {"D": D, "P": PR, "nch": np.asarray(nch), "sizes": np.asarray(sizes), "n_large": n_large}
    imgs = P.make_images(nch, sizes, seed=7)
    batch = ref.one_channel_collate_fn([(i, c, l) for i, (c, l) in enumerate(imgs)])
    for ssl in (True, False):
        cfg = refshim.dino_cfg(embed_dim=D, num_prototypes=PR, num_large_crops=n_large, num_small_crops=len(sizes) - n_large,
                               ssl_val_loss=ssl)
        model = ref.DINO(cfg)
        model.load_state_dict(build_sd(D, PR))
        model.current_epoch = 1
        model.on_train_epoch_start()
        tag = "ssl" if ssl else "plain"
        with torch.no_grad():
            if ssl:
                outs = model.validation_step(batch, 0)
                out[f"{tag}::dino_loss_val"] = np.float64(outs["dino_loss_val"].item())
                out[f"{tag}::z"] = f32(torch.cat(outs["z"]))[:, :64]
                out[f"{tag}::momentum_z"] = f32(torch.cat(outs["momentum_z"]))[:, :64]
                out[f"{tag}::feats"] = f32(torch.cat(outs["feats"][:n_large]))
                out[f"{tag}::n_feats"] = len(outs["feats"])
                out[f"{tag}::logits"] = f32(torch.cat(outs["logits"]))
            else:
                X, targets, ncl = batch
                outs = model.validation_step((X[0], targets, [ncl[0]]), 0)
                out[f"{tag}::z"] = f32(outs["z"])[:, :64]
                out[f"{tag}::feats"] = f32(outs["feats"])
                out[f"{tag}::logits"] = f32(outs["logits"])
            out[f"{tag}::batch_size"] = int(outs["batch_size"])
            out[f"{tag}::n_outputs"] = len(model.validation_step_outputs)
            out[f"{tag}::center"] = f32(model.dino_loss_func.center)[0, :256]
            model.on_validation_epoch_end()
            assert len(model.validation_step_outputs) == 0
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["ssl::dino_loss_val"])


def golden_linear(name, D, nch, S, return_all_tokens, finetune, n_cls=7, lr=0.1, momentum=0.9, wd=1e-4):
    """Reference LinearModel (src/methods/linear.py): one training step (shared_step + backward + torch SGD as
    configure_optimizers builds it with scheduler "none") and one validation step on a second batch."""
    nsl = refshim.load_linear()
    mixed = len(set(nch)) > 1
    cfg = refshim.linear_cfg(embed_dim=D, return_all_tokens=return_all_tokens, img_channels=nch[0], mixed_channels=mixed,
                             num_classes=n_cls, finetune=finetune, lr=lr, weight_decay=wd)
    bb = ref.vit_channels("dino", patch_size=16, embed_dim=D, return_all_tokens=return_all_tokens, max_number_channels=10)
    bb.load_state_dict(P.fill_state_dict(P.backbone_shapes(D), seed=1))
    model = nsl.LinearModel(bb, cfg)
    K = model.classifier.in_features
    model.classifier.load_state_dict(P.fill_state_dict({"weight": (n_cls, K), "bias": (n_cls,)}, seed=21))
    out = {"D": D, "nch": np.asarray(nch), "S": S, "return_all_tokens": int(return_all_tokens), "finetune": int(finetune),
           "n_cls": n_cls, "K": K, "lr": lr, "momentum": momentum, "wd": wd, "mixed": int(mixed)}
    imgs = P.make_images(nch, [S], seed=9)
    batch = ref.one_channel_collate_fn([(i, c, l) for i, (c, l) in enumerate(imgs)])
    model.train()
    opt = torch.optim.SGD(model.classifier.parameters() if not finetune else
                          [{"name": "backbone", "params": model.backbone.parameters()},
                           {"name": "classifier", "params": model.classifier.parameters()}],
                          lr=lr, weight_decay=wd, momentum=momentum)
    met = model.shared_step(batch, 0, 0)
    loss = met["loss"]
    loss.backward()
    with torch.no_grad():
        fw = model(batch[0], 0)
    out.update({"loss": np.float64(loss.item()), "acc1": f32(met["acc1"]), "acc5": f32(met["acc5"]), "batch_size": int(met["batch_size"]),
                "logits": f32(fw["logits"]), "feats_shape": np.asarray(fw["feats"].shape),
                "feats_sum": np.float64(fw["feats"].double().sum().item()), "feats_head": f32(fw["feats"][:, :64]),
                "targets": batch[1].numpy(),
                "dW_norm": np.float64(model.classifier.weight.grad.double().norm().item()),
                "dW_head": f32(model.classifier.weight.grad[:, :64]), "db": f32(model.classifier.bias.grad)})
    if finetune:
        names, gn = [], []
        for n, p in model.backbone.named_parameters():
            if p.grad is not None:
                names.append(n)
                gn.append(p.grad.double().norm().item())
        out["bb_grad_names"], out["bb_grad_norms"] = np.asarray(names), np.asarray(gn)
        out["grad::norm.weight"] = f32(model.backbone.norm.weight.grad)
        out["grad::cls_token"] = f32(model.backbone.cls_token.grad)
    opt.step()
    out["post_W_head"] = f32(model.classifier.weight[:, :64])
    out["post_b"] = f32(model.classifier.bias)
    out["post_W_sum"] = np.float64(model.classifier.weight.double().sum().item())
    if finetune:
        out["post::norm.weight"] = f32(model.backbone.norm.weight)
    # validation on another batch with the updated weights
    model.eval()
    imgs2 = P.make_images(nch, [S], seed=10)
    batch2 = ref.one_channel_collate_fn([(i, c, l) for i, (c, l) in enumerate(imgs2)])
    with torch.no_grad():
        v = model.validation_step(batch2, 0)
    out.update({"val_loss": np.float64(v["val_loss"].item()), "val_acc1": f32(v["val_acc1"]), "val_acc5": f32(v["val_acc5"]),
                "val_batch_size": int(v["batch_size"])})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "loss", loss.item(), "val", v["val_loss"].item())


CTOR_CASES = [  # (constructor kwargs, channels per image, crop sides, seed)
    (dict(embed_dim=192, patch_size=8, img_size=[64], depth=3, num_heads=2, max_number_channels=10), [2, 1], [64, 32], 71),
    (dict(embed_dim=192, patch_size=16, img_size=[224], depth=2, num_heads=6, max_number_channels=5), [3, 5, 1], [224], 72),
    (dict(embed_dim=128, patch_size=16, img_size=[96], depth=2, num_heads=2, max_number_channels=10), [1, 4], [96, 224], 73),
]


def golden_ctor(name):
    """Constructor arguments other than the factory's (chada_vit.py:138-183): patch 8 on a 64-pixel grid, a 96-pixel position grid
    (interpolated UP to 224), depth 2 / 3, six heads, and max_number_channels = 5 -- which makes the reference skip the channel
    tokens altogether (forward() calls the tokenizer with its default max_channels = 10, chada_vit.py:219, 248, 274)."""
    out = {"n_cases": len(CTOR_CASES)}
    for ci, (kw, nch, sizes, seed) in enumerate(CTOR_CASES):
        m = ref.ChAdaViT(return_all_tokens=False, **kw)
        D = kw["embed_dim"]
        m.load_state_dict(P.fill_state_dict(P.backbone_shapes(D, depth=kw["depth"], patch=kw["patch_size"], img=kw["img_size"][0],
                                                              max_channels=kw["max_number_channels"]), seed=seed))
        imgs = P.make_images(nch, sizes, seed=seed + 100)
        crops, labels, ncl = ref.one_channel_collate_fn([(i, c, l) for i, (c, l) in enumerate(imgs)])
        crops = crops if isinstance(crops, list) else [crops]
        with torch.no_grad():
            for k, x in enumerate(crops):
                m.return_all_tokens = False
                out[f"c{ci}_cls{k}"] = f32(m(x, k, ncl))
                m.return_all_tokens = True
                allt = m(x, k, ncl)
                rs = row_subset(allt.shape[0])
                out[f"c{ci}_all{k}_rows"], out[f"c{ci}_all{k}_vals"] = rs, f32(allt[rs])
                out[f"c{ci}_all{k}_shape"], out[f"c{ci}_all{k}_sum"] = np.asarray(allt.shape), np.float64(allt.double().sum().item())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def golden_regression(name, D, nch, S, finetune, lr=0.01, momentum=0.9, wd=0.0):
    """Reference RegressionModel (src/methods/regression.py): shared_step + backward + torch SGD, then a validation step; float
    targets (the class index of the procedural images mapped to 0.37 * label - 1)."""
    nsl = refshim.load_regression()
    cfg = refshim.linear_cfg(embed_dim=D, return_all_tokens=False, img_channels=nch[0], mixed_channels=True, num_classes=1, finetune=finetune,
                             lr=lr, weight_decay=wd)
    bb = ref.vit_channels("dino", patch_size=16, embed_dim=D, return_all_tokens=False, max_number_channels=10)
    bb.load_state_dict(P.fill_state_dict(P.backbone_shapes(D), seed=1))
    model = nsl.RegressionModel(bb, cfg)
    model.regressor.load_state_dict(P.fill_state_dict({"weight": (1, D), "bias": (1,)}, seed=23))
    out = {"D": D, "nch": np.asarray(nch), "S": S, "finetune": int(finetune), "lr": lr, "momentum": momentum, "wd": wd}

    def batch_of(seed):
        X, labels, ncl = ref.one_channel_collate_fn([(i, c, l) for i, (c, l) in enumerate(P.make_images(nch, [S], seed=seed))])
        return X, labels.float() * 0.37 - 1.0, ncl
    batch = batch_of(9)
    model.train()
    params = model.regressor.parameters() if not finetune else [{"name": "backbone", "params": model.backbone.parameters()},
                                                                 {"name": "regressor", "params": model.regressor.parameters()}]
    opt = torch.optim.SGD(params, lr=lr, weight_decay=wd, momentum=momentum)
    met = model.shared_step(batch, 0, 0)
    met["loss"].backward()
    with torch.no_grad():
        fw = model(batch[0], 0)
    out.update({"loss": np.float64(met["loss"].item()), "batch_size": int(met["batch_size"]), "logits": f32(fw["logits"]), "targets": f32(batch[1]),
                "dW": f32(model.regressor.weight.grad), "db": f32(model.regressor.bias.grad)})
    if finetune:
        names, gn = [], []
        for n, p in model.backbone.named_parameters():
            if p.grad is not None:
                names.append(n)
                gn.append(p.grad.double().norm().item())
        out["bb_grad_names"], out["bb_grad_norms"] = np.asarray(names), np.asarray(gn)
    opt.step()
    out["post_W"], out["post_b"] = f32(model.regressor.weight), f32(model.regressor.bias)
    model.eval()
    with torch.no_grad():
        v = model.validation_step(batch_of(10), 0)
    out["val_loss"] = np.float64(v["val_loss"].item())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "loss", met["loss"].item(), "val", v["val_loss"].item())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "val":
        golden_val("val_tiny", 192, 4096, [2, 1, 4], [224, 224, 96], 2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "jitter":
        golden_jitter("jitter")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "eval":
        golden_eval("eval_knn_ckpt")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lars":
        golden_lars("lars")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "standard_multicrop":
        # round 4: the standard-DINO multi-crop loss (flagged option, NOT the reference's default): 2 global + 3 local crops, mixed channels
        golden_step("step_tiny_standard_multicrop", 192, 4096, [3, 1, 5, 2], [224, 224, 96, 96, 96], 2, 1, standard_multicrop=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "steps_all":
        # round 4: every step golden again (same arguments as below), now also holding the passes' outputs and a spread of every gradient
        golden_step("step_tiny_multicrop", 192, 4096, [3, 1, 5], [224, 224, 96, 96], 2, 1)
        golden_step("step_tiny_c1_clip", 192, 4096, [1, 1, 1, 1], [224, 224], 2, 0, clip_grad=0.3)
        golden_step("step_small_mixed", 384, 4096, [2, 7, 1], [224, 224, 96, 96], 2, 1)
        golden_step("step_base_c10", 768, 4096, [10, 3], [224, 224], 2, 1)
        golden_step("step_tiny_fused_rows", 192, 4096, [10, 10, 10, 10, 10, 8, 5, 3, 1], [224, 224, 96, 96], 2, 1)
        golden_step("step_tiny_bn_head", 192, 4096, [1, 2, 1, 3, 1, 2, 1, 1, 2, 1, 1, 1, 2, 1, 3, 1], [224, 224, 96], 2, 1, use_bn=True)
        golden_step("step_tiny_trained_prototype_norms", 192, 4096, [3, 1, 2], [224, 224, 96], 2, 1, norm_last_layer=False)
        golden_step("step_tiny_trained_prototype_norms_epoch0", 192, 4096, [3, 1, 2], [224, 224, 96], 2, 0, norm_last_layer=False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "steps_r02":
        # round 2: Small / Base training steps (the D = 384 / 768 backward kernels) and a >= 24576-token Tiny step
        # (the row count at which ChAdaViT dispatches the whole-block kernel by default)
        which = sys.argv[2:] or ["small", "base", "big"]
        if "small" in which:
            golden_step("step_small_mixed", 384, 4096, [2, 7, 1], [224, 224, 96, 96], 2, 1)
        if "base" in which:
            golden_step("step_base_c10", 768, 4096, [10, 3], [224, 224], 2, 1)
        if "big" in which:
            golden_step("step_tiny_fused_rows", 192, 4096, [10, 10, 10, 10, 10, 8, 5, 3, 1], [224, 224, 96, 96], 2, 1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "steps_r03":
        # round 3: BatchNorm in the head (method_kwargs.use_bn_in_head = True, src/methods/dino.py:59-77)
        golden_step("step_tiny_bn_head", 192, 4096, [1, 2, 1, 3, 1, 2, 1, 1, 2, 1, 1, 1, 2, 1, 3, 1], [224, 224, 96], 2, 1, use_bn=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "linear":
        # round 3: linear / fine-tune evaluation (src/methods/linear.py) on CLS features (mixed channel counts, frozen backbone)
        # and on all patch tokens flattened per image (two channels each, fine-tuning the backbone)
        golden_linear("linear_tiny_cls", 192, [3, 1, 2, 5, 1, 4], 224, False, False)
        golden_linear("linear_tiny_all_tokens_finetune", 192, [2, 2, 2, 2], 224, True, True, lr=2e-4)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "steps_r03b":
        # round 3: method_kwargs.norm_last_layer = False (the prototypes' magnitudes weight_g are trained, dino.py:83-84), at epoch 1
        # (past freeze_last_layer) and at epoch 0 (both last-layer gradients dropped, dino.py:374-376)
        golden_step("step_tiny_trained_prototype_norms", 192, 4096, [3, 1, 2], [224, 224, 96], 2, 1, norm_last_layer=False)
        golden_step("step_tiny_trained_prototype_norms_epoch0", 192, 4096, [3, 1, 2], [224, 224, 96], 2, 0, norm_last_layer=False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "regression":
        golden_regression("regression_tiny_finetune", 192, [3, 1, 2, 5, 1, 4], 224, True, lr=1e-4)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ctor":
        golden_ctor("backbone_ctor_args")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sizes":
        # round 3: image sizes other than 224 / 96 -- 112 (7 x 7 patches), 100 (6 x 6, four pixels dropped by the stride-16 conv),
        # 32 (2 x 2), 16 (ONE patch per channel) and 448 (28 x 28: 2 353 tokens for three channels); bicubic position embedding each
        golden_backbone("backbone_tiny_sizes", 192, [2, 1, 3], [112, 100, 32, 16, 448], 61, 62)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "traj":
        # round 6: five consecutive steps on BASELINE configs[0]'s shape (Tiny, four one-channel images, two global crops), the epoch
        # boundary after step 3 (teacher temperature 0.04 -> 0.055, the last layer thawed)
        which = sys.argv[2:] or ["c1", "mixed"]
        if "c1" in which:
            golden_traj("traj_tiny_c1", 192, 4096, [1, 1, 1, 1], [224, 224], 2)
        if "mixed" in which:
            # ... and with what the path is FOR: a different 1-10 channel mix every step, two global + two local crops (the ragged descriptions, the
            # bicubic position rows of the 96-pixel crops and the channel tokens' gradient slots change from step to step)
            golden_traj("traj_tiny_mixed_multicrop", 192, 4096, [[3, 1, 5], [2, 7, 1], [1, 1, 10], [4, 2, 3], [6, 1, 2]], [224, 224, 96, 96], 2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attnmap":
        golden_attnmap("attnmap_tiny", 192, 2, 224, 51, 52)
        golden_attnmap("attnmap_tiny96", 192, 3, 96, 53, 54)
        sys.exit(0)
    golden_schedules("schedules")
    golden_loss("loss_p4096", 4, 4096, 1)
    golden_loss("loss_p65536", 2, 65536, 5)
    golden_backbone("backbone_tiny", 192, [3, 1, 10, 5], [224, 96], 1, 2)
    golden_backbone("backbone_small", 384, [2, 7], [224], 21, 22)
    golden_backbone("backbone_base", 768, [4, 1], [224], 31, 32)
    golden_backbone("backbone_notebook12h", 192, [2, 3], [224], 41, 42, nheads_direct=12)
    golden_step("step_tiny_multicrop", 192, 4096, [3, 1, 5], [224, 224, 96, 96], 2, 1)
    golden_step("step_tiny_c1_clip", 192, 4096, [1, 1, 1, 1], [224, 224], 2, 0, clip_grad=0.3)
    golden_lars("lars")
    golden_attnmap("attnmap_tiny", 192, 2, 224, 51, 52)
    golden_attnmap("attnmap_tiny96", 192, 3, 96, 53, 54)
    golden_eval("eval_knn_ckpt")
    golden_jitter("jitter")
