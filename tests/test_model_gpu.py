"""End-to-end parity of the HIP path on MI355X against (a) the committed golden vectors produced by the
unmodified reference and (b) the CPU oracle on the same seeded inputs.

Tolerances (bf16 storage / fp32 accumulate):
  CLS embeddings: cosine >= 0.999 and rel-L2 <= 2e-2;  DINO loss: abs <= 2e-2;
  global grad norm: rel <= 5e-2;  per-tensor grad cosine >= 0.99 (tensors with non-negligible norm)."""
import os

import numpy as np
import pytest
import torch

from oracle import chada_ref as R
from oracle import procedural as P
from tests.golden_util import build_sd

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _cos(a, b):
    a = a.double().flatten().cpu()
    b = b.double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _rel(a, b):
    a = a.double().cpu()
    b = b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _backbone(D, seed_w, dev, return_all_tokens=False, num_heads=None):
    from chadavit_amd.backbones import vit_channels
    from chadavit_amd.backbones.vit.chada_vit import ChAdaViT
    if num_heads is None:
        m = vit_channels("dino", patch_size=16, embed_dim=D, return_all_tokens=return_all_tokens, max_number_channels=10)
    else:
        m = ChAdaViT(embed_dim=D, patch_size=16, num_heads=num_heads, return_all_tokens=return_all_tokens, max_number_channels=10)
    m.load_state_dict(P.fill_state_dict(P.backbone_shapes(D), seed=seed_w))
    return m.to(dev)


@pytest.mark.parametrize("name", ["backbone_tiny", "backbone_small", "backbone_base", "backbone_notebook12h", "backbone_tiny_sizes"])
def test_backbone_vs_golden(name):
    """backbone_notebook12h = the reference's DEFAULT constructor as HOW_TO_USE.ipynb cell 13 calls it: 12 heads (dh = 16) and a
    final LayerNorm eps of 1e-5 -- the feature-extraction path of the notebook, forward only."""
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    D = int(g["D"])
    nch = [int(c) for c in g["nch"]]
    sizes = [int(s) for s in g["sizes"]]
    nheads = int(g["nheads"])
    m = _backbone(D, int(g["seed_w"]), dev, num_heads=None if nheads == 2 else nheads)
    m.return_all_tokens = False
    imgs = P.make_images(nch, sizes, seed=int(g["seed_x"]))
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    crops, labels, ncl = one_channel_collate_fn(imgs)
    if not isinstance(crops, list):
        crops = [crops]
    with torch.no_grad():
        for k, x in enumerate(crops):
            x = x.to(dev)
            tok, cu = m.channel_aware_tokenization(x, k, ncl)
            ref_rows = torch.from_numpy(g[f"tok{k}_vals"])
            got = tok[torch.from_numpy(g[f"tok{k}_rows"]).to(dev)]
            assert _rel(got, ref_rows) < 1e-2, "tokens"
            m._capture_blocks = {0: None, len(m.blocks) - 1: None}
            cls = m(x, k, ncl)
            cap, m._capture_blocks = m._capture_blocks, None
            for bi, xb in cap.items():  # block outputs (valid tokens = the packed rows) vs the reference's blocks, rows `tok_rows`
                ref_b = torch.from_numpy(g[f"blk{bi}_{k}_vals"])
                got_b = xb[torch.from_numpy(g[f"tok{k}_rows"]).to(dev)]
                assert _cos(got_b, ref_b) >= 0.999 and _rel(got_b, ref_b) <= 2.5e-2, (name, k, bi, _cos(got_b, ref_b), _rel(got_b, ref_b))
            ref = torch.from_numpy(g[f"cls{k}"])
            assert cls.shape == ref.shape
            assert _cos(cls, ref) >= 0.999, (name, k, _cos(cls, ref))
            assert _rel(cls, ref) <= 2e-2, (name, k, _rel(cls, ref))
            m.return_all_tokens = True
            allt = m(x, k, ncl)
            m.return_all_tokens = False
            assert list(allt.shape) == [int(v) for v in g[f"all{k}_shape"]]
            got = allt[torch.from_numpy(g[f"all{k}_rows"]).to(dev)]
            assert _rel(got, torch.from_numpy(g[f"all{k}_vals"])) <= 2e-2


@pytest.mark.parametrize("name", ["loss_p4096", "loss_p65536"])
def test_dino_loss_module_vs_golden(name):
    """DINOLoss (HIP kernel behind the reference's module surface, losses/dino.py:69-118) on the reference's own golden case:
    loss, dL/dstudent rows + norm, centre update.  fp32 kernel, bf16-stored gradient: loss rel 1e-5, gradient rows rel 1e-2
    (bf16 ulp), gradient norm rel 2e-3, centre abs 1e-6.  P = 65536 is the linear-eval yaml's head size (stress row)."""
    from chadavit_amd.losses.dino import DINOLoss
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    B, PR, epoch = int(g["B"]), int(g["P"]), int(g["epoch"])
    lf = DINOLoss(num_prototypes=PR, warmup_teacher_temp=0.04, teacher_temp=0.07, warmup_teacher_temp_epochs=3, num_epochs=10).to(dev)
    np.testing.assert_allclose(lf.teacher_temp_schedule, g["schedule"], rtol=0, atol=0)
    lf.center.copy_(P.tensor((1, PR), "loss.center", 0.05, seed=11))
    lf.epoch = epoch
    s = P.tensor((2 * B, PR), "loss.student", 1.0, seed=12).to(dev).requires_grad_(True)
    t = P.tensor((2 * B, PR), "loss.teacher", 1.0, seed=13).to(dev)
    loss = lf(s, t)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"])), (loss.item(), float(g["loss"]))
    rows = s.grad[[0, B - 1, B, 2 * B - 1], :128].cpu().numpy()
    ref = g["dstudent_rows"]
    np.testing.assert_allclose(rows, ref, rtol=1e-2, atol=1e-2 * np.abs(ref).max())
    assert abs(s.grad.double().norm().item() - float(g["dstudent_norm"])) <= 2e-3 * float(g["dstudent_norm"])
    np.testing.assert_allclose(lf.center[0, :128].cpu().numpy(), g["center_new"], atol=1e-6, rtol=0)
    assert abs(lf.center.double().sum().item() - float(g["center_new_sum"])) <= 1e-4 * (1 + abs(float(g["center_new_sum"])))


@pytest.mark.parametrize("fp8_dx", [False, True])
def test_fp8_weight_path_vs_golden(fp8_dx):
    """BASELINE.json configs[4]: ChAda-ViT-Base with the encoder's nn.Linear forwards on the MX-scaled fp8 MFMA (weights and their
    input activations in OCP-MX e4m3, fp32 accumulate; attention, LayerNorm, residuals and the whole backward in bf16 / fp32).
    Tolerance of SURVEY 8(c) for this path: CLS cosine >= 0.99 vs the reference's fp32 output; DINO loss abs <= 5e-2."""
    from chadavit_amd import ops
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, "backbone_base.npz"))
    D = int(g["D"])
    m = _backbone(D, int(g["seed_w"]), dev)
    m.weight_dtype = "fp8"
    crops, labels, ncl = one_channel_collate_fn(P.make_images([int(c) for c in g["nch"]], [int(s) for s in g["sizes"]], seed=int(g["seed_x"])))
    crops = crops if isinstance(crops, list) else [crops]
    with torch.no_grad(), ops.LaunchProfiler() as prof:
        cls = m(crops[0].to(dev), 0, ncl if isinstance(ncl[0], list) else [ncl])
    keys = prof.summary()
    assert sum(v["launches"] for k, v in keys.items() if k[0] == "gemm_nt_mx8") == 4 * 12   # the fp8 kernels really ran
    ref = torch.from_numpy(g["cls0"])
    assert _cos(cls, ref) >= 0.99, _cos(cls, ref)
    print("fp8 backbone_base: CLS cosine", _cos(cls, ref), "rel-L2", _rel(cls, ref))
    # the margin above that bar is the ELEMENT FORMAT's, not slack in a kernel (profiles/r05b_fp8_error_budget.md): every block adds the same
    # independent ~3.2 % (e4m3: 3 mantissa bits, both operands quantised), the squares add up to rel-L2 ~0.11 over 12 blocks.  The budget
    # model is held too: with the 24 attention projections back on bf16 operands (fp8_keep_bf16) half of the error variance must be gone --
    # a kernel that adds error beyond the format's breaks this line long before the 0.99 above
    if not fp8_dx:
        m.fp8_keep_bf16 = ("in_proj", "out_proj")
        with torch.no_grad():
            cls_half = m(crops[0].to(dev), 0, ncl if isinstance(ncl[0], list) else [ncl])
        m.fp8_keep_bf16 = ()
        assert _cos(cls_half, ref) >= 0.996 and _rel(cls_half, ref) <= 0.09, (_cos(cls_half, ref), _rel(cls_half, ref))
        assert _rel(cls_half, ref) <= 0.8 * _rel(cls, ref), (_rel(cls_half, ref), _rel(cls, ref))
    # ---- a whole training step (student fwd + bwd, teacher fwd, loss) against the reference's Base step golden
    g = np.load(os.path.join(GOLDEN, "step_base_c10.npz"))
    D, PR = int(g["D"]), int(g["P"])
    cfg = _cfg(D, PR, int(g["n_large"]), 0)
    cfg.backbone.kwargs.weight_dtype = "fp8"
    model = DINO(cfg)
    model.load_state_dict(build_sd(D, PR))
    model = model.to(dev)
    assert model.backbone.weight_dtype == model.momentum_backbone.weight_dtype == "fp8"
    model.backbone.fp8_dx = fp8_dx   # True: the FFN's two dX GEMMs on the MX-scaled MFMA too (same gradient bar)
    crops, labels, ncl = one_channel_collate_fn(P.make_images([int(c) for c in g["nch"]], [int(s) for s in g["sizes"]], seed=7))
    tr = Trainer(max_epochs=10, steps_per_epoch=10)
    tr.current_epoch = int(g["epoch"])
    tr.attach(model)
    model.current_epoch = int(g["epoch"])
    model.on_train_epoch_start()
    with ops.LaunchProfiler() as prof:
        loss = model.training_step(([c.to(dev) for c in crops], labels.to(dev), ncl), 1)
        loss.backward()
    model.on_after_backward()
    bwd_fp8 = sum(v["launches"] for k, v in prof.summary().items() if k[0] == "gemm_nt_mx8" and k[4] in (ops.EPI_RELUMASK,))
    assert bwd_fp8 == (11 if fp8_dx else 0), bwd_fp8    # (the last block's backward runs on the CLS rows, in bf16)
    assert abs(loss.item() - float(g["loss"])) <= 5e-2, (loss.item(), float(g["loss"]))
    named = dict(model.named_parameters())
    tot_h = sum(named[str(n)].grad.double().norm().item() ** 2 for n in g["grad_names"]) ** 0.5
    tot_r = float(np.sqrt((g["grad_norms"] ** 2).sum()))
    print("fp8 step_base_c10: loss", loss.item(), "vs", float(g["loss"]), "grad norm", tot_h, "vs", tot_r)
    # fp8 gradient bar (SURVEY 8(c) states none for this path; stated here): fp8 forward activations feed a bf16 backward --
    #   global gradient norm rel <= 5e-2 (measured 2.7e-2); EVERY tensor's norm rel <= 0.10 (tensors carrying >= 1e-3 of the largest
    #   norm; measured worst 6.7e-2, blocks.7.norm2.weight); cosine >= 0.93 on every tensor the golden holds in full (measured lowest
    #   0.941: cls_token, a single D-vector; the matrices are >= 0.98)
    _fp8_gradient_bar(named, g, "fp8 step_base_c10" + (" + fp8 dX" if fp8_dx else ""))


def _fp8_gradient_bar(named, g, tag, norm_rel_global=5e-2, norm_rel_tensor=0.10, cos_min=0.93):
    names = [str(n) for n in g["grad_names"]]
    ref_norms = {n: float(v) for n, v in zip(names, g["grad_norms"])}
    tot_h = sum(named[n].grad.double().norm().item() ** 2 for n in names) ** 0.5
    tot_r = float(np.sqrt(sum(v * v for v in ref_norms.values())))
    big = max(ref_norms.values())
    worst_n, worst_c = (0.0, None), (1.0, None)
    for n in names:
        if ref_norms[n] >= 1e-3 * big:
            rel = abs(named[n].grad.double().norm().item() - ref_norms[n]) / ref_norms[n]
            worst_n = max(worst_n, (rel, n))
        if "grad::" + n in g.files:
            c = _cos(named[n].grad, torch.from_numpy(g["grad::" + n]))
            worst_c = min(worst_c, (c, n))
    print(f"{tag}: grad norm {tot_h:.5f} vs {tot_r:.5f}; worst per-tensor norm rel {worst_n}; lowest cosine {worst_c} "
          f"over {sum(1 for n in names if 'grad::' + n in g.files)} full tensors")
    assert abs(tot_h - tot_r) <= norm_rel_global * tot_r, (tot_h, tot_r)
    assert worst_n[0] <= norm_rel_tensor, worst_n
    assert worst_c[0] >= cos_min, worst_c


def test_backbone_errors_and_surface():
    dev = _dev()
    m = _backbone(192, 1, dev)
    assert m.num_features == m.embed_dim == 192 and m.token_learner.num_patches == 196 and m.max_channels == 10
    x = torch.zeros(3, 1, 224, 224, device=dev)
    with pytest.raises(RuntimeError):
        m(x, 0, [[2]])  # channel count mismatch
    with pytest.raises(RuntimeError):
        m(torch.zeros(11, 1, 32, 32, device=dev), 0, [[11]])  # > 10 channels (reference: torch.stack fails)
    with pytest.raises(RuntimeError):
        m(x.cpu(), 0, [[3]])  # no CPU fallback


def test_backbone_with_other_constructor_arguments_vs_golden_and_oracle():
    """ChAdaViT(...) with patch 8 on a 64-pixel grid, a 96-pixel position grid interpolated up to 224, depth 2 / 3, six heads, and
    max_number_channels = 5 (the reference then adds no channel tokens, chada_vit.py:219, 248): CLS and all-token outputs against the
    golden of the reference built with the same arguments, gradients of a scalar of the all-token output against the oracle."""
    from chadavit_amd.backbones.vit.chada_vit import ChAdaViT
    from tests.golden_util import CTOR_CASES, ctor_case_state
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, "backbone_ctor_args.npz"))
    for ci, (kw, nch, sizes, seed) in enumerate(CTOR_CASES):
        sd = ctor_case_state(kw, seed)
        m = ChAdaViT(return_all_tokens=False, **kw)
        m.load_state_dict(sd)
        m = m.to(dev)
        crops, _, ncl = R.collate(P.make_images(nch, sizes, seed=seed + 100))
        crops = crops if isinstance(crops, list) else [crops]
        add_chan = kw["max_number_channels"] == 10
        for k, x in enumerate(crops):
            with torch.no_grad():
                m.return_all_tokens = False
                cls = m(x.to(dev), k, ncl)
                m.return_all_tokens = True
                allt = m(x.to(dev), k, ncl)
            ref = torch.from_numpy(g[f"c{ci}_cls{k}"])
            assert _cos(cls, ref) >= 0.999 and _rel(cls, ref) <= 2e-2, (ci, k, _cos(cls, ref), _rel(cls, ref))
            assert list(allt.shape) == [int(v) for v in g[f"c{ci}_all{k}_shape"]]
            rows = torch.from_numpy(g[f"c{ci}_all{k}_rows"])
            assert _rel(allt[rows.to(dev)], torch.from_numpy(g[f"c{ci}_all{k}_vals"])) <= 2e-2, (ci, k)
        # backward through the all-token output of the first crop
        x = crops[0]
        wgt = P.tensor(tuple(int(v) for v in g[f"c{ci}_all0_shape"]), f"ctor.w{ci}", 1.0, seed=5)
        m.zero_grad(set_to_none=True)
        (m(x.to(dev), 0, ncl).float() * wgt.to(dev)).sum().backward()
        po = {n: v.detach().clone().requires_grad_(True) for n, v in sd.items()}
        (R.backbone_ragged(po, x, ncl[0], kw["num_heads"], final_eps=1e-5, return_all_tokens=True, patch=kw["patch_size"],
                           add_channel_token=add_chan) * wgt).sum().backward()
        named = dict(m.named_parameters())
        worst = (1.0, None)
        for n, v in po.items():
            if v.grad is None or float(v.grad.norm()) <= 1e-6 * np.sqrt(v.numel()):
                if n == "channel_token" and not add_chan:
                    assert named[n].grad is None or float(named[n].grad.abs().max()) == 0.0   # never used: no gradient
                continue
            worst = min(worst, (_cos(named[n].grad, v.grad), n))
            assert abs(float(named[n].grad.double().norm()) - float(v.grad.double().norm())) <= 6e-2 * float(v.grad.double().norm()) + 1e-6, (ci, n)
        assert worst[0] >= 0.99, (ci, worst)


def _cfg(D, PR, n_large, n_small, clip_grad=0.0, lr=5e-4, wd=1e-4, base_tau=0.9995, use_bn_in_head=False, norm_last_layer=True,
         hidden=2048, bott=256):
    from chadavit_amd.utils.misc import AttrDict
    return AttrDict({
        "method": "dino",
        "backbone": {"name": "vit_channels", "kwargs": {"embed_dim": D, "patch_size": 16, "return_all_tokens": False,
                                                        "max_number_channels": 10}},
        "data": {"dataset": "synthetic", "num_classes": 7, "max_img_channels": 10, "img_channels": 1,
                 "num_large_crops": n_large, "num_small_crops": n_small},
        "channels_strategy": "multi_channels", "mixed_channels": True, "weights_init": "random", "max_epochs": 10,
        "optimizer": {"name": "adamw", "batch_size": 4, "lr": lr, "weight_decay": wd, "classifier_lr": 0.1},
        "scheduler": {"name": "none"},
        "momentum": {"base_tau": base_tau, "final_tau": 1.0},
        "method_kwargs": {"proj_hidden_dim": hidden, "proj_output_dim": bott, "num_prototypes": PR, "clip_grad": clip_grad,
                          "freeze_last_layer": 1, "warmup_teacher_temperature_epochs": 3, "use_bn_in_head": use_bn_in_head,
                          "norm_last_layer": norm_last_layer},
    })


def _assert_block_kernel_dispatch(summary, D, expect_fused):
    """Which forward chain ran, from the launch record: the whole-block kernel (`proj_ffn_ln_fwd`: out-proj + residual + norm1 +
    FFN + norm2 + next norm1 + next QKV in one launch -- the path bench.py times at cfg2) or the GEMM + LayerNorm chain."""
    from chadavit_amd import ops
    blk = [v["launches"] for k, v in summary.items() if k[0] == "proj_ffn_ln_fwd"]
    outproj = [v["launches"] for k, v in summary.items()
               if k[0] == "gemm_nt" and k[2] == D and k[3] == D and k[4] == ops.EPI_RESID]
    ln2 = [v["launches"] for k, v in summary.items() if k[0] == "layernorm_fwd2"]
    # (the LAST block runs on the CLS rows only -- ChAdaViT.cls_only_last_block -- as a GEMM + LayerNorm chain on B rows: 11 whole-block
    # launches per pass and one small out-proj GEMM)
    if expect_fused == "all":
        assert sum(blk) >= 22 and sum(blk) % 11 == 0, summary.keys()   # 11 per backbone pass (student, teacher[, local])
        assert sum(outproj) == sum(blk) // 11 and not ln2, (outproj, ln2)
    elif expect_fused == "global":  # the global-crop passes are above the row threshold, the local-crop pass below it
        assert sum(blk) == 22 and sum(outproj) == 12 + 2 and sum(ln2) == 11, (blk, outproj, ln2)
    else:
        assert not blk and sum(outproj) >= 24, (blk, outproj)


# (golden, fused_min_rows override, expected dispatch).  `None` = the default threshold (24576 rows).  The Tiny goldens run
# twice: on the GEMM + LayerNorm chain their row counts select by default, and with the whole-block kernel forced, so that the
# dispatch bench.py times is the one held against the reference.  step_tiny_fused_rows is large enough (26282 rows per global
# pass) to take that dispatch by itself.  step_small_mixed / step_base_c10 put the D = 384 / 768 kernels (for D = 384 with the
# whole-block forward kernel forced at this size) under the
# reference (dh = 192 / 384 attention, K = 384 / 768 GEMMs).
_STEP_CASES = [("step_tiny_multicrop", None, "none"), ("step_tiny_c1_clip", None, "none"),
               ("step_tiny_multicrop", 0, "all"), ("step_tiny_c1_clip", 0, "all"),
               # (slow: the 26 282-row step incl. its oracle comparison, 48 s -- `-m "gpu and slow"`; the same golden stays in the default
               #  run through test_training_step_is_deterministic_under_allocator_churn and the x170 replica of the fused dispatch)
               pytest.param("step_tiny_fused_rows", None, "global", marks=pytest.mark.slow),
               ("step_small_mixed", None, "none"), ("step_small_mixed", 0, "small_fused"), ("step_base_c10", None, "none")]


_ORACLE_STEP_CACHE = {}


@pytest.mark.parametrize("name,fused_min_rows,dispatch", _STEP_CASES)
def test_training_step_vs_golden_and_oracle(name, fused_min_rows, dispatch):
    from chadavit_amd import ops
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    D, PR = int(g["D"]), int(g["P"])
    nch = [int(c) for c in g["nch"]]
    sizes = [int(s) for s in g["sizes"]]
    n_large, epoch, clip = int(g["n_large"]), int(g["epoch"]), float(g["clip_grad"])
    sd = build_sd(D, PR)
    model = DINO(_cfg(D, PR, n_large, len(sizes) - n_large, clip_grad=clip, lr=float(g["lr"]), wd=float(g["wd"]),
                      base_tau=float(g["base_tau"])))
    model.load_state_dict(sd)
    model = model.to(dev)
    imgs = P.make_images(nch, sizes, seed=7)
    crops, labels, ncl = one_channel_collate_fn(imgs)
    crops = crops if isinstance(crops, list) else [crops]
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
    tr = Trainer(max_epochs=10, steps_per_epoch=int(g["max_steps"]) // 10)
    tr.current_epoch = epoch
    tr.attach(model)
    model.current_epoch = epoch
    model.on_train_epoch_start()
    if fused_min_rows is not None:
        model.backbone.fused_min_rows = model.momentum_backbone.fused_min_rows = fused_min_rows
    with ops.LaunchProfiler() as prof:
        loss = model.training_step(batch, 1)
        loss.backward()
        model.on_after_backward()
    if D == 192:
        _assert_block_kernel_dispatch(prof.summary(), D, dispatch)
    elif dispatch == "small_fused":  # D = 384: the whole-block forward kernel (8 waves x 16 rows) and the fused backward dX pass
        summ = prof.summary()
        # student, teacher, local passes; the last block of each runs on its CLS rows as a GEMM chain (cls_only_last_block)
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "proj_ffn_ln_fwd") == 33
        assert not [k for k in summ if k[0] == "ffn_ln_fwd"]
        # the next block's QKV rides in the kernel: one stand-alone in_proj GEMM per pass (block 0); stand-alone out-proj / linear1
        # only for the last block's CLS rows
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "gemm_nt" and k[2] == 3 * D and k[3] == D) == 3
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "gemm_nt" and k[2] == D and k[3] == D and k[4] == ops.EPI_RESID) == 3
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "ffn_bwd_dx") == 11
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "gemm_nt" and k[2] == 2048 and k[3] == D and k[4] == ops.EPI_RELU) == 3
    # ---- loss
    assert abs(loss.item() - float(g["loss"])) <= 2e-2, (loss.item(), float(g["loss"]))
    # ---- gradients vs golden norms and vs oracle tensors
    # (the oracle's CPU step is the slow part of this test: computed once per golden and shared by the parametrisations; for the
    #  26 282-row golden -- 85 s of CPU on 8 cores -- skipped only with CHADAVIT_FAST_TESTS set; the reference's own gradient values and
    #  per-pass outputs in the golden are compared either way)
    grads_o = aux_o = None
    if name != "step_tiny_fused_rows" or not os.environ.get("CHADAVIT_FAST_TESTS"):
        if name not in _ORACLE_STEP_CACHE:
            _ORACLE_STEP_CACHE[name] = R.training_step(sd, crops, ncl, n_large, float(g["teacher_temp"]), freeze_last_layer=epoch < 1, clip_grad=clip)[1::2]
        grads_o, aux_o = _ORACLE_STEP_CACHE[name]
    named = dict(model.named_parameters())
    # ---- what the passes produced: student CLS features of every crop (the local crops' too: nothing downstream reads them, so no
    # loss or gradient bar can see that pass), teacher CLS features, student and teacher logits -- against the reference (golden
    # `outs::*`) and against the oracle's aux on all columns.  CLS cosine >= 0.999, rel-L2 <= 2e-2; every row on its own <= 8e-2.
    from tests.golden_util import grad_subsets_vs_golden, step_outputs_vs_golden, step_outputs_vs_oracle
    torch.cuda.synchronize()
    print(name, dispatch, "outputs vs reference (worst cos, rel, key):", step_outputs_vs_golden(model._last_outs, g))
    if aux_o is not None:
        print(name, dispatch, "outputs vs oracle:", step_outputs_vs_oracle(model._last_outs, aux_o))
    # ---- every gradient tensor against the reference's own values on the golden's spread of elements
    print(name, dispatch, "gradient spread vs reference (worst cos, name):", grad_subsets_vs_golden(named, g))
    none_names = set(str(n) for n in g["none_grad_names"])
    for n in none_names:
        assert named[n].grad is None, n
    tot_h = tot_r = 0.0
    worst = (1.0, None)
    for n, gn in zip(g["grad_names"], g["grad_norms"]):
        n = str(n)
        gh = named[n].grad
        assert gh is not None, n
        tot_h += gh.double().norm().item() ** 2
        tot_r += float(gn) ** 2
        if grads_o is None:   # per-tensor norm against the golden's instead of the cosine against the oracle's tensor
            if float(gn) > 1e-3 * float(max(g["grad_norms"])):
                assert abs(gh.double().norm().item() - float(gn)) <= 6e-2 * float(gn), (n, gh.double().norm().item(), float(gn))
            continue
        go = grads_o[n]
        if float(gn) > 1e-4 * np.sqrt(go.numel()) * 1e-2:
            c = _cos(gh, go)
            if c < worst[0]:
                worst = (c, n)
    assert abs(np.sqrt(tot_h) - np.sqrt(tot_r)) <= 5e-2 * np.sqrt(tot_r), (np.sqrt(tot_h), np.sqrt(tot_r))
    assert worst[0] >= 0.99, worst
    for key in ("backbone.norm.weight", "backbone.cls_token", "backbone.channel_token", "head.mlp.4.bias"):
        ref = torch.from_numpy(g["grad::" + key])
        assert _cos(named[key].grad, ref) >= 0.99, key
    # channel slots no image uses get exactly zero gradient
    assert float(named["backbone.channel_token"].grad[0, max(nch):].abs().max()) == 0.0 if max(nch) < 10 else True
    # ---- centre
    np.testing.assert_allclose(model.dino_loss_func.center[0, :256].float().cpu().numpy(), g["center_new"], atol=2e-3)
    # ---- optimiser + EMA + tau (hook order of SURVEY 3.2)
    tr.optimizer.step()
    tr.global_step += 1
    model.optimizer_zero_grad(epoch, 1, tr.optimizer)
    model.on_train_batch_end(None, batch, 1)
    assert abs(model.momentum_updater.cur_tau - float(g["tau_next"])) < 1e-12
    post = dict(zip([str(n) for n in g["post_names"]], g["post_sums"]))
    named = dict(model.named_parameters())
    # AdamW's first step moves every weight by ~lr*sign(g): compare element-wise on a small tensor and by sums
    # (elements whose gradient is ~0 may flip sign under bf16 -> differ by 2*lr; allow a few of those)
    dnw = np.abs(named["backbone.norm.weight"].detach().cpu().numpy() - g["post::backbone.norm.weight"])
    assert dnw.max() <= 2.1 * float(g["lr"]) and (dnw > 2e-4).mean() <= 0.05, (dnw.max(), (dnw > 2e-4).mean())
    np.testing.assert_allclose(named["momentum_backbone.norm.weight"].detach().cpu().numpy(),
                               g["post::momentum_backbone.norm.weight"], atol=1e-6)
    for n in ("momentum_backbone.blocks.3.linear1.weight", "momentum_head.mlp.2.weight", "momentum_head.last_layer.weight_v"):
        v = named[n].double().sum().item()
        assert abs(v - post[n]) <= 1e-4 * (abs(post[n]) + named[n].numel() ** 0.5), n
    assert all(p.grad is None for p in model.parameters())


def test_training_step_at_other_image_sizes_vs_oracle():
    """Crop sides other than 224 / 96: 112 and 100 as the two global crops (the latter loses four pixels per side to the stride-16 conv,
    chada_vit.py:118-134) and 32 / 16 as local crops (four patches / ONE patch per channel: sequences of 2-13 tokens).  The oracle is
    pinned to the reference at exactly these sizes by golden `backbone_tiny_sizes`; here the whole step -- loss, every gradient
    (incl. the bicubic position-embedding resize's and the patch conv's), the centre -- is held against it."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    D, PR, nch, sizes = 192, 4096, [2, 1, 3, 1], [112, 100, 32, 16]
    sd = build_sd(D, PR)
    model = DINO(_cfg(D, PR, 2, 2))
    model.load_state_dict(sd)
    model = model.to(dev)
    crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, sizes, seed=17))
    tr = Trainer(max_epochs=10, steps_per_epoch=10)
    tr.current_epoch = 1
    tr.attach(model)
    model.current_epoch = 1
    model.on_train_epoch_start()
    loss = model.training_step(([c.to(dev) for c in crops], labels.to(dev), ncl), 1)
    loss.backward()
    model.on_after_backward()
    tt = float(model.dino_loss_func.teacher_temp_schedule[1])
    loss_o, grads_o, newc_o, aux = R.training_step(sd, crops, ncl, 2, tt, freeze_last_layer=False)
    assert abs(loss.item() - float(loss_o)) <= 2e-2, (loss.item(), float(loss_o))
    named = dict(model.named_parameters())
    tot_h = tot_o = 0.0
    worst = (1.0, None)
    for n, go in grads_o.items():
        if go is None:
            assert named[n].grad is None, n
            continue
        gh = named[n].grad
        assert gh is not None, n
        tot_h += gh.double().norm().item() ** 2
        tot_o += go.double().norm().item() ** 2
        if float(go.norm()) > 1e-6 * np.sqrt(go.numel()):
            worst = min(worst, (_cos(gh, go), n))
    assert abs(np.sqrt(tot_h) - np.sqrt(tot_o)) <= 5e-2 * np.sqrt(tot_o), (np.sqrt(tot_h), np.sqrt(tot_o))
    assert worst[0] >= 0.99, worst
    np.testing.assert_allclose(model.dino_loss_func.center.float().cpu().numpy(), newc_o.numpy(), atol=2e-3)


@pytest.mark.parametrize("D", [128, 256, 512])
def test_training_step_at_other_embed_dims_vs_oracle(D):
    """`embed_dim` values between the benchmark's three (the factory keeps two heads: head widths 64 / 128 / 256): the GEMM + LayerNorm
    chain and the register-staged attention kernels, whole step against the oracle."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    PR, nch, sizes = 4096, [2, 1, 3], [224, 224, 96]
    sd = build_sd(D, PR)
    model = DINO(_cfg(D, PR, 2, 1))
    model.load_state_dict(sd)
    model = model.to(dev)
    crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, sizes, seed=23))
    tr = Trainer(max_epochs=10, steps_per_epoch=10)
    tr.current_epoch = 1
    tr.attach(model)
    model.current_epoch = 1
    model.on_train_epoch_start()
    loss = model.training_step(([c.to(dev) for c in crops], labels.to(dev), ncl), 1)
    loss.backward()
    model.on_after_backward()
    loss_o, grads_o, newc_o, aux = R.training_step(sd, crops, ncl, 2, float(model.dino_loss_func.teacher_temp_schedule[1]), freeze_last_layer=False)
    assert abs(loss.item() - float(loss_o)) <= 2e-2, (loss.item(), float(loss_o))
    named = dict(model.named_parameters())
    tot_h = tot_o = 0.0
    worst = (1.0, None)
    for n, go in grads_o.items():
        if go is None:
            assert named[n].grad is None, n
            continue
        gh = named[n].grad
        assert gh is not None, n
        tot_h += gh.double().norm().item() ** 2
        tot_o += go.double().norm().item() ** 2
        if float(go.norm()) > 1e-6 * np.sqrt(go.numel()):
            worst = min(worst, (_cos(gh, go), n))
    assert abs(np.sqrt(tot_h) - np.sqrt(tot_o)) <= 5e-2 * np.sqrt(tot_o), (np.sqrt(tot_h), np.sqrt(tot_o))
    assert worst[0] >= 0.99, worst
    np.testing.assert_allclose(model.dino_loss_func.center.float().cpu().numpy(), newc_o.numpy(), atol=2e-3)


@pytest.mark.parametrize("name", ["step_tiny_trained_prototype_norms", "step_tiny_trained_prototype_norms_epoch0"])
def test_training_step_with_trained_prototype_norms_vs_golden_and_oracle(name):
    """method_kwargs.norm_last_layer = False: the magnitudes `last_layer.weight_g` of the weight-normed prototypes are trained like
    every other head parameter (dino.py:83-84) -- their gradient against the reference's (golden, element-wise) at epoch 1; at
    epoch 0 (< freeze_last_layer) BOTH last-layer gradients are dropped (dino.py:374-376)."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    D, PR, epoch = int(g["D"]), int(g["P"]), int(g["epoch"])
    nch, sizes = [int(c) for c in g["nch"]], [int(s) for s in g["sizes"]]
    sd = build_sd(D, PR)
    model = DINO(_cfg(D, PR, int(g["n_large"]), len(sizes) - int(g["n_large"]), norm_last_layer=False))
    model.load_state_dict(sd)
    model = model.to(dev)
    assert model.head.last_layer.weight_g.requires_grad and not model.momentum_head.last_layer.weight_g.requires_grad
    crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, sizes, seed=7))
    tr = Trainer(max_epochs=10, steps_per_epoch=10)
    tr.current_epoch = epoch
    tr.attach(model)
    model.current_epoch = epoch
    model.on_train_epoch_start()
    loss = model.training_step(([c.to(dev) for c in crops], labels.to(dev), ncl), 1)
    loss.backward()
    model.on_after_backward()
    assert abs(loss.item() - float(g["loss"])) <= 2e-2
    named = dict(model.named_parameters())
    from tests.golden_util import grad_subsets_vs_golden, step_outputs_vs_golden
    torch.cuda.synchronize()
    step_outputs_vs_golden(model._last_outs, g)     # per-pass outputs (features of every crop, teacher features, both logits)
    grad_subsets_vs_golden(named, g)                # every gradient tensor on the reference's spread of elements
    for n in (str(n) for n in g["none_grad_names"]):
        assert named[n].grad is None, n
    tot_h = tot_r = 0.0
    for n, gn in zip(g["grad_names"], g["grad_norms"]):
        gh = named[str(n)].grad
        assert gh is not None, str(n)
        tot_h += gh.double().norm().item() ** 2
        tot_r += float(gn) ** 2
    assert abs(np.sqrt(tot_h) - np.sqrt(tot_r)) <= 5e-2 * np.sqrt(tot_r)
    if epoch >= 1:
        dg, ref = named["head.last_layer.weight_g"].grad, torch.from_numpy(g["grad::head.last_layer.weight_g"])
        assert dg.shape == ref.shape and _cos(dg, ref) >= 0.99, _cos(dg, ref)
        assert abs(float(dg.double().norm()) - float(ref.double().norm())) <= 5e-2 * float(ref.double().norm())
    # the optimiser moves weight_g only when it has a gradient (AdamW's first step: lr * sign)
    g0 = model.head.last_layer.weight_g.detach().clone()
    tr.optimizer.step()
    moved = (model.head.last_layer.weight_g.detach() - g0).abs().max().item()
    assert (moved > 0) == (epoch >= 1), moved


def test_training_step_with_the_standard_multicrop_loss_vs_golden_and_oracle():
    """`method_kwargs.standard_multicrop_loss = True` -- a BUILD-SIDE option, off by default and NOT the reference's behaviour (its DINO
    computes the local crops' features and drops them): the DINO paper's multi-crop loss, local crops through the head and into the loss as
    extra student views (2 teacher x 5 student views here), trained through.  Golden: a subclass of the reference's DINO whose
    multicrop_forward also returns the head's output, with the reference's DINOLoss chunking the student logits into all crops
    (tests/golden/make_golden.py `standard_multicrop`).  Loss, every pass's outputs (local crops' logits included), gradients (two
    backward passes per network accumulate), the centre; and the parity path must be untouched by the flag being off."""
    from chadavit_amd import ops
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    from tests.golden_util import grad_subsets_vs_golden, step_outputs_vs_golden, step_outputs_vs_oracle
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, "step_tiny_standard_multicrop.npz"))
    D, PR, n_large, epoch = int(g["D"]), int(g["P"]), int(g["n_large"]), int(g["epoch"])
    nch, sizes = [int(c) for c in g["nch"]], [int(s) for s in g["sizes"]]
    sd = build_sd(D, PR)
    crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, sizes, seed=7))
    loss_o, grads_o, newc_o, aux = R.training_step(sd, crops, ncl, n_large, float(g["teacher_temp"]), freeze_last_layer=False, standard_multicrop=True)
    for fused_min_rows in (None, 0):   # GEMM + LayerNorm chain, and the whole-block kernels forced
        cfg = _cfg(D, PR, n_large, len(sizes) - n_large, lr=float(g["lr"]), wd=float(g["wd"]), base_tau=float(g["base_tau"]))
        cfg.method_kwargs.standard_multicrop_loss = True
        model = DINO(cfg)
        assert model.standard_multicrop_loss and model.dino_loss_func.num_large_crops == len(sizes)
        model.load_state_dict(sd)
        model = model.to(dev)
        if fused_min_rows is not None:
            model.backbone.fused_min_rows = model.momentum_backbone.fused_min_rows = fused_min_rows
        tr = Trainer(max_epochs=10, steps_per_epoch=10)
        tr.current_epoch = epoch
        tr.attach(model)
        model.current_epoch = epoch
        model.on_train_epoch_start()
        with ops.LaunchProfiler() as prof:
            loss = model.training_step(([c.to(dev) for c in crops], labels.to(dev), ncl), 1)
            loss.backward()
            model.on_after_backward()
        summ = prof.summary()   # both backbone passes of the student ran with saves (the local one too): two sets of backward launches
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "attn_bwd") >= 2 * 11
        assert abs(loss.item() - float(g["loss"])) <= 2e-2, (loss.item(), float(g["loss"]))
        torch.cuda.synchronize()
        assert model._last_outs["z"].shape[0] == len(sizes) * len(nch)
        print("standard multicrop, outputs vs reference-subclass golden:", step_outputs_vs_golden(model._last_outs, g))
        print("standard multicrop, outputs vs oracle:", step_outputs_vs_oracle(model._last_outs, aux))
        named = dict(model.named_parameters())
        print("standard multicrop, gradient spread:", grad_subsets_vs_golden(named, g))
        tot_h = tot_r = 0.0
        worst = (1.0, None)
        for n, gn in zip(g["grad_names"], g["grad_norms"]):
            gh = named[str(n)].grad
            assert gh is not None, str(n)
            tot_h += gh.double().norm().item() ** 2
            tot_r += float(gn) ** 2
            go = grads_o[str(n)]
            if float(gn) > 1e-4 * np.sqrt(go.numel()) * 1e-2:
                worst = min(worst, (_cos(gh, go), str(n)))
        assert abs(np.sqrt(tot_h) - np.sqrt(tot_r)) <= 5e-2 * np.sqrt(tot_r), (np.sqrt(tot_h), np.sqrt(tot_r))
        assert worst[0] >= 0.99, worst
        for n in (str(n) for n in g["none_grad_names"]):
            assert named[n].grad is None, n
        np.testing.assert_allclose(model.dino_loss_func.center[0, :256].float().cpu().numpy(), g["center_new"], atol=2e-3)
        tr.optimizer.step()   # and the step goes through the optimiser
        assert all(torch.isfinite(p_).all() for p_ in model.parameters())


def test_training_step_is_deterministic_under_allocator_churn():
    """The same training step (golden step_tiny_fused_rows: whole-block kernels on the global passes, GEMM chain on the local one)
    three times in one process, the caching allocator's free blocks refilled with large values and with NaN in between: loss and
    every gradient tensor must come out bit-identical.  A kernel that reads a buffer before it is complete (a too generous
    `s_waitcnt vmcnt(N)` in front of an LDS-DMA stage), outside its bounds or uninitialised shows up here as a run-to-run difference
    -- the tolerance-based parity tests above let a few wrong rows of the TEACHER pass through (it has no gradient, and the
    loss moved in its fourth digit)."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, "step_tiny_fused_rows.npz"))
    D, PR = int(g["D"]), int(g["P"])
    nch, sizes = [int(c) for c in g["nch"]], [int(s) for s in g["sizes"]]
    crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, sizes, seed=7))

    def poison(value):
        bufs = [torch.full((mb * 1024 * 1024 // 2,), value, device=dev, dtype=torch.bfloat16) for mb in (1024, 512, 512, 256, 256, 128, 128, 64, 64, 32, 32, 16, 8, 4, 2, 1) for _ in range(3)]
        torch.cuda.synchronize()
        del bufs

    ref = None
    for val in (None, 3.0e4, float("nan")):
        if val is not None:
            poison(val)
        model = DINO(_cfg(D, PR, int(g["n_large"]), len(sizes) - int(g["n_large"])))
        model.load_state_dict(build_sd(D, PR))
        model = model.to(dev)
        tr = Trainer(max_epochs=10, steps_per_epoch=10)
        tr.current_epoch = 1
        tr.attach(model)
        model.current_epoch = 1
        model.on_train_epoch_start()
        loss = model.training_step(([c.to(dev) for c in crops], labels.to(dev), ncl), 1)
        loss.backward()
        model.on_after_backward()
        torch.cuda.synchronize()
        got = (loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        if ref is None:
            ref = got
        else:
            assert torch.equal(got[0], ref[0]), (float(got[0]), float(ref[0]), val)
            bad = [n for n, t in got[1].items() if not torch.equal(t, ref[1][n])]
            assert not bad, (val, len(bad), bad[:5])
        del model, tr


@pytest.mark.parametrize("nch,sizes,PR,hidden,bott", [([1], [224, 224], 4096, 2048, 256),            # ONE one-channel image
                                                      ([2, 1], [224, 224, 96], 65536, 2048, 256),    # the linear yaml's 65 536 prototypes
                                                      ([3, 1, 2], [224, 224], 1024, 512, 128)])      # another projector shape
def test_training_step_other_head_shapes_and_tiny_batches_vs_oracle(nch, sizes, PR, hidden, bott):
    """method_kwargs.{num_prototypes, proj_hidden_dim, proj_output_dim} other than the benchmark's, and a batch of one image: whole
    step against the oracle (loss, every gradient, centre)."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    D = 192
    sd = build_sd(D, PR, hidden=hidden, bott=bott)
    model = DINO(_cfg(D, PR, 2, len(sizes) - 2, hidden=hidden, bott=bott))
    model.load_state_dict(sd)
    model = model.to(dev)
    crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, sizes, seed=29))
    tr = Trainer(max_epochs=10, steps_per_epoch=10)
    tr.current_epoch = 1
    tr.attach(model)
    model.current_epoch = 1
    model.on_train_epoch_start()
    loss = model.training_step(([c.to(dev) for c in crops], labels.to(dev), ncl), 1)
    loss.backward()
    model.on_after_backward()
    loss_o, grads_o, newc_o, aux = R.training_step(sd, crops, ncl, 2, float(model.dino_loss_func.teacher_temp_schedule[1]), freeze_last_layer=False)
    assert abs(loss.item() - float(loss_o)) <= 2e-2, (loss.item(), float(loss_o))
    named = dict(model.named_parameters())
    tot_h = tot_o = 0.0
    worst = (1.0, None)
    for n, go in grads_o.items():
        if go is None:
            assert named[n].grad is None, n
            continue
        gh = named[n].grad
        assert gh is not None, n
        tot_h += gh.double().norm().item() ** 2
        tot_o += go.double().norm().item() ** 2
        if float(go.norm()) > 1e-6 * np.sqrt(go.numel()):
            worst = min(worst, (_cos(gh, go), n))
    assert abs(np.sqrt(tot_h) - np.sqrt(tot_o)) <= 5e-2 * np.sqrt(tot_o), (np.sqrt(tot_h), np.sqrt(tot_o))
    assert worst[0] >= 0.99, worst
    np.testing.assert_allclose(model.dino_loss_func.center.float().cpu().numpy(), newc_o.numpy(), atol=2e-3)


@pytest.mark.parametrize("opt_name,kwargs", [("adamw", {}), ("sgd", {"momentum": 0.9}), ("lars", {"momentum": 0.9, "eta": 0.02})])
def test_optimizer_state_dict_resumes_bit_exact(opt_name, kwargs, tmp_path):
    """`optimizer.state_dict()` of the fused optimisers holds the moments in torch's own per-parameter layout (what Lightning
    checkpoints): two steps, save model + optimiser, load both into a fresh model, third step -- bit-identical to the uninterrupted
    run; and torch.optim's own class accepts the saved state (the reference's optimiser could resume from it and vice versa)."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    from chadavit_amd.utils.checkpoint import load_checkpoint, save_checkpoint
    dev = _dev()
    D, PR = 192, 4096
    crops, labels, ncl = one_channel_collate_fn(P.make_images([2, 1, 3], [224, 224, 96], seed=31))
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)

    def make():
        cfg = _cfg(D, PR, 2, 1, lr=1e-3 if opt_name == "adamw" else 0.05)
        cfg.optimizer.name = opt_name
        cfg.optimizer.kwargs = dict(kwargs)
        m = DINO(cfg)
        m.load_state_dict(build_sd(D, PR))
        m = m.to(dev)
        t = Trainer(max_epochs=10, steps_per_epoch=10)
        t.current_epoch = 1
        t.attach(m)
        return m, t

    a, ta = make()
    for i in range(2):
        ta.train_step(batch, 1 + i)
    save_checkpoint(a, str(tmp_path / "m.ckpt"), epoch=1, global_step=ta.global_step)
    osd = ta.optimizer.state_dict()
    torch.save(osd, str(tmp_path / "o.pt"))
    n_params = sum(len(g_["params"]) for g_ in osd["param_groups"])
    key = "exp_avg" if opt_name == "adamw" else "momentum_buffer"
    stepped = [i for i, st in osd["state"].items() if key in st]
    assert len(stepped) >= 150 and len(stepped) <= n_params, (len(stepped), n_params)   # every parameter that got a gradient
    flat_params = [p for g_ in ta.optimizer.param_groups for p in g_["params"]]
    assert all(osd["state"][i][key].shape == flat_params[i].shape for i in stepped)
    ta.train_step(batch, 3)
    b, tb = make()
    load_checkpoint(b, str(tmp_path / "m.ckpt"))
    tb.optimizer.load_state_dict(torch.load(str(tmp_path / "o.pt"), weights_only=False))
    tb.global_step = ta.global_step - 1
    b.momentum_updater.cur_tau = None   # (recomputed below exactly as the uninterrupted run had it before its third step)
    b.momentum_updater.cur_tau = float(a.momentum_updater.base_tau)
    for i in range(2):   # tau schedule state of the uninterrupted run after two steps (host arithmetic, not part of any state_dict)
        b.momentum_updater.update_tau(cur_step=i + 1, max_steps=tb.estimated_stepping_batches)
    b.last_step = 2
    tb.train_step(batch, 3)
    sa, sb = a.state_dict(), b.state_dict()
    bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
    assert not bad, bad[:6]
    # torch's own optimiser class loads the saved state (same parameter order, same shapes)
    ref_cls = {"adamw": torch.optim.AdamW, "sgd": torch.optim.SGD, "lars": torch.optim.SGD}[opt_name]
    groups = [{"params": [torch.nn.Parameter(torch.zeros_like(p, device="cpu")) for p in g_["params"]]} for g_ in ta.optimizer.param_groups]
    ref_opt = ref_cls(groups, lr=1e-3) if opt_name == "adamw" else ref_cls(groups, lr=0.05, momentum=0.9)
    ref_opt.load_state_dict({"state": {i: {k: v for k, v in st.items() if k in ("step", "exp_avg", "exp_avg_sq", "momentum_buffer")}
                                       for i, st in osd["state"].items()}, "param_groups": ref_opt.state_dict()["param_groups"]})
    assert len(ref_opt.state) == len(stepped)


def test_load_state_dict_after_training_steps_refreshes_every_shadow_copy():
    """Loading weights into a model that has already trained (parameters are views of the flat slabs; bf16 shadows, transposes,
    packed FFN streams, weight-normed prototypes are cached per parameter version): the next forward must see the loaded weights --
    features and head outputs bit-identical to a fresh model holding them; also for weights written in place under no_grad."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    D, PR = 192, 4096
    crops, labels, ncl = one_channel_collate_fn(P.make_images([2, 1, 3], [224, 224], seed=37))
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
    sd = build_sd(D, PR)
    fresh = DINO(_cfg(D, PR, 2, 0))
    fresh.load_state_dict(sd)
    fresh = fresh.to(dev).eval()
    fresh.list_num_channels = ncl
    with torch.no_grad():
        want = fresh(batch[0][0], 0)
        want_t = fresh.momentum_forward(batch[0][0], 0)
    m = DINO(_cfg(D, PR, 2, 0, lr=1e-2))
    m.load_state_dict({k: v + 0.05 for k, v in sd.items()})
    m = m.to(dev)
    tr = Trainer(max_epochs=10, steps_per_epoch=10)
    tr.current_epoch = 1
    tr.attach(m)
    for i in range(2):
        tr.train_step(batch, 1 + i)
    m.load_state_dict(sd)
    m.eval()
    m.list_num_channels = ncl
    with torch.no_grad():
        got = m(batch[0][0], 0)
        got_t = m.momentum_forward(batch[0][0], 0)
    for k in ("feats", "z", "logits"):
        assert torch.equal(got[k], want[k]), k
    assert torch.equal(got_t["feats"], want_t["feats"]) and torch.equal(got_t["z"], want_t["z"])
    # in-place edits under no_grad are seen too
    with torch.no_grad():
        m.backbone.norm.weight.mul_(2.0)
        fresh.backbone.norm.weight.mul_(2.0)
        assert torch.equal(m(batch[0][0], 0)["feats"], fresh(batch[0][0], 0)["feats"])


@pytest.mark.parametrize("fused_min_rows", [None, 0])
def test_gradient_accumulation_over_two_backward_passes(fused_min_rows):
    """accumulate_grad_batches = 2 (Lightning: two training_step + backward rounds before one optimiser step, base.py:331-336): the
    gradients after the second backward are the SUM of the two passes' own gradients -- every tensor, backbone and head, on both
    dispatches of the block -- and zero_grad(set_to_none=True) starts the next accumulation from scratch."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    D, PR = 192, 4096
    model = DINO(_cfg(D, PR, 2, 1))
    model.load_state_dict(build_sd(D, PR))
    model = model.to(dev)
    if fused_min_rows is not None:
        model.backbone.fused_min_rows = model.momentum_backbone.fused_min_rows = fused_min_rows
    tr = Trainer(max_epochs=10, steps_per_epoch=10)
    tr.current_epoch = 1
    tr.attach(model)
    model.current_epoch = 1
    model.on_train_epoch_start()
    batches = []
    for seed, nch in ((41, [2, 1, 3]), (43, [1, 4])):
        crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, [224, 224, 96], seed=seed))
        batches.append(([c.to(dev) for c in crops], labels.to(dev), ncl))
    center0 = model.dino_loss_func.center.clone()

    def grads_of(bs):
        model.dino_loss_func.center.copy_(center0)     # (the centre moves with every loss call: same starting point for each variant)
        tr.optimizer.zero_grad(set_to_none=True)
        for b in bs:
            model.training_step(b, 1).backward()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    g0, g1 = grads_of(batches[:1]), grads_of(batches[1:])
    model.dino_loss_func.center.copy_(center0)
    tr.optimizer.zero_grad(set_to_none=True)
    model.training_step(batches[0], 1).backward()
    c1 = model.dino_loss_func.center.clone()
    # second pass of the accumulation sees the centre the first one left: reproduce that for the separate reference of pass 2
    tr.optimizer.zero_grad(set_to_none=True)
    model.dino_loss_func.center.copy_(c1)
    model.training_step(batches[1], 1).backward()
    g1_after0 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    both = grads_of(batches)
    assert set(both) == set(g0) == set(g1_after0)
    worst = (-1.0, "")
    for n in both:
        want = g0[n].double() + g1_after0[n].double()
        err = float((both[n].double() - want).norm() / (want.norm() + 1e-30))
        worst = max(worst, (err, n))
    assert worst[0] <= 2e-3, worst     # (fp32 accumulation into the slab vs a sum of two fp32 results; bf16 never re-rounds a gradient)
    assert any(float((g1[n] - g1_after0[n]).abs().max()) > 0 for n in g1)   # (the centre does matter: the test is not vacuous)


def test_partially_frozen_backbone_and_head():
    """requires_grad = False on some backbone / head parameters (a frozen patch embedding, a frozen first block, a frozen projector
    bias): they get no gradient and the optimiser leaves them alone; every other gradient is what the unfrozen model computes."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    D, PR = 192, 4096
    crops, labels, ncl = one_channel_collate_fn(P.make_images([2, 1, 3], [224, 224, 96], seed=47))
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
    frozen = ("backbone.token_learner.proj.weight", "backbone.pos_embed", "backbone.blocks.0.linear1.weight", "backbone.blocks.0.norm1.bias",
              "backbone.blocks.5.self_attn.in_proj_weight", "head.mlp.2.bias")

    def run(freeze):
        m = DINO(_cfg(D, PR, 2, 1, lr=1e-3))
        m.load_state_dict(build_sd(D, PR))
        m = m.to(dev)
        named = dict(m.named_parameters())
        if freeze:
            for n in frozen:
                named[n].requires_grad_(False)
        tr = Trainer(max_epochs=10, steps_per_epoch=10)
        tr.current_epoch = 1
        tr.attach(m)
        m.current_epoch = 1
        m.on_train_epoch_start()
        before = {n: named[n].detach().clone() for n in frozen}
        m.training_step(batch, 1).backward()
        m.on_after_backward()
        grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}
        tr.optimizer.step()
        return grads, before, {n: named[n].detach().clone() for n in frozen}

    g_all, _, _ = run(False)
    g_frz, before, after = run(True)
    for n in frozen:
        assert g_frz[n] is None, n
        assert torch.equal(before[n], after[n]), n
    for n, gr in g_all.items():
        if n in frozen or gr is None:
            continue
        assert g_frz[n] is not None and torch.equal(g_frz[n], gr), n


@pytest.mark.parametrize("workload", ["cfg2", "cfg5"])
def test_bench_scale_step_is_deterministic(workload):
    """bench.py's own workload (cfg2: 1 024 images, 1 206 272 token rows per global pass; cfg5: Base, fp8 path, side streams on) built
    three times from the same seeds with the allocator's free blocks refilled in between: the loss and every gradient tensor of the
    first training step bit-identical -- every kernel of the step at the size the benchmark times it (scratch/r3/fuzz_bench_scale.py
    as a test; the teacher pass's QKV race of DESIGN section 0 made this differ in every run)."""
    import gc
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    dev = _dev()
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    wl = dict(bench.WORKLOADS[workload])
    ref = None
    for val in (None, 3.0e4, float("nan")):
        if val is not None:
            bufs = [torch.full((mb * 1024 * 1024 // 2,), val, device=dev, dtype=torch.bfloat16) for mb in (4096, 2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 2, 1) for _ in range(2)]
            torch.cuda.synchronize()
            del bufs
        model, tr, _, batch, nch, _ = bench.build_workload(wl, args, 0, 1, dev)
        model.current_epoch = tr.current_epoch
        model.on_train_epoch_start()
        loss = model.training_step(batch, 0)
        loss.backward()
        model.on_after_backward()
        torch.cuda.synchronize()
        got = (loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        if ref is None:
            ref = got
        else:
            assert torch.equal(got[0], ref[0]), (float(got[0]), float(ref[0]), val)
            bad = [n for n, t in got[1].items() if not torch.equal(t, ref[1][n])]
            assert not bad, (val, len(bad), bad[:5])
        del model, tr, batch, got
        gc.collect()
        torch.cuda.empty_cache()


def test_training_step_with_batchnorm_in_the_head_vs_golden_and_oracle():
    """`method_kwargs.use_bn_in_head = True` (reference src/methods/dino.py:59-77: BatchNorm1d behind the first two Linears of both
    heads): loss, gradients (incl. the BatchNorm scale / shift), the running estimates of both heads after one update per global
    crop, post-AdamW / EMA values -- against the golden written by the reference and against the oracle's tensors.  The head is
    called once per crop (BatchNorm statistics are per call), the heads stay in training mode as Lightning keeps them."""
    from chadavit_amd import ops
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, "step_tiny_bn_head.npz"))
    D, PR = int(g["D"]), int(g["P"])
    nch = [int(c) for c in g["nch"]]
    sizes = [int(s) for s in g["sizes"]]
    n_large, epoch = int(g["n_large"]), int(g["epoch"])
    sd = build_sd(D, PR, use_bn=True)
    model = DINO(_cfg(D, PR, n_large, len(sizes) - n_large, lr=float(g["lr"]), wd=float(g["wd"]), base_tau=float(g["base_tau"]),
                      use_bn_in_head=True))
    assert set(model.state_dict().keys()) >= set(sd.keys())   # the reference's key layout incl. mlp.{1,4}.running_* buffers
    model.load_state_dict(sd)
    model = model.to(dev)
    crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, sizes, seed=7))
    crops = crops if isinstance(crops, list) else [crops]
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
    tr = Trainer(max_epochs=10, steps_per_epoch=int(g["max_steps"]) // 10)
    tr.current_epoch = epoch
    tr.attach(model)
    model.current_epoch = epoch
    model.on_train_epoch_start()
    with ops.LaunchProfiler() as prof:
        loss = model.training_step(batch, 1)
        loss.backward()
        model.on_after_backward()
    summ = prof.summary()
    assert sum(v["launches"] for k, v in summ.items() if k[0] == "bn_stats") == 2 * 2 * n_large   # two BatchNorms, two heads, per crop
    assert sum(v["launches"] for k, v in summ.items() if k[0] == "bn_bwd") == 2 * n_large
    assert abs(loss.item() - float(g["loss"])) <= 2e-2, (loss.item(), float(g["loss"]))
    loss_o, grads_o, newc_o, aux = R.training_step(sd, crops, ncl, n_large, float(g["teacher_temp"]), freeze_last_layer=epoch < 1)
    named = dict(model.named_parameters())
    from tests.golden_util import step_outputs_vs_golden, step_outputs_vs_oracle
    torch.cuda.synchronize()
    # per-pass outputs against the reference and the oracle.  Features at the usual bar; the LOGITS behind two BatchNorms over 16 rows
    # at cosine >= 0.99 / rel-L2 <= 0.15 (measured 0.9968 / 8.0e-2): BatchNorm divides a column's batch deviation, a small fraction of the
    # features themselves for CLS features this similar, so their bf16 noise comes out several times larger
    print("bn head, outputs vs reference:", step_outputs_vs_golden(model._last_outs, g, logit_cos_min=0.99, logit_rel_max=0.15))
    print("bn head, outputs vs oracle:", step_outputs_vs_oracle(model._last_outs, aux, logit_cos_min=0.99, logit_rel_max=0.15))
    for n in set(str(n) for n in g["none_grad_names"]):
        assert named[n].grad is None, n
    tot_h = tot_r = 0.0
    worst = (1.0, None)
    big = float(max(g["grad_norms"]))
    for n, gn in zip(g["grad_names"], g["grad_norms"]):
        n = str(n)
        gh = named[n].grad
        assert gh is not None, n
        tot_h += gh.double().norm().item() ** 2
        tot_r += float(gn) ** 2
        if float(gn) >= 1e-4 * big:   # (BatchNorm makes a few gradients zero up to rounding, e.g. backbone.norm.bias: not compared)
            c = _cos(gh, grads_o[n])
            if c < worst[0]:
                worst = (c, n)
    # bars: BatchNorm over the 16 rows of a crop amplifies the bf16 noise of the features it normalises (measured: global norm within
    # 1 %, lowest cosine 0.944 on blocks.11.norm1.bias, the head's own tensors >= 0.989)
    assert abs(np.sqrt(tot_h) - np.sqrt(tot_r)) <= 5e-2 * np.sqrt(tot_r), (np.sqrt(tot_h), np.sqrt(tot_r))
    assert worst[0] >= 0.93, worst
    for key, bar in (("backbone.cls_token", 0.95), ("head.mlp.1.weight", 0.985), ("head.mlp.4.bias", 0.985), ("head.mlp.6.bias", 0.985)):
        assert _cos(named[key].grad, torch.from_numpy(g["grad::" + key])) >= bar, key   # (cls_token: one D-vector, measured 0.969)
    # running estimates of the four BatchNorms (bf16 Linear outputs under fp32 statistics)
    bufs = dict(model.named_buffers())
    for key in g.files:
        if key.startswith("bn::"):
            np.testing.assert_allclose(bufs[key[4:]].float().cpu().numpy(), g[key], rtol=3e-2, atol=8e-3, err_msg=key)
    assert int(bufs["head.mlp.1.num_batches_tracked"]) == n_large and int(bufs["momentum_head.mlp.4.num_batches_tracked"]) == n_large
    np.testing.assert_allclose(model.dino_loss_func.center[0, :256].float().cpu().numpy(), g["center_new"], atol=2e-3)
    tr.optimizer.step()
    tr.global_step += 1
    model.optimizer_zero_grad(epoch, 1, tr.optimizer)
    model.on_train_batch_end(None, batch, 1)
    assert abs(model.momentum_updater.cur_tau - float(g["tau_next"])) < 1e-12
    post = dict(zip([str(n) for n in g["post_names"]], g["post_sums"]))
    named = dict(model.named_parameters())
    for n in ("momentum_head.mlp.3.weight", "momentum_head.mlp.1.weight", "momentum_head.mlp.4.bias", "momentum_head.last_layer.weight_v"):
        v = named[n].double().sum().item()
        assert abs(v - post[n]) <= 1e-4 * (abs(post[n]) + named[n].numel() ** 0.5), n
    # eval mode (validation): the running estimates are used, nothing is updated
    model.eval()
    before = bufs["head.mlp.1.running_mean"].clone()
    with torch.no_grad():
        z = model.head(torch.randn(8, D, device=dev))
    assert torch.isfinite(z).all() and torch.equal(before, bufs["head.mlp.1.running_mean"])


@pytest.mark.parametrize("name,R_,rows,weight_dtype", [("step_tiny_multicrop", 170, 600780, "bf16"), ("step_small_mixed", 70, 274820, "bf16"),
                                                       ("step_base_c10", 25, 127500, "bf16"),   # (23 s; back in the default run since round 6)
                                                       ("step_base_c10", 25, 127500, "fp8"),
                                                       # 2.35 x the bench's rows: activations past 32-bit element and byte offsets
                                                       pytest.param("step_tiny_multicrop", 400, 1413600, "bf16", marks=pytest.mark.slow)])
def test_bench_scale_replicated_batch_vs_golden(name, R_, rows, weight_dtype):
    """The reference golden at BENCH SCALE through a size-independent property: a batch made of R copies of the golden's
    images has the same DINO loss (a mean over images; the centre starts at zero), the same centre update and the same
    gradients (means again) as the golden's own 3-image batch -- but its global-crop pass is 600 780 token rows, i.e. the
    launch shapes, work lists and kernel dispatch of bench.py's default run (603 136 rows), not the small-M ones the
    other step goldens exercise.  Likewise Small at cfg3's scale (274 820 rows; bench: ~278 k) and Base at cfg5's (127 500 rows;
    bench: 125 504), the latter also on the fp8 weight path with that path's tolerances (loss abs <= 5e-2)."""
    from chadavit_amd import ops
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    D, PR = int(g["D"]), int(g["P"])
    nch = [int(c) for c in g["nch"]]
    sizes = [int(s) for s in g["sizes"]]
    n_large, epoch = int(g["n_large"]), int(g["epoch"])
    sd = build_sd(D, PR)
    cfg = _cfg(D, PR, n_large, len(sizes) - n_large, clip_grad=float(g["clip_grad"]), lr=float(g["lr"]), wd=float(g["wd"]),
               base_tau=float(g["base_tau"]))
    fp8 = weight_dtype == "fp8"
    if fp8:
        cfg.backbone.kwargs.weight_dtype = "fp8"
    model = DINO(cfg)
    model.load_state_dict(sd)
    model = model.to(dev)
    imgs = P.make_images(nch, sizes, seed=7)
    crops_s, labels_s, ncl_s = one_channel_collate_fn(imgs)
    crops, labels, ncl = one_channel_collate_fn(list(imgs) * R_)
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
    assert rows == sum(1 + c * 196 for c in nch) * n_large * R_
    tr = Trainer(max_epochs=10, steps_per_epoch=int(g["max_steps"]) // 10)
    tr.current_epoch = epoch
    tr.attach(model)
    model.current_epoch = epoch
    model.on_train_epoch_start()
    with ops.LaunchProfiler() as prof:
        loss = model.training_step(batch, 1)
        loss.backward()
        model.on_after_backward()
    summ = prof.summary()
    # the bench's dispatch at the bench's row count: the whole-block kernel in all three passes, the fused FFN backward
    if D <= 384:
        n_pass = 3 if len(sizes) > n_large else 2
        # (11 of the 12 blocks: the last one runs on the CLS rows, ChAdaViT.cls_only_last_block)
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "proj_ffn_ln_fwd" and k[1] == rows) == 22, list(summ.keys())
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "proj_ffn_ln_fwd") == 11 * n_pass
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "ffn_bwd_dx" and k[1] == rows) == 11
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "attn_cls_fwd" and k[1] == rows) == 2
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "attn_cls_bwd" and k[1] == rows) == 1
    from tests.golden_util import grad_subsets_vs_golden, step_outputs_vs_golden, step_outputs_vs_oracle
    torch.cuda.synchronize()
    if fp8:
        # forward: four per block and pass, the last block QKV only; backward: the FFN's two dX GEMMs of the 11 full-width blocks
        assert sum(v["launches"] for k, v in summ.items() if k[0] == "gemm_nt_mx8" and k[1] == rows) == \
            (4 * 11 + 1) * 2 + (2 * 11 if model.backbone.fp8_dx else 0)
        assert abs(loss.item() - float(g["loss"])) <= 5e-2, (loss.item(), float(g["loss"]))
        _fp8_gradient_bar(dict(model.named_parameters()), g, f"fp8 {name} x {R_}")
        # the fp8 weight path's output bar (CLS cosine >= 0.99), every copy of every image, student and teacher
        print(name, "fp8 outputs vs reference:", step_outputs_vs_golden(model._last_outs, g, copies=R_, cos_min=0.99, rel_max=0.15))
        return
    assert abs(loss.item() - float(g["loss"])) <= 2e-2, (loss.item(), float(g["loss"]))
    loss_o, grads_o, newc_o, aux = R.training_step(sd, crops_s, ncl_s, n_large, float(g["teacher_temp"]), freeze_last_layer=epoch < 1,
                                                   clip_grad=float(g["clip_grad"]))
    named = dict(model.named_parameters())
    # what the three passes produced at the bench's row counts, EVERY copy of every image: student CLS features of all crops (local
    # pass included), teacher CLS features, both logits -- against the reference's golden and the oracle (a wrong row in 1-5 % of the
    # rows of the teacher / local pass, round 3's postlogue bug, fails the per-row bar here)
    print(name, R_, "outputs vs reference:", step_outputs_vs_golden(model._last_outs, g, copies=R_))
    print(name, R_, "outputs vs oracle:", step_outputs_vs_oracle(model._last_outs, aux, copies=R_))
    print(name, R_, "gradient spread vs reference:", grad_subsets_vs_golden(named, g))
    for n in set(str(n) for n in g["none_grad_names"]):
        assert named[n].grad is None, n
    tot_h = tot_r = 0.0
    worst = (1.0, None)
    for n, gn in zip(g["grad_names"], g["grad_norms"]):
        n = str(n)
        gh = named[n].grad
        assert gh is not None, n
        tot_h += gh.double().norm().item() ** 2
        tot_r += float(gn) ** 2
        go = grads_o[n]
        if float(gn) > 1e-4 * np.sqrt(go.numel()) * 1e-2:
            c = _cos(gh, go)
            if c < worst[0]:
                worst = (c, n)
    assert abs(np.sqrt(tot_h) - np.sqrt(tot_r)) <= 5e-2 * np.sqrt(tot_r), (np.sqrt(tot_h), np.sqrt(tot_r))
    assert worst[0] >= 0.99, worst
    np.testing.assert_allclose(model.dino_loss_func.center[0, :256].float().cpu().numpy(), g["center_new"], atol=2e-3)


def test_validation_step_vs_golden():
    """validation_step / on_validation_epoch_end (dino.py:327-365; base.py:753-899, 1278-1436) on the HIP path against the
    reference's outputs for both cfg.ssl_val_loss settings: CLS features cosine >= 0.999, probe logits / z rel-L2 <= 3e-2,
    dino_loss_val abs <= 2e-2, centre abs 2e-3 (bf16 teacher logits), batch_size and output-list bookkeeping exact."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, "val_tiny.npz"))
    D, PR, nl = int(g["D"]), int(g["P"]), int(g["n_large"])
    nch, sizes = [int(c) for c in g["nch"]], [int(s) for s in g["sizes"]]
    crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, sizes, seed=7))
    for ssl in (True, False):
        cfg = _cfg(D, PR, nl, len(sizes) - nl)
        cfg.ssl_val_loss = ssl
        cfg.knn_eval = {"enabled": True, "k": 2, "distance_func": "cosine"}
        model = DINO(cfg)
        model.load_state_dict(build_sd(D, PR))
        model = model.to(dev)
        Trainer(max_epochs=10, steps_per_epoch=10).attach(model)
        model.current_epoch = 1
        model.on_train_epoch_start()
        tag = "ssl" if ssl else "plain"
        if ssl:
            outs = model.validation_step(([c.to(dev) for c in crops], labels.to(dev), ncl), 0)
            assert abs(outs["dino_loss_val"].item() - float(g["ssl::dino_loss_val"])) <= 2e-2
            z, mz = torch.cat(outs["z"]), torch.cat(outs["momentum_z"])
            assert _rel(mz[:, :64], torch.from_numpy(g["ssl::momentum_z"])) <= 3e-2
            logits = torch.cat(outs["logits"])
            assert "feats" not in outs  # consumed by the online k-NN (knn_eval on)
        else:
            outs = model.validation_step((crops[0].to(dev), labels.to(dev), [ncl[0]]), 0)
            z, logits = outs["z"], outs["logits"]
        assert _rel(z[:, :64], torch.from_numpy(g[f"{tag}::z"])) <= 3e-2
        assert _rel(logits, torch.from_numpy(g[f"{tag}::logits"])) <= 3e-2
        assert outs["batch_size"] == int(g[f"{tag}::batch_size"])
        assert len(model.validation_step_outputs) == int(g[f"{tag}::n_outputs"]) == 1
        np.testing.assert_allclose(model.dino_loss_func.center[0, :256].cpu().numpy(), g[f"{tag}::center"], atol=2e-3)
        # the test features the k-NN received are the CLS features of the golden
        feats = torch.cat(model.knn.test_features)
        assert _cos(feats, torch.from_numpy(g[f"{tag}::feats"])) >= 0.999
        model.knn.update(train_features=feats, train_targets=torch.cat(model.knn.test_targets))
        model.on_validation_epoch_end()
        assert len(model.validation_step_outputs) == 0 and model._logged["val_knn_acc1"] == 100.0


def test_two_steps_run_and_loss_moves():
    """Smoke for the full hook loop with the scheduler on (2 global + 2 local crops, mixed channels)."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    cfg = _cfg(192, 4096, 2, 2)
    cfg.scheduler.name = "warmup_cosine"
    model = DINO(cfg).to(dev)
    imgs = P.make_images([2, 1, 3, 1], [224, 224, 96, 96], seed=3)
    crops, labels, ncl = one_channel_collate_fn(imgs)
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
    tr = Trainer(max_epochs=2, steps_per_epoch=4).attach(model)
    losses = [tr.train_step(batch, i).item() for i in range(3)]
    assert all(np.isfinite(losses)), losses
    assert abs(losses[0] - np.log(4096)) < 1.0  # random init: close to the uniform-prediction loss ln(P)
    sd = model.state_dict()
    assert "dino_loss_func.center" in sd and "momentum_head.last_layer.weight_v" in sd and "backbone.blocks.11.norm2.bias" in sd


@pytest.mark.parametrize("name,mode", [("traj_tiny_c1", "eager"), ("traj_tiny_c1", "graph"), ("traj_tiny_mixed_multicrop", "eager"),
                                       ("traj_tiny_mixed_multicrop", "graph")])
def test_five_step_trajectory_vs_golden(name, mode):
    """Round 6: state carried between steps.  Golden traj_tiny_c1 = five consecutive steps of the unmodified reference (BASELINE configs[0]'s
    shape: Tiny, four one-channel images, two global crops; a new batch every step; the epoch boundary after step 3 moves the teacher
    temperature and thaws the last layer).  The HIP path through the hook loop (Trainer.train_step) and through the whole-step hipGraph
    (GraphedTrainStep): per step the loss (abs <= 2e-2), the centre that the next step's loss reads (abs <= 2e-3; losses/dino.py:103-118), tau
    to 1e-12 (base.py:1270-1273), the teacher's logits (row sums of squares) and the EMA teacher's parameter sums (rel <= 1e-4;
    momentum.py:63-87), the student's parameters after AdamW's moments have run on for five steps."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.graphed import GraphedTrainStep
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    from tests.golden_util import traj_channels
    g = np.load(os.path.join(GOLDEN, name + ".npz"))   # (traj_tiny_mixed_multicrop: another 1-10 channel mix every step, two global + two local crops)
    D, PR, n_large = int(g["D"]), int(g["P"]), int(g["n_large"])
    nch_of, sizes = traj_channels(g), [int(s) for s in g["sizes"]]
    spe = int(g["steps_per_epoch"])
    model = DINO(_cfg(D, PR, n_large, len(sizes) - n_large, lr=float(g["lr"]), wd=float(g["wd"]), base_tau=float(g["base_tau"])))
    model.load_state_dict(build_sd(D, PR))
    model = model.to(dev)
    tr = Trainer(max_epochs=10, steps_per_epoch=spe).attach(model)
    tr.estimated_stepping_batches = int(g["max_steps"])
    step = GraphedTrainStep(tr) if mode == "graph" else tr.train_step
    names = [str(n) for n in g["param_names"]]
    worst = {"loss": 0.0, "center": 0.0, "student_sq_rel": 0.0, "tz_sq_rel": 0.0}
    bad = []
    for k in range(int(g["steps"])):
        crops, labels, ncl = one_channel_collate_fn(P.make_images(nch_of(k), sizes, seed=7 + k))
        batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
        tr.current_epoch = k // spe
        tau_used = model.momentum_updater.cur_tau
        loss = float(step(batch, k % spe).item())
        torch.cuda.synchronize()
        assert abs(tau_used - float(g["tau_used"][k])) < 1e-12 and abs(model.momentum_updater.cur_tau - float(g["tau_next"][k])) < 1e-12, k
        assert abs(model.dino_loss_func.teacher_temp_schedule[k // spe] - float(g["teacher_temp"][k])) < 1e-12
        worst["loss"] = max(worst["loss"], abs(loss - float(g["loss"][k])))
        assert abs(loss - float(g["loss"][k])) <= 2e-2, (mode, k, loss, float(g["loss"][k]))
        c = model.dino_loss_func.center.float().cpu()
        worst["center"] = max(worst["center"], float(np.abs(c[0, :256].numpy() - g["center"][k]).max()))
        np.testing.assert_allclose(c[0, :256].numpy(), g["center"][k], atol=2e-3, rtol=0)
        assert abs(float(c.double().sum()) - float(g["center_sum"][k])) <= 2e-3 * PR ** 0.5, k
        named = dict(model.named_parameters())
        lr_, steps_done = float(g["lr"]), k + 1
        for i, n in enumerate(names):
            t, st = named["momentum_" + n].detach().double(), named[n].detach().double()
            ref_t, ref_tq, ref_sq, ref_s = float(g["teacher_sums"][k][i]), float(g["teacher_sq"][k][i]), float(g["student_sq"][k][i]), float(g["student_sums"][k][i])
            # AdamW moves EVERY entry by ~lr per step whatever its gradient's size, so an entry whose gradient is rounding noise under bf16 may go
            # the other way: 2 lr per step and entry, signs independent -> a tensor's SUM differs by a random walk of sqrt(N) such steps (3 sigma);
            # the EMA teacher inherits (1 - tau) of the student's deviation per step (tau >= 0.99 here).  A wrong tau (by 1e-4), a skipped update or
            # a stale shadow moves these sums by far more: |teacher - student| ~ the weights themselves.
            walk = 3.0 * t.numel() ** 0.5 * 2.0 * lr_ * steps_done
            d_t, d_s = abs(float(t.sum()) - ref_t), abs(float(st.sum()) - ref_s)
            worst["teacher_sum_over_bound"] = max(worst.get("teacher_sum_over_bound", 0.0), d_t / (0.01 * steps_done * walk + 1e-5 * abs(ref_t) + 1e-6))
            worst["student_sum_over_bound"] = max(worst.get("student_sum_over_bound", 0.0), d_s / (walk + 1e-5 * abs(ref_s) + 1e-6))
            if d_t > 0.01 * steps_done * walk + 1e-5 * abs(ref_t) + 1e-6:
                bad.append((mode, k, n, "teacher sum", float(t.sum()), ref_t))
            if d_s > walk + 1e-5 * abs(ref_s) + 1e-6:
                bad.append((mode, k, n, "student sum", float(st.sum()), ref_s))
            relt = abs(float((t ** 2).sum()) - ref_tq) / ref_tq
            relq = abs(float((st ** 2).sum()) - ref_sq) / ref_sq
            worst["teacher_sq_rel"] = max(worst.get("teacher_sq_rel", 0.0), relt)
            worst["student_sq_rel"] = max(worst["student_sq_rel"], relq)
            if relt > 1e-4:
                bad.append((mode, k, n, "teacher sum of squares", relt))
            if relq > 2e-3:
                bad.append((mode, k, n, "student sum of squares", relq))
        if mode == "eager":   # what the teacher pass produced THIS step with the EMA weights of the step before
            tz = model._last_outs["momentum_z"].detach().double().cpu()
            rq = torch.from_numpy(g["momentum_z_rowsq"][k])
            relz = float((((tz ** 2).sum(1) - rq).abs() / rq).max())
            worst["tz_sq_rel"] = max(worst["tz_sq_rel"], relz)
            assert relz <= 8e-2, (k, relz)
            assert bool(((tz.sum(1) - torch.from_numpy(g["momentum_z_rowsum"][k])).abs() <= 0.16 * rq.sqrt()).all()), k
    print("trajectory", mode, worst)
    assert not bad, bad[:8]
    if mode == "graph":
        assert len(step.graphs) == (int(g["steps"]) if "nch_per_step" in g.files and int(g["nch_per_step"]) else 2)   # one per batch signature and frozen / thawed state
        step.close()
    named = dict(model.named_parameters())
    # element-wise on a small tensor: an entry of the student whose gradient's sign is bf16 noise sits up to 2 lr k off after step k, and the EMA
    # teacher collects (1 - tau) of that per step: <= 0.01 * 2 lr * (1 + 2 + .. + 5) = 1.5e-4 for such an entry (3 of 192 at 5e-5 .. 9e-5 observed),
    # everything else to fp32 round-off
    dt = np.abs(named["momentum_backbone.norm.weight"].detach().float().cpu().numpy() - g["post::momentum_backbone.norm.weight"])
    n_steps = int(g["steps"])
    assert dt.max() <= 0.01 * 2.0 * float(g["lr"]) * n_steps * (n_steps + 1) / 2 + 1e-5 and (dt > 2e-5).mean() <= 0.05, (dt.max(), (dt > 2e-5).mean())
    d = np.abs(named["backbone.norm.weight"].detach().float().cpu().numpy() - g["post::backbone.norm.weight"])
    assert d.max() <= 2.1 * int(g["steps"]) * float(g["lr"]) and (d > 5e-4).mean() <= 0.10, (d.max(), (d > 5e-4).mean())


def test_training_memory_is_stable_across_steps():
    """Side-stream overlap must not make the caching allocator grow: after warm-up no new device segments appear
    (guards against record_stream-style deferred frees that made the reserved pool grow every step)."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    model = DINO(_cfg(192, 4096, 2, 2)).to(dev)
    imgs = P.make_images([3] * 8, [224, 224, 96, 96], seed=9)
    crops, labels, ncl = one_channel_collate_fn(imgs)
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
    tr = Trainer(max_epochs=2, steps_per_epoch=10).attach(model)
    for i in range(4):
        tr.train_step(batch, i)
    torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()["segment.all.allocated"]
    for i in range(4):
        tr.train_step(batch, 4 + i)
    torch.cuda.synchronize()
    s1 = torch.cuda.memory_stats()["segment.all.allocated"]
    assert s1 - s0 <= 2, (s0, s1)


def test_loss_decreases_when_overfitting_one_batch():
    """Full steps (all hooks, AdamW, EMA, centre) on one fixed mixed-channel batch: the DINO loss must go down over the first
    steps and stay finite -- an end-to-end check that forward, backward and the optimiser agree in sign and scale.
    (Run long enough at this learning rate DINO collapses to the uniform solution, loss -> ln P; only the descent is asserted.)"""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    torch.manual_seed(0)
    cfg = _cfg(192, 4096, 2, 0, lr=2e-3, base_tau=0.99)
    model = DINO(cfg).to(dev)
    imgs = P.make_images([3, 1, 2, 5, 1, 3, 2, 4], [224, 224], seed=21)
    crops, labels, ncl = one_channel_collate_fn(imgs)
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
    tr = Trainer(max_epochs=40, steps_per_epoch=1).attach(model)
    losses = []
    for i in range(12):
        tr.current_epoch = 1  # past the prototype freeze, teacher temperature on its schedule
        losses.append(tr.train_step(batch, 1).item())
    assert all(np.isfinite(losses)), losses
    assert min(losses[2:8]) < losses[0] - 1.0, losses


def test_lars_with_wd_split_through_the_hooks():
    """optimizer.name=lars + exclude_bias_n_norm_wd (base.py:416-443, lars.py:112-167): after two full steps every
    parameter equals the oracle's LARS applied to the gradients the HIP backward produced."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    cfg = _cfg(192, 4096, 2, 0, lr=0.3, wd=1e-2)
    cfg.optimizer.name = "lars"
    cfg.optimizer.exclude_bias_n_norm_wd = True
    cfg.optimizer.kwargs = {"momentum": 0.9, "eta": 0.02, "exclude_bias_n_norm": True, "clip_lr": True}
    model = DINO(cfg)
    model.load_state_dict(build_sd(192, 4096))
    model = model.to(dev)
    crops, labels, ncl = one_channel_collate_fn(P.make_images([2, 1, 3, 1], [224, 224], seed=5))
    crops = crops if isinstance(crops, list) else [crops]
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
    tr = Trainer(max_epochs=10, steps_per_epoch=4).attach(model)
    names = [g["name"] for g in tr.optimizer.param_groups]
    assert names == ["backbone", "backbone_no_decay", "classifier", "classifier_no_decay", "head", "head_no_decay"], names
    assert [g["weight_decay"] for g in tr.optimizer.param_groups] == [1e-2, 0, 0, 0, 1e-2, 0]
    wd_of = {id(p): g["weight_decay"] for g in tr.optimizer.param_groups for p in g["params"]}
    expect = {n: p.detach().float().cpu().clone() for n, p in model.named_parameters() if not n.startswith("momentum_")}
    bufs = {}
    tr.current_epoch = 1  # prototypes unfrozen
    model.current_epoch = 1
    model.on_train_epoch_start()
    for step in range(2):
        loss = model.training_step(batch, step)
        loss.backward()
        model.on_after_backward()
        for n, p in model.named_parameters():
            if p.grad is None or n.startswith("momentum_"):
                continue
            new_p, bufs[n] = R.lars_step(expect[n], p.grad.detach().float().cpu(), bufs.get(n), lr=0.3, momentum=0.9, dampening=0.0,
                                         weight_decay=wd_of[id(p)], nesterov=False, eta=0.02, eps=1e-8, clip_lr=True,
                                         exclude_bias_n_norm=True)
            expect[n] = new_p
        tr.optimizer.step()
        tr.global_step += 1
        model.optimizer_zero_grad(1, step, tr.optimizer)
        model.on_train_batch_end(None, batch, step)
    moved = 0
    for n, p in model.named_parameters():
        if n.startswith("momentum_") or n.startswith("classifier"):
            continue
        got = p.detach().float().cpu()
        np.testing.assert_allclose(got.numpy(), expect[n].numpy(), rtol=2e-5, atol=2e-6, err_msg=n)
        moved += int(not torch.equal(got, build_sd(192, 4096)[n].float())) if n in ("backbone.norm.weight", "head.mlp.0.weight") else 0
    assert moved == 2
    # the bf16 shadows the next forward reads were refreshed from the updated slab
    f = model.backbone.flat_params()
    f.refresh()
    w = f.w("blocks.0.linear1.weight")
    assert torch.equal(w.float(), dict(model.named_parameters())["backbone.blocks.0.linear1.weight"].detach().to(torch.bfloat16).float())


def test_fresh_models_back_to_back_are_deterministic():
    """Several models built and stepped one after another in one process give the same finite losses.  Guards the
    stream/allocator discipline: the teacher pass runs on a side HIP stream, and a parameter slab built lazily there used to
    read parameter storage the caching allocator had already handed back to the main stream (intermittent NaN teachers)."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    imgs = P.make_images([3, 1, 2, 5, 1, 3, 2, 4], [224, 224], seed=21)
    crops, labels, ncl = one_channel_collate_fn(imgs)
    seen = set()
    for rep in range(8):
        torch.manual_seed(0)
        model = DINO(_cfg(192, 4096, 2, 0, lr=2e-3, base_tau=0.99)).to(dev)
        batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
        tr = Trainer(max_epochs=40, steps_per_epoch=1).attach(model)
        tr.current_epoch = 1
        ls = [tr.train_step(batch, 1) for _ in range(2)]
        vals = tuple(round(v.item(), 4) for v in ls)
        assert all(np.isfinite(vals)), (rep, vals)
        seen.add(vals)
        del model, tr
    assert len(seen) == 1, seen


@pytest.mark.parametrize("name", ["attnmap_tiny", "attnmap_tiny96"])
def test_get_last_selfattention_vs_golden_and_oracle(name):
    """Attention-map export (chada_vit.py:313-320): HIP path vs the reference's golden rows and the full oracle tensor.
    Tolerance: probabilities computed from bf16 activations of 11 blocks -- abs 3e-3 on entries (row sums to 1 within 1e-5),
    cosine of the CLS rows the consumer plots (main_attn.py:207) >= 0.999."""
    dev = _dev()
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    D, B, S = int(g["D"]), int(g["B"]), int(g["S"])
    m = _backbone(D, int(g["seed_w"]), dev)
    imgs = P.make_images([1] * B, [S], seed=int(g["seed_x"]))
    x = torch.stack([c[0] for c, _ in imgs])
    att = m.get_last_selfattention(x.to(dev))
    assert list(att.shape) == [int(v) for v in g["shape"]] and att.dtype == torch.float32
    assert float((att.sum(-1) - 1).abs().max()) <= 1e-5
    ref_full = R.last_selfattention(P.fill_state_dict(P.backbone_shapes(D), seed=int(g["seed_w"])), x)
    assert float((att.cpu() - ref_full).abs().max()) <= 3e-3
    cls = att[:, :, 0, :].cpu()
    assert float((cls - torch.from_numpy(g["cls_rows"])).abs().max()) <= 3e-3
    for b in range(B):
        for h in range(att.shape[1]):
            assert _cos(cls[b, h, 1:], torch.from_numpy(g["cls_rows"])[b, h, 1:]) >= 0.999
    sel = att[:, :, torch.from_numpy(g["rows"]).to(dev), :].cpu()
    assert float((sel - torch.from_numpy(g["sel_rows"])).abs().max()) <= 3e-3


def test_attn_probs_ragged_matches_softmax():
    """ops.attn_probs on a ragged batch (lengths 109, 589, 37) vs fp32 softmax of the same bf16 q, k."""
    from chadavit_amd import ops
    from chadavit_amd.ragged import RaggedBatch
    dev = _dev()
    for H, D in ((2, 192), (12, 192), (2, 768)):
        lens_c = [(3, 36), (3, 196), (1, 36)]
        outs = []
        for c, p in lens_c:
            rb = RaggedBatch([c], p, dev)
            qkv = (torch.randn((rb.T, 3 * D), generator=torch.Generator().manual_seed(rb.T + H)) * 0.7).to(dev).to(torch.bfloat16)
            pr = ops.attn_probs(qkv, rb.cu_seqlens, rb.lens, H)
            q, k, _ = qkv.float().split(D, -1)
            dh = D // H
            qh = q.view(rb.T, H, dh).transpose(0, 1)
            kh = k.view(rb.T, H, dh).transpose(0, 1)
            ref = torch.softmax(qh @ kh.transpose(1, 2) / dh ** 0.5, -1)
            assert pr.shape == (1, H, rb.T, rb.T)
            assert float((pr[0] - ref).abs().max()) <= 2e-6
        rb = RaggedBatch([3, 1, 2], 36, dev)
        qkv = torch.randn((rb.T, 3 * D), generator=torch.Generator().manual_seed(7)).to(dev).to(torch.bfloat16)
        prs = ops.attn_probs(qkv, rb.cu_seqlens, rb.lens, H)
        assert isinstance(prs, list) and [tuple(t.shape) for t in prs] == [(H, n, n) for n in rb.lens]
        for i, n in enumerate(rb.lens):
            s0 = int(rb.host_cu_seqlens[i])
            q, k, _ = qkv[s0:s0 + n].float().split(D, -1)
            dh = D // H
            ref = torch.softmax(q.view(n, H, dh).transpose(0, 1) @ k.view(n, H, dh).transpose(0, 1).transpose(1, 2) / dh ** 0.5, -1)
            assert float((prs[i] - ref).abs().max()) <= 2e-6


@pytest.mark.parametrize("num_heads", [None, 12])
def test_backward_through_all_tokens_output_vs_oracle(num_heads):
    """return_all_tokens=True (chada_vit.py:283-287) is differentiable on the HIP path: gradients of a linear functional of the
    patch-token output vs autograd through the oracle on the same weights / inputs (per-tensor cosine >= 0.99, global
    grad-norm within 5 %).  num_heads = 12 is the reference's DEFAULT constructor (chada_vit.py:138-139: 12 heads, dh = 16,
    final LayerNorm eps 1e-5) -- fine-tuning a notebook-constructed model runs the widened-head attention backward."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    dev = _dev()
    D = 192
    m = _backbone(D, 61, dev, return_all_tokens=True, num_heads=num_heads)
    imgs = P.make_images([2, 1, 3], [96], seed=62)
    crops, labels, ncl = one_channel_collate_fn(imgs)
    x = crops if isinstance(crops, torch.Tensor) else crops[0]
    nch = ncl[0] if isinstance(ncl[0], list) else ncl
    out = m(x.to(dev), 0, [nch])
    wgt = P.tensor(tuple(out.shape), "alltok.w", 1.0, seed=63)
    (out * wgt.to(dev)).sum().backward()
    sd = {k: v.clone().requires_grad_(True) for k, v in P.fill_state_dict(P.backbone_shapes(D), seed=61).items()}
    ref = R.backbone_ragged(sd, x, nch, return_all_tokens=True, **({} if num_heads is None else {"nheads": num_heads, "final_eps": 1e-5}))
    assert tuple(ref.shape) == tuple(out.shape)
    assert _cos(out.detach(), ref.detach()) >= 0.999
    (ref * wgt).sum().backward()
    tot_h = tot_r = 0.0
    worst = (1.0, None)
    for n, p in m.named_parameters():
        gr = sd[n].grad
        if gr is None:
            continue
        assert p.grad is not None, n
        tot_h += p.grad.double().norm().item() ** 2
        tot_r += gr.double().norm().item() ** 2
        if gr.norm() > 1e-6 * gr.numel() ** 0.5:
            c = _cos(p.grad, gr)
            if c < worst[0]:
                worst = (c, n)
    assert abs(tot_h ** 0.5 - tot_r ** 0.5) <= 5e-2 * tot_r ** 0.5, (tot_h ** 0.5, tot_r ** 0.5)
    assert worst[0] >= 0.99, worst


def test_pos_embed_span_is_handed_over_after_autograd_has_accumulated_into_it():
    """with crops of another size than img_size the position rows are bicubic-resized and their gradient reaches pos_embed
    through autograd AFTER the backbone's backward function has returned.  The gradient-span hook (what GradSync all-reduces) must see
    the span that holds pos_embed only once that accumulation has happened: every span handed to the hook is compared with the final
    gradient slab -- a span handed over early would miss the pos_embed term."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    dev = _dev()
    m = _backbone(192, 71, dev)
    imgs = P.make_images([2, 1, 3], [96], seed=72)
    crops, labels, ncl = one_channel_collate_fn(imgs)
    x = crops if isinstance(crops, torch.Tensor) else crops[0]
    nch = ncl[0] if isinstance(ncl[0], list) else ncl
    seen = []
    m.grad_ready_hook = lambda flat, b, e: seen.append((b, e, flat.grad[b:e].clone()))
    out = m(x.to(dev), 0, [nch])
    (out * P.tensor(tuple(out.shape), "pos.w", 1.0, seed=73).to(dev)).sum().backward()
    torch.cuda.synchronize()
    flat = m.flat_params()
    covered = torch.zeros(flat.numel, dtype=torch.bool)
    for b, e, snap in seen:
        assert torch.equal(snap, flat.grad[b:e]), (b, e)   # the span was final when it was handed over
        covered[b:e] = True
    assert bool(covered.all())                              # every span was handed over, the deferred one included
    pb, pe = flat.span(["pos_embed"])
    assert float(flat.grad[pb:pe][192:].abs().sum()) > 0    # (the patch rows of pos_embed do receive their gradient on this path)
    assert m._pos_span_deferred is None and not m._pos_autograd_pending


def test_graphed_backbone_replays_bit_exact_and_faster():
    """chadavit_amd.serving.GraphedBackbone: the hipGraph replay of the forward gives exactly the eager result, follows weight
    updates (refresh outside the graph), and costs less host+device time than ~110 eager launches at a small batch."""
    import time
    from chadavit_amd.serving import GraphedBackbone
    dev = _dev()
    m = _backbone(192, 71, dev)
    nch = [3, 1, 2, 5]
    imgs = P.make_images(nch, [224], seed=72)
    x = torch.cat([c[0] for c, _ in imgs]).unsqueeze(1).to(dev)
    with torch.no_grad():
        eager = m.forward_ragged(x, nch)
    gb = GraphedBackbone(m, nch, 224)
    out = gb(x)
    assert torch.equal(out, eager)
    x2 = torch.cat([c[0] for c, _ in P.make_images(nch, [224], seed=73)]).unsqueeze(1).to(dev)
    with torch.no_grad():
        assert torch.equal(gb(x2), m.forward_ragged(x2, nch))
    with torch.no_grad():   # weights change -> the next replay sees them
        for p in m.parameters():
            p.mul_(1.01)
        m.flat_params().mark_dirty()
        eager3 = m.forward_ragged(x, nch)
    assert torch.equal(gb(x), eager3) and not torch.equal(eager3, eager)

    def timeit(fn, n=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    with torch.no_grad():
        te = timeit(lambda: m.forward_ragged(x, nch))
    tg = timeit(lambda: gb(x))
    print(f"eager {te * 1e3:.2f} ms  graph {tg * 1e3:.2f} ms")   # informational: a correctness suite asserts no timings


@pytest.mark.parametrize("D,num_heads", [(192, None), (384, None), (192, 12)])
def test_cls_only_last_block_vs_full_block_and_oracle(D, num_heads):
    """ChAdaViT.cls_only_last_block (the last block on the CLS rows only when return_all_tokens = False) against the same model with
    the flag off -- CLS features and every gradient -- and against autograd through the oracle.  The 12-head default constructor
    (dh = 16) runs the CLS attention natively."""
    from chadavit_amd import ops
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    dev = _dev()
    imgs = P.make_images([2, 1, 3, 10], [224], seed=72)
    crops, labels, ncl = one_channel_collate_fn(imgs)
    x = (crops if isinstance(crops, torch.Tensor) else crops[0]).to(dev)
    nch = ncl[0] if isinstance(ncl[0], list) else ncl
    wgt = P.tensor((len(nch), D), "cls.w", 1.0, seed=73).to(dev)
    res = {}
    for flag in (True, False):
        m = _backbone(D, 71, dev, num_heads=num_heads)
        m.cls_only_last_block = flag
        with ops.LaunchProfiler() as prof:
            out = m(x, 0, [nch])
            (out * wgt).sum().backward()
        n_cls = sum(v["launches"] for k, v in prof.summary().items() if k[0] in ("attn_cls_fwd", "attn_cls_bwd"))
        assert n_cls == (2 if flag else 0)
        res[flag] = (out.detach().float(), {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None})
    (o1, g1), (o0, g0) = res[True], res[False]
    assert _cos(o1, o0) >= 0.9999 and float((o1 - o0).abs().max()) <= 1e-2 * float(o0.abs().max()) + 1e-2   # a bf16 ulp or two
    assert set(g1) == set(g0)
    for n in g0:
        if g0[n].norm() > 1e-6 * g0[n].numel() ** 0.5:
            assert _cos(g1[n], g0[n]) >= 0.995, n
    sd = {k: v.clone().requires_grad_(True) for k, v in P.fill_state_dict(P.backbone_shapes(D), seed=71).items()}
    ref = R.backbone_ragged(sd, x.cpu(), nch, **({} if num_heads is None else {"nheads": num_heads, "final_eps": 1e-5}))
    assert _cos(o1, ref.detach()) >= 0.999
    (ref * wgt.cpu()).sum().backward()
    tot_h = sum(v.double().norm().item() ** 2 for v in g1.values()) ** 0.5
    tot_r = sum(sd[n].grad.double().norm().item() ** 2 for n in g1) ** 0.5
    assert abs(tot_h - tot_r) <= 5e-2 * tot_r, (tot_h, tot_r)
    for n in g1:
        gr = sd[n].grad
        if gr.norm() > 1e-6 * gr.numel() ** 0.5:
            assert _cos(g1[n], gr) >= 0.99, n


def test_skipping_the_unused_local_pass_changes_nothing():
    """DINO.compute_unused_local_pass = False: the local crops' student pass (never read by the reference's step: SURVEY A7) is skipped;
    loss and every gradient are bit-identical to the default, which runs it."""
    from chadavit_amd import ops
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    dev = _dev()
    imgs = P.make_images([2, 1, 3], [224, 224, 96, 96], seed=81)
    crops, labels, ncl = one_channel_collate_fn(imgs)
    batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
    res = {}
    for flag in (True, False):
        model = DINO(_cfg(192, 4096, 2, 2))
        model.load_state_dict(build_sd(192, 4096))
        model = model.to(dev)
        model.compute_unused_local_pass = flag
        model.current_epoch = 1
        model.on_train_epoch_start()
        with ops.LaunchProfiler() as prof:
            loss = model.training_step(batch, 0)
            loss.backward()
        n_fwd = sum(v["launches"] for k, v in prof.summary().items() if k[0] == "attn_cls_fwd")
        assert n_fwd == (3 if flag else 2)   # student, teacher[, local] passes
        res[flag] = (loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        assert len(model._last_outs["feats"]) == (4 if flag else 2)
    assert res[True][0] == res[False][0]
    assert set(res[True][1]) == set(res[False][1])
    for n, g in res[True][1].items():
        assert torch.equal(g, res[False][1][n]), n


def test_side_streams_with_a_new_channel_mix_every_step_match_one_stream():
    """DINO.overlap_streams (teacher / local-crop passes on side HIP streams) with a NEW channel list every step: the ragged
    description of each crop batch is uploaded once and shared by the passes on different streams -- each consumer has to order
    itself behind that upload (RaggedBatch.use_on_current_stream), otherwise its tokenizer / attention kernels read index
    arrays that have not landed.  Loss, teacher logits and every gradient must equal the single-stream run, step after step."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd import ragged
    dev = _dev()
    mixes = [[2, 1, 3, 5], [1, 4, 2, 2], [3, 3, 1, 6], [10, 1, 1, 2], [2, 2, 7, 1], [4, 1, 5, 3]]
    res = {}
    for overlap in (False, True):
        model = DINO(_cfg(192, 4096, 2, 2))
        model.load_state_dict(build_sd(192, 4096))
        model = model.to(dev)
        model.overlap_streams = overlap
        model.current_epoch = 1
        model.on_train_epoch_start()
        ragged._CACHE.clear()
        out = []
        for step, nch in enumerate(mixes):
            imgs = P.make_images(nch, [224, 224, 96, 96], seed=200 + step)
            crops, labels, ncl = one_channel_collate_fn(imgs)
            batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
            for p in model.parameters():
                p.grad = None
            loss = model.training_step(batch, step)
            loss.backward()
            model.on_after_backward()
            torch.cuda.synchronize()
            out.append((loss.item(), model._last_outs["momentum_z"].float().clone(),
                        {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
        res[overlap] = out
    for step, (a, b) in enumerate(zip(res[False], res[True])):
        assert np.isfinite(a[0]) and a[0] == b[0], (step, a[0], b[0])
        assert torch.equal(a[1], b[1]), step
        assert set(a[2]) == set(b[2])
        for n, g in a[2].items():
            # the weight-gradient GEMMs run on a side stream too in one of the runs: same kernels, same split order -> identical
            assert torch.equal(g, b[2][n]), (step, n)


def test_crop_buffer_modified_between_forward_and_backward_is_detected():
    """The patch-conv weight gradient reads the caller's crop buffer in backward (no private copy is kept for fp32 contiguous input):
    an in-place torch op on it in between must raise instead of silently corrupting the gradient."""
    dev = _dev()
    m = _backbone(192, 5, dev)
    x = torch.rand((4, 1, 224, 224), device=dev)
    out = m.forward_ragged(x, [3, 1])
    x.mul_(0.5)
    with pytest.raises(RuntimeError, match="modified in place between forward and backward"):
        out.sum().backward()
    out = m.forward_ragged(x, [3, 1])
    out.sum().backward()   # untouched buffer: fine
    assert torch.isfinite(m.token_learner.proj.weight.grad).all()


def test_a_kept_loss_does_not_keep_the_batch_alive():
    """autograd lets go of saved tensors when a backward without retain_graph is done; the custom passes (backbone, head, loss) keep theirs
    on `ctx` and must do the same -- a training loop that keeps its loss tensors (a list of per-step losses) once kept every step's crop
    buffer, index arrays and final activations alive through the losses' graphs (1-2 GB per step at the bench's batch; found by
    scratch/r4/fed_soak.py).  A second backward through a released pass fails loudly."""
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    model = DINO(_cfg(192, 4096, 2, 2)).to(dev)
    tr = Trainer(max_epochs=4, steps_per_epoch=100).attach(model)
    kept, mem = [], []
    for i in range(8):
        crops, labels, ncl = one_channel_collate_fn(P.make_images([3] * 16, [224, 224, 96, 96], seed=900 + i))
        kept.append(tr.train_step(([c.to(dev) for c in crops], labels.to(dev), ncl), i))
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated())
    assert all(torch.isfinite(l) for l in kept)
    per_step_inputs = 16 * 3 * (2 * 224 * 224 + 2 * 96 * 96) * 4
    assert mem[-1] - mem[3] < per_step_inputs // 2, (mem, per_step_inputs)   # four more kept losses: nowhere near four batches
    with pytest.raises(RuntimeError, match="second backward"):
        kept[-1].backward()


@pytest.mark.parametrize("n_small,use_bn,parallel", [(0, False, True), (2, False, True), (0, True, True), (2, False, False), (0, True, False)])
def test_graphed_train_step_matches_eager(n_small, use_bn, parallel):
    """chadavit_amd.graphed.GraphedTrainStep (the whole training step as one hipGraph, device-resident LR / bias corrections / tau /
    teacher temperature) against Trainer.train_step on the same batches: same kernels in the same order, so the losses, the
    student, the EMA teacher, the centre, Adam's moments and the schedules must come out IDENTICAL -- across the epoch boundary
    where the prototypes unfreeze (a second graph: other parameters are active, their Adam step counters lag) and the teacher
    temperature moves.  Round 4 (advisor): with BatchNorm in the heads the running estimates must come out identical and finite (the
    capture's warm-up steps once ran the optimiser on unset device scalars and left NaN statistics behind); `parallel`: the step captured
    with DINO's side streams on (teacher || student forward, local-crop pass || backward, weight-gradient GEMMs || backward: a graph
    with parallel branches, the default) and on one stream -- identical results either way; between replays the
    8-entry cache of ragged descriptions is churned and the freed device blocks are overwritten (a graph must own the index arrays
    it baked in); the loss tensors of all steps are kept and read at the end (each replay must hand out its own)."""
    from chadavit_amd import ragged
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from chadavit_amd.graphed import GraphedTrainStep
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    dev = _dev()
    sizes = [224, 224] + [96] * n_small
    batches = []
    for s_ in range(6):
        nch = [1, 1, 2, 1] if s_ % 2 == 0 else [2, 1, 1, 1]   # two channel mixes alternate: each has its own graph, revisited
        crops, labels, ncl = one_channel_collate_fn(P.make_images(nch, sizes, seed=300 + s_))
        crops = crops if isinstance(crops, list) else [crops]
        batches.append(([c.to(dev) for c in crops], labels.to(dev), ncl if isinstance(ncl[0], list) else [ncl]))
    runs = {}
    for mode in ("eager", "graph"):
        cfg = _cfg(192, 4096, 2, n_small, use_bn_in_head=use_bn)
        cfg.method_kwargs.warmup_teacher_temperature_epochs = 3
        model = DINO(cfg)
        model.load_state_dict(build_sd(192, 4096, use_bn=use_bn))
        model = model.to(dev)
        # (the default capture through the trainer's own switch in one case: Trainer(graph=True).train_step IS the replay)
        via_trainer = mode == "graph" and parallel and n_small == 0 and not use_bn
        tr = Trainer(max_epochs=4, steps_per_epoch=3, graph=via_trainer).attach(model)
        if via_trainer:
            step = tr.train_step
        else:
            step = GraphedTrainStep(tr, parallel_streams=parallel) if mode == "graph" else tr.train_step
            assert mode != "graph" or (model.overlap_streams == parallel and model.backbone.dw_side_stream == parallel)
        kept, junk = [], []
        for i, b in enumerate(batches):
            tr.current_epoch = i // 3
            kept.append(step(b, i % 3))
            if mode == "graph":
                # nine other descriptions push the step's own out of the cache; whatever device blocks that frees are re-issued and
                # overwritten before the next replay reads its index arrays
                for k in range(9):
                    ragged.ragged_batch([1 + (k + i) % 3] * (k + 2), 36, dev)
                torch.cuda.synchronize()
                junk = [torch.full((n,), -1, dtype=torch.int32, device=dev) for n in (16, 64, 256, 1024, 4096, 16384) for _ in range(8)]
        losses = [float(t.item()) for t in kept]
        del junk
        if mode == "graph":
            gts = tr._graphed if via_trainer else step
            assert len(gts.graphs) == 4   # (two channel mixes) x (frozen / unfrozen prototypes)
            tr.close_graph() if via_trainer else step.close()
            assert not tr.graph
        torch.cuda.synchronize()
        opt = tr.optimizer
        with torch.no_grad():   # an eager pass after the last replay reads the weights that replay produced (not one-step-stale shadows)
            feats_after = (model.backbone(batches[0][0][0], 0, batches[0][2]).clone(), model.momentum_backbone(batches[0][0][0], 0, batches[0][2]).clone())
        runs[mode] = {"losses": losses, "feats_after": feats_after, "sd": {k: v.clone() for k, v in model.state_dict().items()},
                      "m": [sl["m"].clone() for sl in opt._slabs.values()], "tau": model.momentum_updater.cur_tau,
                      "lr": [g["lr"] for g in opt.param_groups], "gs": tr.global_step,
                      "steps": sorted({int(st.get("step", 0)) for st in opt.state.values() if "step" in st})}
    e, g = runs["eager"], runs["graph"]
    assert e["gs"] == g["gs"] == 6 and e["tau"] == g["tau"] and e["lr"] == g["lr"] and e["steps"] == g["steps"] == [3, 6]
    assert e["losses"] == g["losses"], (e["losses"], g["losses"])
    assert torch.equal(e["feats_after"][0], g["feats_after"][0]) and torch.equal(e["feats_after"][1], g["feats_after"][1])
    for k, v in e["sd"].items():
        assert torch.equal(v, g["sd"][k]), k
        assert not v.is_floating_point() or bool(torch.isfinite(v).all()), k
    for a, b in zip(e["m"], g["m"]):
        assert torch.equal(a, b)
