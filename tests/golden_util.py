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
