"""Where the fp8 weight path's error comes from: the Base backbone of the golden `backbone_base` run in bf16 and
with MX-fp8 GEMMs on the SAME weights and inputs -- deviation of every block's output (all token rows) fp8 vs bf16, the final CLS
features against the reference's fp32 golden, and the same with groups of GEMMs kept on bf16 operands (ChAdaViT.fp8_keep_bf16)."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import procedural as P
from chadavit_amd.backbones import vit_channels
import importlib
CV = importlib.import_module('chadavit_amd.backbones.vit.chada_vit')
from chadavit_amd.data.channels_strategies import one_channel_collate_fn
dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "backbone_base.npz"))
D = int(g["D"])
def cos(a, b): a, b = a.double().flatten().cpu(), b.double().flatten().cpu(); return float(a @ b / (a.norm() * b.norm()))
def rel(a, b): a, b = a.double().cpu(), b.double().cpu(); return float((a - b).norm() / b.norm())
def build():
    m = vit_channels("dino", patch_size=16, embed_dim=D, return_all_tokens=False, max_number_channels=10)
    m.load_state_dict(P.fill_state_dict(P.backbone_shapes(D), seed=int(g["seed_w"])))
    m = m.to(dev); m.cls_only_last_block = False
    return m
crops, labels, ncl = one_channel_collate_fn(P.make_images([int(c) for c in g["nch"]], [int(s) for s in g["sizes"]], seed=int(g["seed_x"])))
crops = crops if isinstance(crops, list) else [crops]
nch = ncl if isinstance(ncl[0], list) else [ncl]
rec = []
orig = CV._block_fwd
def spy(m, flat, i, x, rb, save, **k):
    r = orig(m, flat, i, x, rb, save, **k)
    rec.append(r[0].float().clone())
    return r
CV._block_fwd = spy
def run(dtype, keep=()):
    m = build(); m.weight_dtype = dtype; m.fp8_keep_bf16 = tuple(keep)
    rec.clear()
    with torch.no_grad():
        cls = m(crops[0].to(dev), 0, nch)
    return cls.float(), [r for r in rec]
ref = torch.from_numpy(g["cls0"])
c16, b16 = run("bf16")
c8, b8 = run("fp8")
print(f"bf16 vs fp32 golden: CLS cosine {cos(c16, ref):.5f} rel-L2 {rel(c16, ref):.4f}")
print(f"fp8  vs fp32 golden: CLS cosine {cos(c8, ref):.5f} rel-L2 {rel(c8, ref):.4f};  fp8 vs bf16: cosine {cos(c8, c16):.5f} rel-L2 {rel(c8, c16):.4f}")
print("per block (output of block i, all token rows), fp8 vs bf16:")
for i, (a, b) in enumerate(zip(b8, b16)):
    print(f"  block {i:2d}: cosine {cos(a, b):.5f}  rel-L2 {rel(a, b):.4f}")
print("groups of forward GEMMs kept on bf16 operands (fp8_keep_bf16) -> final CLS vs fp32 golden:")
groups = [("in_proj (QKV) of every block", ["in_proj"]), ("out_proj of every block", ["out_proj"]), ("linear1 of every block", ["linear1"]),
          ("linear2 of every block", ["linear2"]), ("block 0 (all four)", ["blocks.0."]), ("block 11 (all four)", ["blocks.11."]),
          ("blocks 0-1", ["blocks.0.", "blocks.1."]), ("blocks 10-11", ["blocks.10.", "blocks.11."]),
          ("blocks 0-5", [f"blocks.{i}." for i in range(6)]), ("blocks 6-11", [f"blocks.{i}." for i in range(6, 12)]),
          ("block 0 QKV only", ["blocks.0.self_attn.in_proj"]), ("attention projections (QKV + out) of every block", ["in_proj", "out_proj"]),
          ("FFN (linear1 + linear2) of every block", ["linear1", "linear2"])]
for name, keep in groups:
    c, _ = run("fp8", keep)
    n = sum(1 for i in range(12) for k in ("in_proj", "out_proj", "linear1", "linear2")
            if any(p in f"blocks.{i}.self_attn.{k}_weight" or p in f"blocks.{i}.self_attn.{k}.weight" or p in f"blocks.{i}.{k}.weight" for p in keep))
    print(f"  {name:52s} ({n:2d} of 48 GEMMs on bf16): cosine {cos(c, ref):.5f}  rel-L2 {rel(c, ref):.4f}")
