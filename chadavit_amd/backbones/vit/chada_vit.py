"""ChAda-ViT backbone on hand-written HIP kernels (gfx950), ragged-packed tokens.

Drop-in for the reference module surface (src/backbones/vit/chada_vit.py:136-339):
same constructor, attributes, `forward(x, index, list_num_channels)`, factory and state_dict keys.
The arithmetic is NOT the reference's op sequence: images are packed ragged (no 10-channel padding,
no key mask -- chada_vit.py:226-239 is replaced by cu_seqlens), the patch embed is one MFMA GEMM whose
epilogue adds bias + positional + channel tokens and writes the packed buffer, attention is a
var-len flash kernel, and the backward is an explicit kernel chain (no autograd graph inside).

The torch modules below (`nn.MultiheadAttention`, `nn.Linear`, ...) are PARAMETER CONTAINERS only:
they give the reference's parameter names and initial distributions; their forward is never called.
"""
from __future__ import annotations

import math
import os
from functools import partial
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...flat import FlatParams
from ...ragged import RaggedBatch, ragged_batch

FFN_DIM = 2048  # hard-wired in the reference (chada_vit.py:160)
# one fused-FFN block owns 128 token rows for the whole hidden range: below ~1 block per CU the two-GEMM path (which also
# tiles over N) fills the chip better (measured: 66 us vs 42 us at 1000 rows, 475 us vs 760 us at 301568 rows)
FUSED_FFN_MIN_ROWS = 24576


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """Same distribution as the reference helper (src/utils/misc.py:134-178): N(mean, std) truncated to
    the ABSOLUTE interval [a, b]."""
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class TransformerEncoderLayer(nn.Module):
    """Parameter container for one post-norm block (reference chada_vit.py:29-116)."""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int = FFN_DIM, dropout: float = 0.0,
                 layer_norm_eps: float = 1e-5):
        super().__init__()
        if dropout != 0.0:
            raise RuntimeError("chadavit_amd: dropout / drop_path_rate > 0 is not supported by the HIP path")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=0.0, batch_first=True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.nhead = nhead

    def forward(self, *a, **k):
        raise RuntimeError("blocks are parameter containers; call ChAdaViT.forward (HIP engine)")


class TokenLearner(nn.Module):
    """Parameter container for the 1 -> D, kernel = stride = patch conv (reference chada_vit.py:118-134)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=1, embed_dim=768):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) * (img_size // patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


_BLOCK_2D = ("self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight")


class ChAdaViT(nn.Module):
    """Channel Adaptive Vision Transformer (HIP engine)."""

    def __init__(self, img_size=[224], in_chans=1, embed_dim=192, patch_size=16, num_classes=0, depth=12, num_heads=12,
                 drop_rate=0.0, drop_path_rate=0.0, norm_layer=nn.LayerNorm, return_all_tokens=True,
                 max_number_channels=10, **kwargs):
        super().__init__()
        if drop_rate != 0.0 or drop_path_rate != 0.0:
            raise RuntimeError("chadavit_amd: drop_rate / drop_path_rate > 0 is not supported by the HIP path")
        if in_chans != 1:
            raise RuntimeError("ChAdaViT tokenises one channel at a time (in_chans must be 1)")
        self.num_features = self.embed_dim = embed_dim
        self.max_channels = max_number_channels
        self.num_heads = num_heads
        self.token_learner = TokenLearner(img_size=img_size[0], patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.token_learner.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.channel_token = nn.Parameter(torch.zeros(1, self.max_channels, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1, num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([TransformerEncoderLayer(embed_dim, num_heads, FFN_DIM) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.return_all_tokens = return_all_tokens
        trunc_normal_(self.pos_embed, std=0.02)
        trunc_normal_(self.cls_token, std=0.02)
        trunc_normal_(self.channel_token, std=0.02)
        self.apply(self._init_weights)
        self._flat: Optional[FlatParams] = None
        self._tn_ws: Optional[torch.Tensor] = None
        self._ln_ws: Optional[torch.Tensor] = None
        self.grad_ready_hook = None  # callable(flat, begin, end) fired as each slab of gradients completes
        # > 1: that many backward passes of this step accumulate into the gradient slab (DINO's standard_multicrop_loss option: the
        # global-crop and the local-crop pass); the spans are final -- and the hook fires -- during the LAST of them
        self._pending_backwards = 0
        # gradient-span hand-over vs the part of pos_embed's gradient that travels through autograd (bicubic-resized position rows): see
        # _expect_pos_accumulation
        self._pos_autograd_pending = False
        self._pos_span_deferred = None
        self._pos_hook_handle = None
        # weight-gradient GEMMs on a second HIP stream beside the dX chain.  Default by measurement (round 2, same box, img/s
        # overlapped vs one stream): Base 150.4 vs 145.4 -- its GEMM-chain kernels leave grid tails the side stream fills;
        # Tiny 4469 vs 4540, Small 1087 vs 1084, the 4-image reference config 386 vs 428 -- the fused block kernels fill the
        # chip by themselves (2 resident blocks per CU) and a concurrent kernel only splits L2 / CU slots and adds fork-join edges
        self.dw_side_stream = embed_dim >= 768
        self.fused_ffn = True        # linear1 -> relu -> linear2 (+ residual) in one kernel where the shape allows (D = 192)
        # token rows from which a block runs as the whole-block kernel (out-proj .. next QKV in one launch) instead of the
        # GEMM + LayerNorm chain; 0 forces the fused path at any size (the parity tests do, so that the benchmarked dispatch
        # is the one checked against the oracle)
        self.fused_min_rows = FUSED_FFN_MIN_ROWS
        self._dw_stream = None
        # the LayerNorm backward at a block boundary (norm1' of block i, norm2' of block i-1) as one sweep instead of two launches
        self.fused_ln_pair = True
        # "bf16" (default) or "fp8": the block's four nn.Linear forwards (in_proj, out_proj, linear1, linear2) on the MX-scaled fp8
        # MFMA -- weights AND their input activations quantised to OCP-MX e4m3 (BASELINE.json configs[4], ChAda-ViT-Base); the
        # backward keeps bf16 operands except the FFN's dX GEMMs (fp8_dx below).  Needs embed_dim % 128 == 0.
        self.weight_dtype = "bf16"
        # return_all_tokens = False: only norm(x)[:, 0] leaves forward() (reference chada_vit.py:272-289), so of the LAST block's output only
        # the CLS rows are ever read -- its attention output, out-projection, LayerNorms and FFN run on one row per image (K / V
        # projections on all rows: the CLS query attends to every token).  Same CLS features and gradients; 1/12 of the encoder's
        # row-wise work and 4/5 of that block's attention less.  (_last_block_cls_fwd / _bwd)
        self.cls_only_last_block = not os.environ.get("CHADAVIT_FULL_LAST_BLOCK")
        self.fp8_ln_emits_operand = True  # fp8 path: LayerNorm kernels also emit the following GEMM's quantised operand (measured neutral)
        # fp8 path: the FFN's two dX GEMMs (dH = dz W2 under the ReLU pattern, dx1 = dz + dH W1) on the MX-scaled MFMA as well: dz is
        # quantised in one pass, dH leaves its GEMM both as bf16 (the weight gradient's operand) and quantised, W^T has its own MX copy.
        # cfg5 224 -> 230 images/s same box; gradient bar of tests/test_model_gpu.py::test_fp8_weight_path_vs_golden unchanged (worst
        # per-tensor norm 6.7e-2 -> 7.1e-2, lowest cosine 0.941 -> 0.937).  CHADAVIT_FP8_DX=0: bf16 dX GEMMs (same-box A/B).
        self.fp8_dx = os.environ.get("CHADAVIT_FP8_DX", "1") == "1"
        # fp8 path: substrings of weight names whose FORWARD GEMM stays on bf16 operands (e.g. ("blocks.0.self_attn.in_proj",)): the error
        # budget of the 45 fp8 GEMMs of a pass, measured per GEMM kind and per block in profiles/r05b_fp8_error_budget.md
        self.fp8_keep_bf16: Tuple[str, ...] = ()
        self._capture_blocks = None  # tests: {block index: None} -> filled with that block's output (packed rows) by the forward

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------------------------------
    # flat storage
    # ------------------------------------------------------------------------------------------
    def _named_own_params(self):
        return [(n, p) for n, p in self.named_parameters() if not n.startswith("head.")]

    def flat_params(self) -> FlatParams:
        dev = self.cls_token.device
        if dev.type != "cuda":
            raise RuntimeError("ChAdaViT (chadavit_amd) runs on the GPU only: move the module to cuda first")
        if self._flat is None or self._flat.device != dev or not self._flat.attached():
            tn = ["token_learner.proj.weight"] + [f"blocks.{i}.{s}" for i in range(len(self.blocks)) for s in _BLOCK_2D]
            nb = len(self.blocks)
            ffn = [(f"blocks.{i}.linear1.weight", f"blocks.{i}.linear2.weight", f"blocks.{i}.self_attn.out_proj.weight",
                    f"blocks.{i + 1}.self_attn.in_proj_weight" if i + 1 < nb else None) for i in range(nb)] if self.fused_ffn else []
            self._flat = FlatParams(self._named_own_params(), dev, transpose_names=tn, ffn_pairs=ffn)
        return self._flat

    def _workspaces(self, dev):
        D = self.embed_dim
        if self._tn_ws is None or self._tn_ws.device != dev:
            # partial slabs of the weight-gradient GEMM: room for 32 T-splits of the widest gradient (dW_qkv: 3D x D) -- chadavit_gemm_tn picks
            # the split count that fills whole rounds of every XCD and is clamped by what fits here (Base: 57 M floats)
            self._tn_ws = torch.empty(max(24 * 1024 * 1024, 32 * (3 * D * D + 3 * D), 4 * (FFN_DIM * D + FFN_DIM)), device=dev, dtype=torch.float32)
            self._ln_ws = ops.layernorm_bwd_workspace(D, dev)
        return self._tn_ws, self._ln_ws

    # ------------------------------------------------------------------------------------------
    # positional rows for the patch tokens (tiny; stays a torch op so autograd reaches pos_embed)
    # ------------------------------------------------------------------------------------------
    def patch_pos_embed(self, w: int, h: int) -> torch.Tensor:
        """(g*g, D) rows added to the patch tokens of a w x h crop.  Same-size crops use pos_embed[1:];
        otherwise bicubic with scale_factor=(g+0.1)/sqrt(N) exactly as the reference (chada_vit.py:202-217)."""
        pos = self.pos_embed[0, 0]
        N = pos.shape[0] - 1
        ps = self.token_learner.patch_size
        npatch = (w // ps) * (h // ps)
        if npatch == N and w == h:
            return pos[1:]
        dim = pos.shape[1]
        w0, h0 = w // ps + 0.1, h // ps + 0.1
        n0 = int(math.sqrt(N))
        out = F.interpolate(pos[1:].reshape(1, n0, n0, dim).permute(0, 3, 1, 2),
                            scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
        assert int(w0) == out.shape[-2] and int(h0) == out.shape[-1]
        return out.permute(0, 2, 3, 1).reshape(-1, dim)

    # ------------------------------------------------------------------------------------------
    # public forward surface
    # ------------------------------------------------------------------------------------------
    def _expect_pos_accumulation(self):
        """A backward pass returned `dpos_patch` to autograd: pos_embed.grad (a view of the gradient slab) is complete only after autograd's
        accumulation into it, which happens once per backward call, some time after the pass's own backward function has returned.  Until
        then the slab span that holds pos_embed must not be handed to the gradient all-reduce: the pass that fires the span hooks parks that
        span in `_pos_span_deferred`, and this (persistent, lazily registered) post-accumulate hook releases it."""
        self._pos_autograd_pending = True
        if self._pos_hook_handle is None:
            def _after_accumulate(_p, self=self):
                self._pos_autograd_pending = False
                deferred, self._pos_span_deferred = self._pos_span_deferred, None
                if deferred is not None:
                    deferred()
            self._pos_hook_handle = self.pos_embed.register_post_accumulate_grad_hook(_after_accumulate)

    def forward(self, x, index, list_num_channels):
        nch = list_num_channels[index]
        if isinstance(nch, int):
            nch = [nch] * (x.shape[0] // nch)
        return self.forward_ragged(x, nch)

    def forward_ragged(self, x: torch.Tensor, num_channels: Sequence[int], rb: Optional[RaggedBatch] = None,
                       max_channels: int = 10) -> torch.Tensor:
        """x (sum C_i, 1, S, S) fp32 -> (B, D) CLS features, or (sum C_i*p, D) valid patch tokens when
        return_all_tokens.  `max_channels` mirrors the reference's hard-coded tokenizer default (10):
        channel tokens are added only if it equals self.max_channels (chada_vit.py:219,248)."""
        if x.device.type != "cuda":
            raise RuntimeError("chadavit_amd has no CPU path: input must be a GPU tensor")
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != x.shape[3]:
            raise RuntimeError(f"expected (sum C, 1, S, S) input, got {tuple(x.shape)}")
        if sum(num_channels) != x.shape[0]:
            raise RuntimeError(f"list_num_channels sums to {sum(num_channels)} but x has {x.shape[0]} channel images")
        if max(num_channels) > max_channels:
            raise RuntimeError(f"an image has more than {max_channels} channels")  # torch.stack fails in the reference (:232)
        S = x.shape[-1]
        ps = self.token_learner.patch_size
        if S < ps:
            raise RuntimeError("crop side smaller than the patch size")   # (a side that is no multiple of it loses its remainder, as in the conv)
        if rb is None:
            rb = ragged_batch(num_channels, (S // ps) ** 2, x.device)
        else:
            rb.use_on_current_stream()
        flat = self.flat_params()
        pos_patch = self.patch_pos_embed(S, S)
        add_chan = (max_channels == self.max_channels)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in flat.params)
        params = [p for p in flat.params] if need_grad else []
        pos_direct = pos_patch.shape[0] == self.pos_embed.shape[2] - 1 and S == self.token_learner.img_size
        if pos_direct:
            pos_patch = pos_patch.detach()  # same-size crops: dpos is added to the flat gradient slab directly
        return _BackboneFn.apply(self, x, rb, pos_patch, add_chan, need_grad, pos_direct, *params)

    def channel_aware_tokenization(self, x, index, list_num_channels, max_channels=10):
        """Ragged restatement of chada_vit.py:219-270: returns (packed tokens (T, D) fp32, cu_seqlens)
        instead of the padded (B, 1+10p, D) tensor + mask, which never exists here."""
        nch = list_num_channels[index]
        if isinstance(nch, int):
            nch = [nch] * (x.shape[0] // nch)
        rb = RaggedBatch(nch, (x.shape[-1] // self.token_learner.patch_size) ** 2, x.device)
        with torch.no_grad():
            flat = self.flat_params()
            flat.refresh(need_transposes=False)
            tok, _ = _tokenize(self, flat, x, rb, self.patch_pos_embed(x.shape[-1], x.shape[-1]).float().contiguous(),
                               max_channels == self.max_channels)
        return tok.float(), rb.cu_seqlens

    @torch.no_grad()
    def get_last_selfattention(self, x: torch.Tensor) -> torch.Tensor:
        """Per-head attention probabilities of the LAST block, (B, H, N, N) fp32 -- reference chada_vit.py:313-320
        (consumer main_attn.py:202-207).  As there: x is (B, 1, S, S) one-channel images, tokenised with max_channels=1, so no
        channel token is added; blocks 0..depth-2 run normally, the last block stops after softmax(QK^T/sqrt(dh))."""
        if x.device.type != "cuda":
            raise RuntimeError("chadavit_amd has no CPU path: input must be a GPU tensor")
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != x.shape[3]:
            raise RuntimeError(f"expected (B, 1, S, S) input, got {tuple(x.shape)}")
        S = x.shape[-1]
        ps = self.token_learner.patch_size
        rb = RaggedBatch([1] * x.shape[0], (S // ps) ** 2, x.device)
        flat = self.flat_params()
        flat.refresh(need_transposes=False)
        tok, _ = _tokenize(self, flat, x, rb, self.patch_pos_embed(S, S).detach().float().contiguous(), 1 == self.max_channels)
        xcur, hcur, stcur, qcur = tok, None, None, None
        last = len(self.blocks) - 1
        for i in range(last):
            xcur, _, hcur, stcur, qcur = _block_fwd(self, flat, i, xcur, rb, False, h=hcur, st=stcur, qkv=qcur)
        b = f"blocks.{last}."
        qkv = qcur  # (already produced by the previous block's kernel where that is fused)
        if isinstance(hcur, tuple):  # (fp8 weight path: (h, its quantised copy))
            hcur = hcur[0]
        if qkv is None:
            if hcur is None:  # depth 1: no previous block computed norm1 of this one
                hcur = ops.layernorm_fwd(xcur, flat.f(b + "norm1.weight"), flat.f(b + "norm1.bias"), self.blocks[last].norm1.eps)
            qkv = ops.gemm_nt(hcur, flat.w(b + "self_attn.in_proj_weight"), bias=flat.f(b + "self_attn.in_proj_bias"))
        return ops.attn_probs(qkv, rb.cu_seqlens, rb.lens, self.blocks[last].nhead)

    def extra_repr(self):
        return f"embed_dim={self.embed_dim}, heads={self.num_heads}, depth={len(self.blocks)}, engine=hip/gfx950"


# ==============================================================================================
# engine: explicit forward / backward kernel chains
# ==============================================================================================
def _tokenize(m: ChAdaViT, flat: FlatParams, x, rb: RaggedBatch, pos_patch, add_chan):
    S = x.shape[-1]
    ps = m.token_learner.patch_size
    D = m.embed_dim
    xs = x.reshape(-1, S, S)
    if S % ps != 0:   # Conv2d(kernel = stride = patch) never reads the trailing S % patch rows / columns (chada_vit.py:118-134)
        S = S // ps * ps
        xs = xs[:, :S, :S]
    xs = xs.float().contiguous()
    tokens = torch.empty((rb.T, D), device=x.device, dtype=torch.bfloat16)
    chan = flat.f("channel_token").view(m.max_channels, D) if add_chan else None
    if ps == 16:  # the conv unfold happens inside the GEMM's operand staging: no patch buffer in HBM
        ops.tokenizer_fused(xs, flat.w("token_learner.proj.weight"), flat.f("token_learner.proj.bias"), pos_patch, chan, rb.chan_img,
                            rb.chan_idx, tokens, rb.p)
    else:
        ops.tokenizer_gemm(ops.im2col(xs, ps), flat.w("token_learner.proj.weight"), flat.f("token_learner.proj.bias"), pos_patch, chan,
                           rb.chan_img, rb.chan_idx, tokens, rb.p)
    ops.write_cls(tokens, rb.cu_seqlens, flat.f("cls_token").view(-1), flat.f("pos_embed").view(-1, D)[0].contiguous())
    return tokens, xs


def _linear(m: ChAdaViT, flat: FlatParams, xb, wname: str, bias, epilogue=ops.EPI_NONE, aux=None, xq=None, emit_q=False, want_out=True):
    """epilogue(xb W^T + bias): bf16 MFMA GEMM, or -- weight_dtype "fp8" -- OCP-MX fp8 operands on the scaled MFMA.
    fp8 only: `xq` = (bytes, scales) of xb when a producing epilogue already quantised it; `emit_q` returns (out, (bytes, scales)) with
    the result quantised by THIS GEMM's epilogue for the next one (out is None when want_out is False)."""
    if m.weight_dtype == "fp8" and m.fp8_keep_bf16 and any(pat in wname for pat in m.fp8_keep_bf16):
        out = ops.gemm_nt(xb, flat.w(wname), bias=bias, epilogue=epilogue, aux=aux)
        return (out, ops.mx8_quantize(out)) if emit_q else out
    if m.weight_dtype == "fp8":
        n, k = flat.shapes[wname]
        if n % 128 == 0 and k % 128 == 0 and epilogue in (ops.EPI_NONE, ops.EPI_RELU, ops.EPI_RESID):
            wq, ws = flat.mx8(wname)
            xq, xs = xq if xq is not None else ops.mx8_quantize(xb)
            return ops.gemm_nt_mx8(xq, xs, wq, ws, bias=bias, epilogue=epilogue, aux=aux, emit_q=emit_q, want_out=want_out)
        raise RuntimeError(f"weight_dtype='fp8': {wname} {n}x{k} is not a multiple of 128 (embed_dim must be)")
    return ops.gemm_nt(xb, flat.w(wname), bias=bias, epilogue=epilogue, aux=aux)


def _block_fwd(m: ChAdaViT, flat: FlatParams, i: int, x, rb: RaggedBatch, save: bool, h=None, st=None, qkv=None):
    """One post-norm block.  `h` = LN1(x) may come precomputed (with its stats in st[0:2]) from the previous block's fused
    norm2 -> next-norm1 pass; the block in turn returns the NEXT block's h the same way."""
    b = f"blocks.{i}."
    T = x.shape[0]
    dev = x.device
    eps = m.blocks[i].norm1.eps
    H = m.blocks[i].nhead
    if st is None:
        st = torch.empty((6, T), device=dev, dtype=torch.float32) if save else None
    g1, b1 = flat.f(b + "norm1.weight"), flat.f(b + "norm1.bias")
    # fp8 weight path: the LayerNorm kernels hand the following GEMM its quantised operand (no separate quantise pass)
    fq = m.weight_dtype == "fp8" and x.shape[1] in ops.LN_PAIR_WIDTHS and m.fp8_ln_emits_operand
    hq = None
    if isinstance(h, tuple):
        h, hq = h
    if h is None and qkv is None:
        if fq:
            h, hq = ops.layernorm_fwd(x, g1, b1, eps, mean=st[0] if save else None, rstd=st[1] if save else None, emit_q=True)
        else:
            h = ops.layernorm_fwd(x, g1, b1, eps, mean=st[0] if save else None, rstd=st[1] if save else None)
    if qkv is None:  # (else: produced by the previous block's kernel)
        qkv = _linear(m, flat, h, b + "self_attn.in_proj_weight", flat.f(b + "self_attn.in_proj_bias"), xq=hq)
    a, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    pk = flat.ffn_packed(b + "linear1.weight") if (T >= m.fused_min_rows and m.weight_dtype != "fp8") else None
    pkp = flat.proj_ffn_packed(b + "linear1.weight") if pk is not None else None
    if pkp is None:
        y = _linear(m, flat, a, b + "self_attn.out_proj.weight", flat.f(b + "self_attn.out_proj.bias"), ops.EPI_RESID, x)
        x1q = None
        if fq:
            x1, x1q = ops.layernorm_fwd(y, g1, b1, eps, mean=st[2] if save else None, rstd=st[3] if save else None, emit_q=True)
        else:
            x1 = ops.layernorm_fwd(y, g1, b1, eps, mean=st[2] if save else None, rstd=st[3] if save else None)
    last = i + 1 >= len(m.blocks)
    h_next = st_next = qkv_next = rbits = None
    ln2 = (flat.f(b + "norm2.weight"), flat.f(b + "norm2.bias"), m.blocks[i].norm2.eps)
    if not last:
        nb = f"blocks.{i + 1}."
        ln1n = (flat.f(nb + "norm1.weight"), flat.f(nb + "norm1.bias"), m.blocks[i + 1].norm1.eps)
        st_next = torch.empty((6, T), device=dev, dtype=torch.float32) if save else None
    if pkp is not None:
        # one kernel from the attention output on: out-proj + residual + norm1, FFN with the hidden activation on chip (written
        # out only when saving), norm2 and the next block's norm1
        hid = torch.empty((T, flat.shapes[b + "linear1.weight"][0]), device=dev, dtype=torch.bfloat16) if save else None
        z = torch.empty((T, x.shape[1]), device=dev, dtype=torch.bfloat16) if save else None
        y = torch.empty((T, x.shape[1]), device=dev, dtype=torch.bfloat16) if save else None
        fuse_q = (not last) and flat.packed_has_next_qkv(b + "linear1.weight")
        # the ReLU pattern of the hidden activation (1 bit per element) for the fused backward dX pass
        rbits = ops.relu_bits_buffer(T, flat.shapes[b + "linear1.weight"][0], dev) if (save and flat.ffn_packed_bwd(b + "linear1.weight") is not None) else None
        r = ops.proj_ffn_ln_fwd(a, x, pkp, flat.f(b + "self_attn.out_proj.bias"), (g1, b1, eps), flat.f(b + "linear1.bias"),
                                flat.f(b + "linear2.bias"), ln2, y=y, stats1=(st[2], st[3]) if save else None, z=z, h=hid,
                                ln_b=None if last else ln1n, stats_a=(st[4], st[5]) if save else None,
                                stats_b=(st_next[0], st_next[1]) if (save and not last) else None, want_x1=save,
                                qkv_bias=flat.f(nb + "self_attn.in_proj_bias") if fuse_q else None, want_hn=save, relu_bits=rbits)
        if fuse_q:  # the next block's qkv comes out of this kernel; its h is written only for the backward (dW_qkv)
            x1, x2, h_next, qkv_next = r
        else:
            x1, x2, h_next = r
    elif pk is not None:
        # one kernel: FFN with the hidden activation on chip (written out only when saving) + norm2 + the next block's norm1
        hid = torch.empty((T, flat.shapes[b + "linear1.weight"][0]), device=dev, dtype=torch.bfloat16) if save else None
        z = torch.empty((T, x1.shape[1]), device=dev, dtype=torch.bfloat16) if save else None
        rbits = ops.relu_bits_buffer(T, flat.shapes[b + "linear1.weight"][0], dev) if (save and flat.ffn_packed_bwd(b + "linear1.weight") is not None) else None
        x2, h_next = ops.ffn_ln_fwd(x1, pk, flat.f(b + "linear1.bias"), flat.f(b + "linear2.bias"), ln2, resid=x1, z=z, h=hid,
                                    ln_b=None if last else ln1n, stats_a=(st[4], st[5]) if save else None,
                                    stats_b=(st_next[0], st_next[1]) if (save and not last) else None, relu_bits=rbits)
    else:
        if m.weight_dtype == "fp8":
            # linear1's epilogue hands linear2 its fp8 operand (no quantise pass over the hidden activation); the bf16 copy is written
            # only when the backward needs it
            hid, hidq = _linear(m, flat, x1, b + "linear1.weight", flat.f(b + "linear1.bias"), ops.EPI_RELU, xq=x1q, emit_q=True,
                                want_out=save or any(pat in b + "linear2.weight" for pat in m.fp8_keep_bf16))
            z = _linear(m, flat, hid, b + "linear2.weight", flat.f(b + "linear2.bias"), ops.EPI_RESID, x1, xq=hidq)
        else:
            hid = _linear(m, flat, x1, b + "linear1.weight", flat.f(b + "linear1.bias"), ops.EPI_RELU)
            z = _linear(m, flat, hid, b + "linear2.weight", flat.f(b + "linear2.bias"), ops.EPI_RESID, x1)
        if not last and fq:
            x2, h_next, hq_next = ops.layernorm_fwd2(z, ln2[0], ln2[1], ln1n[0], ln1n[1], ln2[2], ln1n[2],
                                                     stats1=(st[4], st[5]) if save else None,
                                                     stats2=(st_next[0], st_next[1]) if save else None, emit_q=True)
            h_next = (h_next, hq_next)  # the next block unpacks it
        elif not last:
            x2, h_next = ops.layernorm_fwd2(z, ln2[0], ln2[1], ln1n[0], ln1n[1], ln2[2], ln1n[2],
                                            stats1=(st[4], st[5]) if save else None, stats2=(st_next[0], st_next[1]) if save else None)
        else:
            x2 = ops.layernorm_fwd(z, ln2[0], ln2[1], ln2[2], mean=st[4] if save else None, rstd=st[5] if save else None)
    saved = (x, h, qkv, a, lse, y, x1, hid, z, st, rbits) if save else None
    return x2, saved, h_next, st_next, qkv_next


def _last_block_cls_fwd(m: ChAdaViT, flat: FlatParams, i: int, x, rb: RaggedBatch, save: bool, h=None, st=None, qkv=None):
    """The last post-norm block when only its CLS rows are read (ChAdaViT.cls_only_last_block): LN1 and the QKV projection on all
    rows (unless the previous block's kernel already produced them), attention of the CLS query per image, then out-proj +
    residual + norm1 + FFN + norm2 on B rows.  Returns (x2 of the CLS rows [B, D], saved)."""
    b = f"blocks.{i}."
    T, B = x.shape[0], rb.B
    dev = x.device
    eps = m.blocks[i].norm1.eps
    H = m.blocks[i].nhead
    if st is None:
        st = torch.empty((2, T), device=dev, dtype=torch.float32) if save else None
    g1, b1 = flat.f(b + "norm1.weight"), flat.f(b + "norm1.bias")
    fq = m.weight_dtype == "fp8" and x.shape[1] in ops.LN_PAIR_WIDTHS and m.fp8_ln_emits_operand
    hq = None
    if isinstance(h, tuple):
        h, hq = h
    if h is None and qkv is None:
        if fq:
            h, hq = ops.layernorm_fwd(x, g1, b1, eps, mean=st[0] if save else None, rstd=st[1] if save else None, emit_q=True)
        else:
            h = ops.layernorm_fwd(x, g1, b1, eps, mean=st[0] if save else None, rstd=st[1] if save else None)
    if qkv is None:
        qkv = _linear(m, flat, h, b + "self_attn.in_proj_weight", flat.f(b + "self_attn.in_proj_bias"), xq=hq)
    a, lse = ops.attn_cls_fwd(qkv, rb.cu_seqlens, H)
    xc = ops.gather_rows(x, rb.cls_rows)
    stc = torch.empty((4, B), device=dev, dtype=torch.float32) if save else None
    y = _linear(m, flat, a, b + "self_attn.out_proj.weight", flat.f(b + "self_attn.out_proj.bias"), ops.EPI_RESID, xc)
    ln2 = (flat.f(b + "norm2.weight"), flat.f(b + "norm2.bias"), m.blocks[i].norm2.eps)
    if fq:
        x1, x1q = ops.layernorm_fwd(y, g1, b1, eps, mean=stc[0] if save else None, rstd=stc[1] if save else None, emit_q=True)
        hid, hidq = _linear(m, flat, x1, b + "linear1.weight", flat.f(b + "linear1.bias"), ops.EPI_RELU, xq=x1q, emit_q=True,
                            want_out=save or any(pat in b + "linear2.weight" for pat in m.fp8_keep_bf16))
        z = _linear(m, flat, hid, b + "linear2.weight", flat.f(b + "linear2.bias"), ops.EPI_RESID, x1, xq=hidq)
    else:
        x1 = ops.layernorm_fwd(y, g1, b1, eps, mean=stc[0] if save else None, rstd=stc[1] if save else None)
        hid = _linear(m, flat, x1, b + "linear1.weight", flat.f(b + "linear1.bias"), ops.EPI_RELU)
        z = _linear(m, flat, hid, b + "linear2.weight", flat.f(b + "linear2.bias"), ops.EPI_RESID, x1)
    x2 = ops.layernorm_fwd(z, ln2[0], ln2[1], ln2[2], mean=stc[2] if save else None, rstd=stc[3] if save else None)
    saved = (x, h, qkv, a, lse, y, x1, hid, z, st, stc) if save else None
    return x2, saved


def _last_block_cls_bwd(m: ChAdaViT, flat: FlatParams, i: int, dx2, saved, rb: RaggedBatch, acc: bool, tn_ws, ln_ws, defer=False):
    """Backward of _last_block_cls_fwd: dx2 [B, D] is the gradient of the CLS rows of the block's output.  The row-wise part runs on B
    rows; the attention backward yields dK / dV for every token (and dQ on the CLS rows) from the one query row per image; from the
    QKV projection on, everything is full-width again.  Returns what _block_bwd returns."""
    b = f"blocks.{i}."
    x, h, qkv, a, lse, y, x1, hid, z, st, stc = saved
    H = m.blocks[i].nhead
    G = flat.g

    def dw(a_t, b_t, wname, bname):
        ops.gemm_tn(a_t, b_t, G(wname), colsum=G(bname), accumulate=acc, workspace=tn_ws)

    dz = ops.layernorm_bwd(dx2, z, stc[2], stc[3], flat.f(b + "norm2.weight"), G(b + "norm2.weight"), G(b + "norm2.bias"), ln_ws,
                           accumulate=acc)
    dhid = ops.gemm_nt(dz, flat.wt(b + "linear2.weight"), epilogue=ops.EPI_RELUMASK, aux=hid)
    dw(dz, hid, b + "linear2.weight", b + "linear2.bias")
    dx1 = ops.gemm_nt(dhid, flat.wt(b + "linear1.weight"), epilogue=ops.EPI_RESID, aux=dz)
    dw(dhid, x1, b + "linear1.weight", b + "linear1.bias")
    g1 = flat.f(b + "norm1.weight")
    dy = ops.layernorm_bwd(dx1, y, stc[0], stc[1], g1, G(b + "norm1.weight"), G(b + "norm1.bias"), ln_ws, accumulate=acc)
    da = ops.gemm_nt(dy, flat.wt(b + "self_attn.out_proj.weight"))
    dw(dy, a, b + "self_attn.out_proj.weight", b + "self_attn.out_proj.bias")
    dqkv = ops.attn_cls_bwd(qkv, rb.cu_seqlens, a, da, lse, H)
    dh = ops.gemm_nt(dqkv, flat.wt(b + "self_attn.in_proj_weight"))
    dw(dqkv, h, b + "self_attn.in_proj_weight", b + "self_attn.in_proj_bias")
    dy_full = ops.scatter_rows_zero(dy, rb.cls_rows, rb.T)  # the residual branch x -> y exists on the CLS rows only
    # norm1 is applied twice in the forward (chada_vit.py:96,99): its gradient gets both contributions
    if defer:
        return None, (dh, x, st[0], st[1], g1, G(b + "norm1.weight"), G(b + "norm1.bias"), dy_full)
    dx = ops.layernorm_bwd(dh, x, st[0], st[1], g1, G(b + "norm1.weight"), G(b + "norm1.bias"), ln_ws, dres=dy_full, accumulate=True)
    return dx, None


def _block_bwd(m: ChAdaViT, flat: FlatParams, i: int, dx2, saved, rb: RaggedBatch, acc: bool, tn_ws, ln_ws, side=None, keep=None,
               pend=None, defer=False):
    """acc: gradients of this backward call are ADDED to what the flat grad buffer already holds.
    pend: what block i+1 left undone (its closing norm1 backward, see `defer`) -- then dx2 is None and this block's norm2 backward runs
    chained behind it in one sweep (ops.layernorm_bwd_pair).  defer: return (None, pend) instead of (dx, None).
    side: optional HIP stream for the weight-gradient (TN) GEMMs -- they only feed the gradient slab, so they run beside
    the dX chain (LN bwd -> GEMM -> attention bwd ...) and fill its grid tails."""
    b = f"blocks.{i}."
    x, h, qkv, a, lse, y, x1, hid, z, st, rbits = saved
    H = m.blocks[i].nhead
    G = flat.g
    main = torch.cuda.current_stream()

    def dw(a_t, b_t, wname, bname):
        if side is None:
            ops.gemm_tn(a_t, b_t, G(wname), colsum=G(bname), accumulate=acc, workspace=tn_ws)
            return
        side.wait_stream(main)  # operands are complete on the main stream at this point
        # The operands must outlive the side-stream GEMM: the caller holds them in `keep` until the main stream has waited
        # for `side`.  (record_stream would also be correct but defers the allocator's reuse of multi-GB activations:
        # measured +1.6 % throughput but the reserved pool keeps growing by ~20 segments per step.)
        keep.append((a_t, b_t))
        with torch.cuda.stream(side):
            ops.gemm_tn(a_t, b_t, G(wname), colsum=G(bname), accumulate=acc, workspace=tn_ws)

    if pend is not None:
        dh_n, x_n, mean_n, rstd_n, g1_n, gw_n, gb_n, dy_n = pend  # x_n is this block's output: LN2(z)
        dz = ops.layernorm_bwd_pair(dh_n, x_n, mean_n, rstd_n, g1_n, dy_n, z, st[4], st[5], flat.f(b + "norm2.weight"), gw_n, gb_n,
                                    G(b + "norm2.weight"), G(b + "norm2.bias"), ln_ws, accumulate_a=True, accumulate_b=acc,
                                    beta_b=flat.f(b + "norm2.bias"))  # x_n is rebuilt from z in the kernel, not read
    else:
        dz = ops.layernorm_bwd(dx2, z, st[4], st[5], flat.f(b + "norm2.weight"), G(b + "norm2.weight"), G(b + "norm2.bias"), ln_ws,
                               accumulate=acc)
    if rbits is not None:
        # one launch: dH = dz W2 masked by the recorded ReLU pattern, dx1 = dz + dH W1 -- H is not re-read, dH makes no extra trip
        dhid = torch.empty_like(hid)
        dx1 = ops.ffn_bwd_dx(dz, flat.ffn_packed_bwd(b + "linear1.weight"), rbits, dpre=dhid)
        dw(dz, hid, b + "linear2.weight", b + "linear2.bias")
    elif m.weight_dtype == "fp8" and m.fp8_dx:
        w2q, w2s = flat.mx8_t(b + "linear2.weight")
        w1q, w1s = flat.mx8_t(b + "linear1.weight")
        dzq, dzs = ops.mx8_quantize(dz)
        dhid, (dhq, dhs) = ops.gemm_nt_mx8(dzq, dzs, w2q, w2s, epilogue=ops.EPI_RELUMASK, aux=hid, emit_q=True)
        dw(dz, hid, b + "linear2.weight", b + "linear2.bias")
        dx1 = ops.gemm_nt_mx8(dhq, dhs, w1q, w1s, epilogue=ops.EPI_RESID, aux=dz)
        del dzq, dzs, dhq, dhs
    else:
        dhid = ops.gemm_nt(dz, flat.wt(b + "linear2.weight"), epilogue=ops.EPI_RELUMASK, aux=hid)
        dw(dz, hid, b + "linear2.weight", b + "linear2.bias")
        dx1 = ops.gemm_nt(dhid, flat.wt(b + "linear1.weight"), epilogue=ops.EPI_RESID, aux=dz)
    dw(dhid, x1, b + "linear1.weight", b + "linear1.bias")
    del dhid
    g1 = flat.f(b + "norm1.weight")
    dy = ops.layernorm_bwd(dx1, y, st[2], st[3], g1, G(b + "norm1.weight"), G(b + "norm1.bias"), ln_ws, accumulate=acc)
    da = ops.gemm_nt(dy, flat.wt(b + "self_attn.out_proj.weight"))
    dw(dy, a, b + "self_attn.out_proj.weight", b + "self_attn.out_proj.bias")
    dqkv = ops.attn_bwd(qkv, a, da, lse, rb.cu_seqlens, rb.work, H)
    dh = ops.gemm_nt(dqkv, flat.wt(b + "self_attn.in_proj_weight"))
    dw(dqkv, h, b + "self_attn.in_proj_weight", b + "self_attn.in_proj_bias")
    # norm1 is applied twice in the forward (chada_vit.py:96,99): its gradient gets both contributions
    if defer:  # ... the second one in the next call, chained in front of block i-1's norm2 backward
        return None, (dh, x, st[0], st[1], g1, G(b + "norm1.weight"), G(b + "norm1.bias"), dy)
    dx = ops.layernorm_bwd(dh, x, st[0], st[1], g1, G(b + "norm1.weight"), G(b + "norm1.bias"), ln_ws, dres=dy,
                           accumulate=True)
    return dx, None


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m: ChAdaViT, x, rb: RaggedBatch, pos_patch, add_chan: bool, need_grad: bool, pos_direct: bool, *params):
        flat = m.flat_params()
        flat.refresh(need_transposes=need_grad)
        D = m.embed_dim
        pos_c = pos_patch.detach().float().contiguous()
        tok, patches = _tokenize(m, flat, x, rb, pos_c, add_chan)
        saved_blocks = []
        xcur, hcur, stcur, qcur = tok, None, None, None
        L = len(m.blocks) - 1
        cls_last = (m.cls_only_last_block and not m.return_all_tokens
                    and not (m._capture_blocks is not None and L in m._capture_blocks))
        for i in range(len(m.blocks)):
            if i == L and cls_last:
                xcur, sv = _last_block_cls_fwd(m, flat, i, xcur, rb, need_grad, h=hcur, st=stcur, qkv=qcur)
                saved_blocks.append(sv)
                break
            xcur, sv, hcur, stcur, qcur = _block_fwd(m, flat, i, xcur, rb, need_grad, h=hcur, st=stcur, qkv=qcur)
            saved_blocks.append(sv)
            if m._capture_blocks is not None and i in m._capture_blocks:
                m._capture_blocks[i] = xcur.float()
        gn, bn = flat.f("norm.weight"), flat.f("norm.bias")
        if m.return_all_tokens:
            st = torch.empty((2, rb.T), device=x.device, dtype=torch.float32) if need_grad else None
            full = ops.layernorm_fwd(xcur, gn, bn, m.norm.eps, mean=st[0] if need_grad else None, rstd=st[1] if need_grad else None)
            keep = torch.ones(rb.T, device=x.device, dtype=torch.bool)
            keep[rb.cls_rows.long()] = False
            out = full[keep].float()  # valid non-CLS tokens, image-major (chada_vit.py:283-287)
            ctx.mode = "all"
            ctx.final = (xcur, st, keep) if need_grad else None
        else:
            xc = xcur if cls_last else ops.gather_rows(xcur, rb.cls_rows)   # (cls_last: the last block already ran on the CLS rows)
            st = torch.empty((2, rb.B), device=x.device, dtype=torch.float32)
            fc = ops.layernorm_fwd(xc, gn, bn, m.norm.eps, mean=st[0], rstd=st[1])
            out = fc.float()
            ctx.mode = "cls"
            ctx.final = (xc, st)
        ctx.cls_last = cls_last
        ctx.m, ctx.rb, ctx.add_chan, ctx.need_grad = m, rb, add_chan, need_grad
        ctx.saved_blocks = saved_blocks if need_grad else None
        ctx.patches = patches if need_grad else None
        # `patches` may be a VIEW of the caller's crop buffer (fp32 contiguous input): it must stay untouched until backward, and
        # as a plain ctx attribute it is outside autograd's own version check -- so keep the counter and check it there
        ctx.patches_version = patches._version if need_grad else None
        ctx.pos_needs_grad = pos_patch.requires_grad
        ctx.pos_direct = pos_direct
        ctx.n_params = len(params)
        return out

    @staticmethod
    def backward(ctx, dout):
        m: ChAdaViT = ctx.m
        rb: RaggedBatch = ctx.rb
        if not ctx.need_grad:
            return (None,) * (7 + ctx.n_params)
        if ctx.saved_blocks is None:
            raise RuntimeError("second backward through the same ChAdaViT pass: its saved activations were released by the first one "
                               "(retain_graph is not supported; run the forward again)")
        flat = m.flat_params()
        dev = dout.device
        D = m.embed_dim
        tn_ws, ln_ws = m._workspaces(dev)
        G = flat.g
        # accumulate iff an earlier backward of this step already filled the flat gradient slab
        acc = m.cls_token.grad is not None and m.cls_token.grad.data_ptr() == G("cls_token").data_ptr()
        if m.cls_token.grad is not None and not acc:
            for n, p in zip(flat.names, flat.params):  # foreign .grad tensors: fold them into the slab
                if p.grad is not None:
                    G(n).copy_(p.grad)
            acc = True
        if ctx.mode == "cls":
            xc, st = ctx.final
            dxc = ops.layernorm_bwd(dout.to(torch.bfloat16).contiguous(), xc, st[0], st[1], flat.f("norm.weight"), G("norm.weight"),
                                    G("norm.bias"), ln_ws, accumulate=acc)
            dx = dxc if ctx.cls_last else ops.scatter_rows_zero(dxc, rb.cls_rows, rb.T)
        else:  # all patch tokens returned (chada_vit.py:283-287): the CLS rows of the final LayerNorm get no gradient
            xlast, st, keep = ctx.final
            dfull = torch.zeros((rb.T, D), device=dev, dtype=torch.bfloat16)
            dfull[keep] = dout.to(torch.bfloat16)
            dx = ops.layernorm_bwd(dfull, xlast, st[0], st[1], flat.f("norm.weight"), G("norm.weight"), G("norm.bias"), ln_ws,
                                   accumulate=acc)
        hook = m.grad_ready_hook
        if m._pending_backwards > 1:   # an earlier pass of several: nothing is final yet
            m._pending_backwards -= 1
            hook = None
        else:
            m._pending_backwards = 0
        if hook is not None:
            hook(flat, *flat.span(["norm.weight", "norm.bias"]))
        main = torch.cuda.current_stream(dev)
        side = None
        if m.dw_side_stream:
            if m._dw_stream is None:
                m._dw_stream = torch.cuda.Stream(device=dev)
            side = m._dw_stream
            side.wait_stream(main)
        keep = []
        pair = m.fused_ln_pair and D in ops.LN_PAIR_WIDTHS
        pend = None
        for i in reversed(range(len(m.blocks))):
            defer = pair and i > 0
            if ctx.cls_last and i == len(m.blocks) - 1:
                dx, new_pend = _last_block_cls_bwd(m, flat, i, dx, ctx.saved_blocks[i], rb, acc, tn_ws, ln_ws, defer=defer)
            else:
                dx, new_pend = _block_bwd(m, flat, i, dx, ctx.saved_blocks[i], rb, acc, tn_ws, ln_ws, side, keep, pend=pend, defer=defer)
            ctx.saved_blocks[i] = None
            if side is not None and (hook is not None or (i % 3) == 0):
                main.wait_stream(side)  # weight gradients of the blocks so far are final; their operands may be released
                keep.clear()
            if hook is not None:
                # a block whose closing norm1 backward was deferred is complete only now (its second norm1 contribution just landed)
                for j in ([i + 1] if pend is not None else []) + ([] if defer else [i]):
                    b = f"blocks.{j}."
                    hook(flat, *flat.span([b + "self_attn.in_proj_weight", b + "norm2.bias"]))
            pend = new_pend
        if side is not None:
            main.wait_stream(side)
            keep.clear()
        # tokenizer backward (autograd of chada_vit.py:223-265)
        dpatch, dpos, dchan, dcls = ops.tokenizer_bwd(dx, rb.cu_seqlens, rb.chan_img, rb.chan_idx, rb.p, m.max_channels)
        gw = G("token_learner.proj.weight")
        # the unfolded patches exist only here, for the weight gradient of the patch conv (the forward gathers them on the fly)
        if ctx.patches._version != ctx.patches_version:
            raise RuntimeError("the crop buffer handed to ChAdaViT.forward was modified in place between forward and backward: the "
                               "patch-conv weight gradient reads it (keep crop buffers immutable until backward, or pass a copy)")
        patches = ops.im2col(ctx.patches, m.token_learner.patch_size)
        ops.gemm_tn(dpatch, patches, gw.view(D, -1), colsum=G("token_learner.proj.bias"), accumulate=acc, workspace=tn_ws,
                    t_rows=rb.n_chan * rb.p)
        del patches
        gcls, gchan, gpos = G("cls_token").view(-1), G("channel_token").view(m.max_channels, D), G("pos_embed").view(-1, D)
        if not acc:
            gchan.zero_()
            gpos.zero_()
            gcls.zero_()
        gcls.add_(dcls)
        gpos[0].add_(dcls)  # cls row = cls_token + pos_embed[0]
        if ctx.add_chan:
            gchan.add_(dchan)
        dpos_patch = None
        if ctx.pos_direct:
            gpos[1:].add_(dpos)
        elif ctx.pos_needs_grad:
            dpos_patch = dpos  # bicubic-resized rows (other crop sizes): autograd carries dpos back to pos_embed
            if m.grad_ready_hook is not None and m.pos_embed.requires_grad:
                m._expect_pos_accumulation()
        flat.publish_grads()
        if hook is not None:
            # the span that holds pos_embed is final HERE only if no part of pos_embed's gradient still travels through autograd: with
            # bicubic-resized position rows (crops of another size than img_size) `dpos_patch` is added to pos_embed.grad -- a view of the
            # gradient slab -- by autograd AFTER this function returns.  Reducing the span now would race with that add and drop the term on
            # the other ranks, so in that case the span is handed over from a one-shot hook that runs once the accumulation has happened.
            span = flat.span(["cls_token", "token_learner.proj.bias"])
            if m._pos_autograd_pending:
                m._pos_span_deferred = lambda hook=hook, flat=flat, span=span: hook(flat, *span)
            else:
                hook(flat, *span)
        # what autograd does with saved tensors after a backward without retain_graph: let go of them.  A loss tensor that is kept around
        # (a list of per-step losses) would otherwise keep this pass's crop buffer, index arrays and final activations alive through its
        # graph -- 1-2 GB per step at the bench's batch (found by scratch/r4/fed_soak.py)
        ctx.patches = ctx.final = ctx.saved_blocks = ctx.rb = None
        return (None, None, None, dpos_patch, None, None, None) + (None,) * ctx.n_params


def chada_vit(**kwargs):
    """Training factory (reference chada_vit.py:333-339): depth 12, 2 heads, final LayerNorm eps 1e-6."""
    return ChAdaViT(patch_size=kwargs["patch_size"], embed_dim=kwargs["embed_dim"], depth=12, num_heads=2,
                    norm_layer=partial(nn.LayerNorm, eps=1e-6), return_all_tokens=kwargs["return_all_tokens"],
                    max_number_channels=kwargs["max_number_channels"])
