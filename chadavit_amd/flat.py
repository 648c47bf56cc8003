"""Flat fp32 parameter / gradient storage with bf16 shadow copies (MI355X-first memory layout).

All parameters of a module tree live in ONE contiguous fp32 buffer (and their gradients in another),
so that EMA, AdamW, the bf16 weight cast and the RCCL gradient all-reduce are each a single pass over
one slab instead of ~160 per-tensor launches (the reference loops per tensor: momentum.py:73-74,
torch.optim foreach, DDP buckets).  `nn.Parameter` objects keep their identity and reference key
names; only their storage is re-pointed at views of the slab.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops

ALIGN = 64  # elements; keeps every tensor 256-byte aligned in fp32 and 128-byte aligned in bf16


class FlatParams:
    def __init__(self, named_params: Iterable[Tuple[str, nn.Parameter]], device, transpose_names: Iterable[str] = (),
                 ffn_pairs: Iterable[Tuple[str, str]] = ()):
        self.device = torch.device(device)
        self.names: List[str] = []
        self.params: List[nn.Parameter] = []
        self.offsets: Dict[str, int] = {}
        self.shapes: Dict[str, torch.Size] = {}
        off = 0
        for n, p in named_params:
            self.names.append(n)
            self.params.append(p)
            self.offsets[n] = off
            self.shapes[n] = p.shape
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self._views: Dict[tuple, torch.Tensor] = {}
        self._home: Optional[List[int]] = None   # where each parameter's storage must point (attached())
        self.flat = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.bf16 = torch.zeros(off, device=self.device, dtype=torch.bfloat16)
        with torch.no_grad():
            cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
            for n, p in zip(self.names, self.params):
                v = self.view(self.flat, n)
                old = p.data
                v.copy_(old.to(self.device, torch.float32))
                if cur is not None and old.is_cuda:
                    # the old storage is released by the next line while the copy above may still be queued on `cur`; if it
                    # was allocated on another stream the allocator would hand it out again at once
                    old.record_stream(cur)
                p.data = v
        # transposed bf16 copies (W^T) for the dX GEMMs of 2-D weights
        self.transpose_names = [n for n in transpose_names if n in self.offsets]
        self._t: Dict[str, torch.Tensor] = {}
        # all of them live in ONE buffer and are refreshed by ONE launch (descriptor table on the device)
        t_off, desc, self._t_max_tiles = 0, [], 1
        for n in self.transpose_names:
            rows = self.shapes[n][0]
            cols = self.shapes[n].numel() // rows  # conv weights (D,1,P,P) are viewed (D, P*P)
            desc += [self.offsets[n], t_off, rows, cols]
            self._t_max_tiles = max(self._t_max_tiles, ((rows + 31) // 32) * ((cols + 31) // 32))
            t_off += (rows * cols + ALIGN - 1) // ALIGN * ALIGN
        self._t_buf = torch.empty(max(t_off, 1), device=self.device, dtype=torch.bfloat16)
        self._t_desc = torch.tensor(desc, device=self.device, dtype=torch.int64) if desc else None
        for i, n in enumerate(self.transpose_names):
            rows, cols = desc[4 * i + 2], desc[4 * i + 3]
            self._t[n] = self._t_buf[desc[4 * i + 1]:desc[4 * i + 1] + rows * cols].view(cols, rows)
        # fragment-major (W1, W2) streams for the fused FFN kernel (ops.ffn_fwd), where the shape is supported
        self._pk: Dict[str, torch.Tensor] = {}
        self._pk_src: Dict[str, Tuple[str, str]] = {}
        # ffn_pairs entries are (W1, W2), (W1, W2, Wo) or (W1, W2, Wo, next in_proj weight or None): with the block's out-proj
        # weight the stream is [3 Wo blocks | FFN blocks | 9 next-QKV blocks] (ops.proj_ffn_ln_fwd) and ffn_packed() hands out
        # the FFN part of it
        pk_off, pk_desc, self._pk_shape, self._pk_proj = 0, [], None, None
        self._pk_has_qkv = set()
        self._pkp: Dict[str, torch.Tensor] = {}
        for pair in ffn_pairs:
            w1, w2 = pair[0], pair[1]
            wo = pair[2] if len(pair) > 2 else None
            wq = pair[3] if len(pair) > 3 else None
            if w1 in self.offsets and w2 in self.offsets and (wo is None or wo in self.offsets):
                ff, d = self.shapes[w1]
                proj = wo is not None and tuple(self.shapes[wo]) == (d, d) and ops.ffn_proj_packed_bytes(d, ff) > 0
                nbytes = ops.ffn_proj_packed_bytes(d, ff) if proj else ops.ffn_packed_bytes(d, ff)  # (D = 384: FFN stream only)
                if nbytes > 0 and self._pk_shape in (None, (d, ff)) and self._pk_proj in (None, proj):  # one launch packs all layers
                    self._pk_shape, self._pk_proj = (d, ff), proj
                    has_q = proj and wq is not None and wq in self.offsets and tuple(self.shapes[wq]) == (3 * d, d)
                    pk_desc += [self.offsets[w1], self.offsets[w2], self.offsets[wo], self.offsets[wq] if has_q else -1, pk_off] \
                        if proj else [self.offsets[w1], self.offsets[w2], pk_off]
                    if has_q:
                        self._pk_has_qkv.add(w1)
                    self._pk_src[w1] = (w1, w2)
                    pk_off += nbytes // 2
        self._pk_buf = torch.empty(max(pk_off, 1), device=self.device, dtype=torch.bfloat16)
        self._pk_desc = torch.tensor(pk_desc, device=self.device, dtype=torch.int64) if pk_desc else None
        for i, w1 in enumerate(self._pk_src):
            if self._pk_proj:
                n = ops.ffn_proj_packed_bytes(*self._pk_shape) // 2
                whole = self._pk_buf[pk_desc[5 * i + 4]:pk_desc[5 * i + 4] + n]
                self._pkp[w1] = whole
                d_ = self._pk_shape[0]
                blk, npb = 64 * d_, d_ // 64  # elements per stream block; blocks per [D x D] projection matrix
                self._pk[w1] = whole[npb * blk:n - 3 * npb * blk]  # the FFN blocks alone (ops.ffn_fwd / ffn_ln_fwd)
            else:
                n = ops.ffn_packed_bytes(*self._pk_shape) // 2
                self._pk[w1] = self._pk_buf[pk_desc[3 * i + 2]:pk_desc[3 * i + 2] + n]
        # the same fragment-major stream for the BACKWARD dX pass (ops.ffn_bwd_dx): W2^T in the W1 slot, W1^T in the W2 slot, both
        # taken from the transposed bf16 copies above; rebuilt with them
        self._pkb: Dict[str, torch.Tensor] = {}
        pkb_off, pkb_desc, t_off_of = 0, [], {n: desc[4 * i + 1] for i, n in enumerate(self.transpose_names)}
        if self._pk_shape is not None:
            nb = ops.ffn_packed_bytes(*self._pk_shape) // 2
            for w1, (_, w2) in self._pk_src.items():
                if w1 in t_off_of and w2 in t_off_of:
                    pkb_desc += [t_off_of[w2], t_off_of[w1], pkb_off]
                    pkb_off += nb
        self._pkb_buf = torch.empty(max(pkb_off, 1), device=self.device, dtype=torch.bfloat16)
        self._pkb_desc = torch.tensor(pkb_desc, device=self.device, dtype=torch.int64) if pkb_desc else None
        if pkb_desc:
            for i, w1 in enumerate(self._pk_src):
                self._pkb[w1] = self._pkb_buf[pkb_desc[3 * i + 2]:pkb_desc[3 * i + 2] + nb]
        self._mx8: Dict[str, Tuple[int, torch.Tensor, torch.Tensor]] = {}
        self._mx8_t: Dict[str, Tuple[int, torch.Tensor, torch.Tensor]] = {}
        self._cast_version = None
        self._cast_version_t = None
        self._manual_version = 0

    # ---- views -------------------------------------------------------------------------------
    def view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        # the slabs live as long as this object and a parameter's place in them never moves: each (slab, name) view is built once (a
        # training step asks for ~500 of them; slicing + reshaping anew cost ~1.2 ms of interpreter per step in the launch-bound regime)
        key = (buf.data_ptr(), buf.dtype, name)
        v = self._views.get(key)
        if v is None:
            o = self.offsets[name]
            shape = self.shapes[name]
            n = 1
            for s in shape:
                n *= s
            v = self._views[key] = buf[o:o + n].view(shape)
        return v

    def w(self, name: str) -> torch.Tensor:
        """bf16 shadow of a weight, viewed 2-D (rows = out features)."""
        v = self._views.get(("w2d", name))
        if v is None:
            v0 = self.view(self.bf16, name)
            v = self._views[("w2d", name)] = v0.view(v0.shape[0], -1)
        return v

    def wt(self, name: str) -> torch.Tensor:
        return self._t[name]

    def ffn_packed(self, w1_name: str) -> Optional[torch.Tensor]:
        """Packed (W1, W2) stream keyed by the first linear's weight name, or None if the shape has no fused kernel."""
        return self._pk.get(w1_name)

    def ffn_packed_bwd(self, w1_name: str) -> Optional[torch.Tensor]:
        """[W2^T | W1^T] stream of the block's FFN for ops.ffn_bwd_dx, or None."""
        return self._pkb.get(w1_name)

    def proj_ffn_packed(self, w1_name: str) -> Optional[torch.Tensor]:
        """[Wo | FFN | next QKV] stream of the block (ops.proj_ffn_ln_fwd), or None."""
        return self._pkp.get(w1_name)

    def packed_has_next_qkv(self, w1_name: str) -> bool:
        return w1_name in self._pk_has_qkv

    def mx8(self, name: str) -> Tuple[torch.Tensor, torch.Tensor]:
        """(e4m3 bytes (rows, cols), E8M0 scales (cols/32, rows)) of a 2-D weight: the OCP-MX fp8 copy the fp8 weight path multiplies
        (ops.gemm_nt_mx8), quantised from the bf16 shadow once per parameter version."""
        ver = self._cast_version
        hit = self._mx8.get(name)
        if hit is None or hit[0] != ver:
            wq, ws = ops.mx8_quantize(self.w(name), q=hit[1] if hit else None, scales=hit[2] if hit else None)
            hit = (ver, wq, ws)
            self._mx8[name] = hit
        return hit[1], hit[2]

    def mx8_t(self, name: str) -> Tuple[torch.Tensor, torch.Tensor]:
        """The OCP-MX fp8 copy of the TRANSPOSED weight (rows = in features, blocks of 32 along the out features): the W operand
        of the dX GEMMs on the scaled MFMA (ChAdaViT.fp8_dx).  Quantised from the bf16 transpose once per parameter version."""
        ver = self._cast_version_t
        hit = self._mx8_t.get(name)
        if hit is None or hit[0] != ver:
            wq, ws = ops.mx8_quantize(self.wt(name), q=hit[1] if hit else None, scales=hit[2] if hit else None)
            hit = (ver, wq, ws)
            self._mx8_t[name] = hit
        return hit[1], hit[2]

    def f(self, name: str) -> torch.Tensor:
        return self.view(self.flat, name)

    def g(self, name: str) -> torch.Tensor:
        return self.view(self.grad, name)

    def span(self, names: List[str]) -> Tuple[int, int]:
        """[begin, end) element range covering a run of consecutively laid out parameters."""
        o0 = self.offsets[names[0]]
        last = names[-1]
        n = 1
        for s in self.shapes[last]:
            n *= s
        return o0, self.offsets[last] + (n + ALIGN - 1) // ALIGN * ALIGN

    # ---- freshness of the bf16 shadows ---------------------------------------------------------
    def attached(self) -> bool:
        """Do the parameters still live in the slab (nobody re-pointed a `.data`)?  Asked a dozen times per step: one list comparison."""
        if self._home is None:
            base = self.flat.data_ptr()
            self._home = [base + 4 * self.offsets[n] for n in self.names]
        return [p.data_ptr() for p in self.params] == self._home

    def _version(self) -> int:
        return sum(p._version for p in self.params) + self._manual_version

    def mark_dirty(self):
        self._manual_version += 1

    def refresh(self, need_transposes: bool = True):
        """Re-cast fp32 -> bf16 (and W^T) if any parameter changed since the last cast."""
        ver = self._version()
        if ver != self._cast_version:
            ops.cast_bf16(self.flat, self.bf16)
            if self._pk_desc is not None and self._pk_proj:
                ops.ffn_pack_proj_batched(self.bf16, self._pk_buf, self._pk_desc, len(self._pk_src), *self._pk_shape)
            elif self._pk_desc is not None:
                ops.ffn_pack_batched(self.bf16, self._pk_buf, self._pk_desc, len(self._pk_src), *self._pk_shape)
            self._cast_version = ver
            # fp8 weight path: every weight that HAS an MX-fp8 copy is re-quantised here, eagerly, on the stream refresh() runs
            # on (DINO.training_step calls it on the main stream before the side streams fork).  Left lazy, the first pass to
            # ask re-quantised in place on ITS stream while a pass on another stream could still hit the cache and read the
            # buffers half written.
            for name in list(self._mx8):
                self.mx8(name)
        if need_transposes and ver != self._cast_version_t:
            if self._t_desc is not None:
                ops.cast_transpose_batched(self.flat, self._t_buf, self._t_desc, len(self.transpose_names), self._t_max_tiles)
                if self._pkb_desc is not None:
                    ops.ffn_pack_batched(self._t_buf, self._pkb_buf, self._pkb_desc, len(self._pkb), *self._pk_shape)
            self._cast_version_t = ver
            for name in list(self._mx8_t):   # (eagerly, as the forward copies above)
                self.mx8_t(name)

    # ---- gradient views handed to autograd users ---------------------------------------------------
    def grad_target(self, name: str, p: nn.Parameter) -> Tuple[torch.Tensor, bool]:
        """(flat grad view, accumulate?) honouring an existing p.grad (set by an earlier backward call)."""
        gv = self.g(name)
        if p.grad is None:
            return gv, False
        if p.grad.data_ptr() != gv.data_ptr():
            gv.copy_(p.grad)
        return gv, True

    def publish_grads(self, names: Optional[Iterable[str]] = None):
        for n, p in zip(self.names, self.params):
            if names is not None and n not in names:
                continue
            if p.requires_grad:
                p.grad = self.g(n)
