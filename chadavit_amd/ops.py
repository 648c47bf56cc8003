"""Thin torch-tensor -> C-ABI call wrappers (device pointers + sizes + current HIP stream).

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic happens in
libchadavit_hip.so.  Every wrapper validates dtype / contiguity / device and raises RuntimeError on a
non-zero return code (mirrors the `c10::Error -> RuntimeError` contract of SURVEY.md section 8(b)).
"""
from __future__ import annotations

import ctypes
import os
import time
from typing import Optional

import torch

from ._lib import lib

# The entry points carry ctypes argtypes (chadavit_amd._lib, parsed from the header): arguments go in as plain Python numbers / addresses.
# The names below survive from the time every argument was wrapped in a ctypes object; they now only normalise the Python type.
c_int, c_float, c_ll = int, float, int

EPI_NONE, EPI_RELU, EPI_GELU, EPI_RESID, EPI_RELUMASK, EPI_GELUBWD = range(6)


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


# torch.cuda.current_stream() builds a Stream object through several Python layers (~8 us): at ~200 launches per pass that was a fifth of the
# launch-bound step's host time.  The raw handle of the current stream of the current device is one C call away.
_raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def _stream():
    return _raw_stream(_cur_device())


_DEBUG_SYNC = bool(os.environ.get("CHADAVIT_DEBUG_SYNC"))


def _chk(rc: int, name: str):
    if _DEBUG_SYNC:  # localise an asynchronous GPU fault to the entry point that caused it
        torch.cuda.synchronize()
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc} (1=bad argument, 2=unsupported shape, >=1000: hipError_t+1000)")


def _req(t: torch.Tensor, dtype, name: str):
    if t.device.type != "cuda":
        raise RuntimeError(f"{name}: expected a GPU tensor (chadavit_amd has no CPU path)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")


BF16, F32, I32, I64 = torch.bfloat16, torch.float32, torch.int32, torch.int64


# ---- optional per-launch timing with HIP events on the launch stream (used by bench.py's roofline leg) ----
class LaunchProfiler:
    """Records (key, start, stop) HIP-event pairs around the instrumented entry points while active.
    Events are recorded on torch's current stream, which is the stream the kernels are launched on."""

    def __init__(self, only=None):
        self.records = []
        # restrict the event pairs to one (entry point, shape) key or a set of them: negligible perturbation of the step
        self.only = None if only is None else (set(only) if isinstance(only, (set, frozenset, list)) else {only})

    def __enter__(self):
        global _PROF
        _PROF = self
        return self

    def __exit__(self, *exc):
        global _PROF
        _PROF = None

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, a, b, host_s in self.records:
            d = agg.setdefault(key, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += a.elapsed_time(b)
            d[2] += host_s
        return {k: {"launches": v[0], "total_ms": v[1], "avg_us": 1e3 * v[1] / v[0], "host_avg_us": 1e6 * v[2] / v[0]}
                for k, v in agg.items()}


_PROF = None


def _timed(keyfn):
    def deco(fn):
        def wrapper(*a, **k):
            prof = _PROF
            if prof is None:
                return fn(*a, **k)
            if prof.only is not None and keyfn(*a, **k) not in prof.only:
                return fn(*a, **k)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            h0 = time.perf_counter()
            out = fn(*a, **k)
            h1 = time.perf_counter()
            e1.record()
            prof.records.append((keyfn(*a, **k), e0, e1, h1 - h0))
            return out
        wrapper.__name__ = fn.__name__
        wrapper.__doc__ = fn.__doc__
        return wrapper
    return deco


# ----------------------------------------------------------------------------------------------
@_timed(lambda x, w, *a, **k: ("gemm_nt", x.shape[0], w.shape[0], x.shape[1], k.get("epilogue", 0), bool(k.get("out_fp32", False))))
def gemm_nt(x, w, out=None, bias=None, epilogue=EPI_NONE, aux=None, aux_out=None, out_fp32=False):
    """out[M,N] = epilogue(x[M,K] @ w[N,K]^T)."""
    _req(x, BF16, "x"); _req(w, BF16, "w")
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise RuntimeError(f"gemm_nt: K mismatch {x.shape} vs {w.shape}")
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=F32 if out_fp32 else BF16)
    _req(out, F32 if out_fp32 else BF16, "out")
    if bias is not None:
        _req(bias, F32, "bias")
    ldaux = N
    if aux is not None:
        _req(aux, BF16, "aux")
    if aux_out is not None:
        _req(aux_out, BF16, "aux_out")
    rc = lib().chadavit_gemm_nt(_ptr(x), c_int(K), _ptr(w), c_int(K), _ptr(out), c_int(N), c_int(M), c_int(N), c_int(K),
                                _ptr(bias), c_int(epilogue), _ptr(aux), c_int(ldaux), _ptr(aux_out),
                                c_int(1 if out_fp32 else 0), _stream())
    _chk(rc, "chadavit_gemm_nt")
    return out


@_timed(lambda a, b, c, *r, **k: ("gemm_tn", a.shape[0] if k.get("t_rows") is None else k["t_rows"], a.shape[1], b.shape[1]))
def gemm_tn(a, b, c, colsum=None, accumulate=False, workspace=None, t_rows=None):
    """c[I,J] (+)= a[T,I]^T @ b[T,J]; colsum[I] (+)= a.sum(0)."""
    _req(a, BF16, "a"); _req(b, BF16, "b"); _req(c, F32, "c"); _req(workspace, F32, "workspace")
    T = a.shape[0] if t_rows is None else t_rows
    I, J = a.shape[1], b.shape[1]
    if colsum is not None:
        _req(colsum, F32, "colsum")
    rc = lib().chadavit_gemm_tn(_ptr(a), c_int(I), _ptr(b), c_int(J), _ptr(c), c_int(J), _ptr(colsum), c_int(T), c_int(I),
                                c_int(J), c_int(1 if accumulate else 0), _ptr(workspace), c_ll(workspace.numel()), _stream())
    _chk(rc, "chadavit_gemm_tn")
    return c


def ffn_packed_bytes(D, FF):
    return int(lib().chadavit_ffn_packed_bytes(c_int(D), c_int(FF)))


def ffn_pack(w1, w2, packed=None):
    """Fragment-major weight stream for ffn_fwd (rebuilt whenever W1 / W2 / b1 change)."""
    _req(w1, BF16, "w1"); _req(w2, BF16, "w2")
    FF, D = w1.shape
    n = ffn_packed_bytes(D, FF)
    if n < 0:
        raise RuntimeError(f"ffn_pack: unsupported shape D={D} FF={FF}")
    if packed is None:
        packed = torch.empty(n // 2, device=w1.device, dtype=BF16)
    _req(packed, BF16, "packed")
    _chk(lib().chadavit_ffn_pack(_ptr(w1), _ptr(w2), _ptr(packed), c_int(D), c_int(FF), _stream()), "chadavit_ffn_pack")
    return packed


def relu_bits_buffer(M, FF, device):
    """Caller-owned buffer for the ReLU pattern the fused forward records for the backward (M * FF / 8 bytes)."""
    n = int(lib().chadavit_relu_bits_bytes(c_int(M), c_int(FF)))
    if n < 0:
        raise RuntimeError(f"relu_bits: unsupported shape M={M} FF={FF}")
    return torch.empty(n, device=device, dtype=torch.uint8)


@_timed(lambda dz, packed_bwd, relu_bits, *a, **k: ("ffn_bwd_dx", dz.shape[0], dz.shape[1], (packed_bwd.numel() // (64 * dz.shape[1]) - 1) * 32, k.get("dpre") is not None))
def ffn_bwd_dx(dz, packed_bwd, relu_bits, dx1=None, dpre=None):
    """dx1 = dz + ((dz W2) * [H > 0]) W1 in one kernel (H > 0 from the forward's relu_bits); `dpre` (M, FF) optionally receives
    (dz W2) * [H > 0].  packed_bwd: FlatParams.ffn_packed_bwd."""
    _req(dz, BF16, "dz"); _req(packed_bwd, BF16, "packed_bwd"); _req(relu_bits, torch.uint8, "relu_bits")
    M, D = dz.shape
    FF = (packed_bwd.numel() // (64 * D) - 1) * 32
    if dx1 is None:
        dx1 = torch.empty((M, D), device=dz.device, dtype=BF16)
    _req(dx1, BF16, "dx1")
    if dpre is not None:
        _req(dpre, BF16, "dpre")
    if relu_bits.numel() < int(lib().chadavit_relu_bits_bytes(c_int(M), c_int(FF))):
        raise RuntimeError("ffn_bwd_dx: relu_bits buffer too small")
    rc = lib().chadavit_ffn_bwd_dx(_ptr(dz), c_int(dz.stride(0)), _ptr(packed_bwd), _ptr(relu_bits), _ptr(dx1), c_int(dx1.stride(0)),
                                   _ptr(dpre), c_int(dpre.stride(0) if dpre is not None else 0), c_int(M), c_int(D), c_int(FF), _stream())
    _chk(rc, "chadavit_ffn_bwd_dx")
    return dx1


@_timed(lambda x, packed, b1, b2, *a, **k: ("ffn_fwd", x.shape[0], x.shape[1], (packed.numel() // (64 * x.shape[1]) - 1) * 32, k.get("h") is not None))
def ffn_fwd(x, packed, b1, b2, resid=None, out=None, h=None, rows_per_wave=32, relu_bits=None):
    """out = resid + b2 + relu(x W1^T + b1) W2^T in one kernel; `h` (M, FF) receives relu(.) when given, `relu_bits` its sign
    pattern (relu_bits_buffer)."""
    _req(x, BF16, "x"); _req(packed, BF16, "packed"); _req(b1, F32, "b1"); _req(b2, F32, "b2")
    M, D = x.shape
    FF = (packed.numel() // (64 * D) - 1) * 32
    if out is None:
        out = torch.empty((M, D), device=x.device, dtype=BF16)
    _req(out, BF16, "out")
    if resid is not None:
        _req(resid, BF16, "resid")
    if h is not None:
        _req(h, BF16, "h")
    rc = lib().chadavit_ffn_fwd(_ptr(x), c_int(x.stride(0)), _ptr(packed), _ptr(b1), _ptr(b2), _ptr(resid), c_int(resid.stride(0) if resid is not None else 0),
                                _ptr(out), c_int(out.stride(0)), _ptr(h), c_int(h.stride(0) if h is not None else 0), _ptr(relu_bits),
                                c_int(M), c_int(D), c_int(FF), c_int(rows_per_wave), _stream())
    _chk(rc, "chadavit_ffn_fwd")
    return out


def channel_jitter_(x, shift, gamma, flip=None):
    """In place on the collated crop tensor x (n, 1, S, S) fp32: clamp(gamma_c * (x + shift_c), 0, 1), optional h-flip per c."""
    _req(x, F32, "x"); _req(shift, F32, "shift"); _req(gamma, F32, "gamma")
    if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != x.shape[3] or shift.numel() != x.shape[0] or gamma.numel() != x.shape[0]:
        raise RuntimeError("channel_jitter_: expected x (n, 1, S, S) and one shift / gamma per channel image")
    if flip is not None:
        _req(flip, torch.uint8, "flip")
    _chk(lib().chadavit_channel_jitter(_ptr(x), _ptr(shift), _ptr(gamma), _ptr(flip), c_int(x.shape[0]), c_int(x.shape[-1]), _stream()),
         "chadavit_channel_jitter")
    return x


@_timed(lambda x, packed, *a, **k: ("ffn_ln_fwd", x.shape[0], x.shape[1], (packed.numel() // (64 * x.shape[1]) - 1) * 32, k.get("h") is not None, k.get("ln_b") is not None))
def ffn_ln_fwd(x, packed, b1, b2, ln_a, resid=None, z=None, h=None, ln_b=None, stats_a=None, stats_b=None, relu_bits=None):
    """Fused FFN + LayerNorm tail: x2 = LN_a(z), hn = LN_b(x2) (if ln_b), z = resid + b2 + relu(x W1^T + b1) W2^T.
    ln_a / ln_b = (gamma, beta, eps); z / h are written only when given (backward needs them); returns (x2, hn or None)."""
    _req(x, BF16, "x"); _req(packed, BF16, "packed"); _req(b1, F32, "b1"); _req(b2, F32, "b2")
    M, D = x.shape
    FF = (packed.numel() // (64 * D) - 1) * 32
    x2 = torch.empty((M, D), device=x.device, dtype=BF16)
    hn = torch.empty((M, D), device=x.device, dtype=BF16) if ln_b is not None else None
    for t, nm in ((resid, "resid"), (z, "z"), (h, "h")):
        if t is not None:
            _req(t, BF16, nm)
    ga, ba, ea = ln_a
    gb, bb, eb = ln_b if ln_b is not None else (None, None, 0.0)
    _req(ga, F32, "gamma_a"); _req(ba, F32, "beta_a")
    sa = stats_a if stats_a is not None else (None, None)
    sb = stats_b if stats_b is not None else (None, None)
    rc = lib().chadavit_ffn_ln_fwd(_ptr(x), c_int(x.stride(0)), _ptr(packed), _ptr(b1), _ptr(b2), _ptr(resid),
                                   c_int(resid.stride(0) if resid is not None else 0), _ptr(z), c_int(z.stride(0) if z is not None else 0),
                                   _ptr(h), c_int(h.stride(0) if h is not None else 0), _ptr(ga), _ptr(ba), c_float(ea), _ptr(x2),
                                   _ptr(sa[0]), _ptr(sa[1]), _ptr(gb), _ptr(bb), c_float(eb), _ptr(hn), _ptr(sb[0]), _ptr(sb[1]),
                                   _ptr(relu_bits), c_int(M), c_int(D), c_int(FF), _stream())
    _chk(rc, "chadavit_ffn_ln_fwd")
    return x2, hn


def ffn_proj_packed_bytes(D, FF):
    return int(lib().chadavit_ffn_proj_packed_bytes(c_int(D), c_int(FF)))


def ffn_pack_proj_batched(slab, packed, desc, n_layers, D, FF):
    _req(slab, BF16, "slab"); _req(packed, BF16, "packed"); _req(desc, I64, "desc")
    _chk(lib().chadavit_ffn_pack_proj_batched(_ptr(slab), _ptr(packed), _ptr(desc), c_int(n_layers), c_int(D), c_int(FF), _stream()),
         "chadavit_ffn_pack_proj_batched")


@_timed(lambda a, xres, packed, *r, **k: ("proj_ffn_ln_fwd", a.shape[0], a.shape[1], (packed.numel() // (64 * a.shape[1]) - 1 - a.shape[1] // 16) * 32, k.get("h") is not None, k.get("ln_b") is not None, k.get("qkv_bias") is not None, k.get("relu_bits") is not None))
def proj_ffn_ln_fwd(a, xres, packed, bo, ln1, b1, b2, ln_a, y=None, x1=None, stats1=None, z=None, h=None, ln_b=None, stats_a=None,
                    stats_b=None, want_x1=True, qkv_bias=None, qkv=None, want_hn=True, relu_bits=None):
    """One block from the attention output on: y = xres + a Wo^T + bo; x1 = LN1(y); z = x1 + b2 + relu(x1 W1^T + b1) W2^T;
    x2 = LN_a(z); hn = LN_b(x2).  `packed` = [Wo | FFN] stream (FlatParams.proj_ffn_packed).  x1 is written only when wanted
    (the backward needs it; the kernel itself keeps it in registers).  With qkv_bias (and the next block's in_proj weight in
    the packed stream) the NEXT block's qkv = hn Wqkv^T + bias is produced too and hn is written only when wanted.
    Returns (x1 or None, x2, hn or None[, qkv])."""
    _req(a, BF16, "a"); _req(xres, BF16, "xres"); _req(packed, BF16, "packed")
    for t, nm in ((bo, "bo"), (b1, "b1"), (b2, "b2")):
        _req(t, F32, nm)
    M, D = a.shape
    FF = (packed.numel() // (64 * D) - 1 - D // 16) * 32   # stream = [D/64 Wo blocks | FF/32 + 1 FFN blocks | 3 D/64 next-QKV blocks] of 64 D elements
    dev = a.device
    if x1 is None and want_x1:
        x1 = torch.empty((M, D), device=dev, dtype=BF16)
    x2 = torch.empty((M, D), device=dev, dtype=BF16)
    hn = torch.empty((M, D), device=dev, dtype=BF16) if (ln_b is not None and (want_hn or qkv_bias is None)) else None
    if qkv_bias is not None:
        if ln_b is None:
            raise RuntimeError("proj_ffn_ln_fwd: the QKV postlogue needs ln_b (the next block's norm1)")
        _req(qkv_bias, F32, "qkv_bias")
        if qkv is None:
            qkv = torch.empty((M, 3 * D), device=dev, dtype=BF16)
        _req(qkv, BF16, "qkv")
    for t, nm in ((y, "y"), (x1, "x1"), (z, "z"), (h, "h")):
        if t is not None:
            _req(t, BF16, nm)
    g1, be1, e1 = ln1
    ga, ba, ea = ln_a
    gb, bb, eb = ln_b if ln_b is not None else (None, None, 0.0)
    s1 = stats1 if stats1 is not None else (None, None)
    sa = stats_a if stats_a is not None else (None, None)
    sb = stats_b if stats_b is not None else (None, None)
    rc = lib().chadavit_block_fwd(_ptr(a), c_int(a.stride(0)), _ptr(xres), c_int(xres.stride(0)), _ptr(packed), _ptr(bo), _ptr(g1),
                                        _ptr(be1), c_float(e1), _ptr(y), c_int(y.stride(0) if y is not None else 0), _ptr(x1),
                                        c_int(x1.stride(0) if x1 is not None else 0), _ptr(s1[0]), _ptr(s1[1]), _ptr(b1), _ptr(b2), _ptr(z),
                                        c_int(z.stride(0) if z is not None else 0), _ptr(h), c_int(h.stride(0) if h is not None else 0),
                                        _ptr(ga), _ptr(ba), c_float(ea), _ptr(x2), _ptr(sa[0]), _ptr(sa[1]), _ptr(gb), _ptr(bb),
                                        c_float(eb), _ptr(hn), _ptr(sb[0]), _ptr(sb[1]), _ptr(qkv if qkv_bias is not None else None),
                                        c_int(qkv.stride(0) if qkv_bias is not None else 0), _ptr(qkv_bias), _ptr(relu_bits), c_int(M),
                                        c_int(D), c_int(FF), _stream())
    _chk(rc, "chadavit_block_fwd")
    if qkv_bias is not None:
        return x1, x2, hn, qkv
    return x1, x2, hn


def im2col(x, patch, out=None):
    _req(x, F32, "x")
    n_chan, S = x.shape[0], x.shape[-1]
    g = S // patch
    if out is None:
        out = torch.empty((n_chan * g * g, patch * patch), device=x.device, dtype=BF16)
    _chk(lib().chadavit_im2col(_ptr(x), _ptr(out), c_int(n_chan), c_int(S), c_int(patch), _stream()), "chadavit_im2col")
    return out


def tokenizer_gemm(patches, wp, bias, pos, chan, chan_img, chan_idx, tokens, p):
    _req(patches, BF16, "patches"); _req(wp, BF16, "wp"); _req(bias, F32, "bias"); _req(pos, F32, "pos")
    _req(chan_img, I32, "chan_img"); _req(chan_idx, I32, "chan_idx"); _req(tokens, BF16, "tokens")
    if chan is not None:
        _req(chan, F32, "chan")
    Mp, K = patches.shape
    D = wp.shape[0]
    rc = lib().chadavit_tokenizer_gemm(_ptr(patches), _ptr(wp), _ptr(bias), _ptr(pos), _ptr(chan), _ptr(chan_img), _ptr(chan_idx),
                                       _ptr(tokens), c_int(Mp), c_int(D), c_int(K), c_int(p), _stream())
    _chk(rc, "chadavit_tokenizer_gemm")
    return tokens


def tokenizer_fused(x, wp, bias, pos, chan, chan_img, chan_idx, tokens, p):
    """Patch embed straight from the fp32 crops x (n_chan, S, S): conv unfold folded into the GEMM's operand staging."""
    _req(x, F32, "x"); _req(wp, BF16, "wp"); _req(bias, F32, "bias"); _req(pos, F32, "pos")
    _req(chan_img, I32, "chan_img"); _req(chan_idx, I32, "chan_idx"); _req(tokens, BF16, "tokens")
    if chan is not None:
        _req(chan, F32, "chan")
    n_chan, S = x.shape[0], x.shape[-1]
    D = wp.shape[0]
    if wp.shape[1] != 256:
        raise RuntimeError("tokenizer_fused: 16 x 16 patches only")
    rc = lib().chadavit_tokenizer_fused(_ptr(x), _ptr(wp), _ptr(bias), _ptr(pos), _ptr(chan), _ptr(chan_img), _ptr(chan_idx), _ptr(tokens),
                                        c_int(n_chan), c_int(S), c_int(D), c_int(p), _stream())
    _chk(rc, "chadavit_tokenizer_fused")
    return tokens


def write_cls(tokens, cu, cls, pos0):
    _req(tokens, BF16, "tokens"); _req(cu, I32, "cu"); _req(cls, F32, "cls"); _req(pos0, F32, "pos0")
    B, D = cu.numel() - 1, tokens.shape[1]
    _chk(lib().chadavit_write_cls(_ptr(tokens), _ptr(cu), _ptr(cls), _ptr(pos0), c_int(B), c_int(D), _stream()), "chadavit_write_cls")


def _mx_out(T, D, device):
    return torch.empty((T, D), device=device, dtype=U8), torch.empty((D // 32, (T + 3) // 4 * 4), device=device, dtype=U8)[:, :T]


@_timed(lambda x, *a, **k: ("layernorm_fwd", x.shape[0], x.shape[1]))
def layernorm_fwd(x, gamma, beta, eps, out=None, mean=None, rstd=None, emit_q=False):
    """emit_q (D in 192 / 384 / 768): returns (y, (bytes, scales)) with y also quantised as an OCP-MX fp8 operand (= mx8_quantize(y))."""
    _req(x, BF16, "x"); _req(gamma, F32, "gamma"); _req(beta, F32, "beta")
    T, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    if emit_q:
        yq, ys = _mx_out(T, D, x.device)
        rc = lib().chadavit_layernorm_fwd_q(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(mean), _ptr(rstd), _ptr(yq), _ptr(ys),
                                            c_int(ys.stride(0)), c_int(T), c_int(D), c_float(eps), _stream())
        _chk(rc, "chadavit_layernorm_fwd_q")
        return out, (yq, ys)
    rc = lib().chadavit_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(mean), _ptr(rstd), c_int(T), c_int(D),
                                      c_float(eps), _stream())
    _chk(rc, "chadavit_layernorm_fwd")
    return out


@_timed(lambda x, *a, **k: ("layernorm_fwd2", x.shape[0], x.shape[1]))
def layernorm_fwd2(x, ga, ba, gb, bb, eps_a, eps_b, stats1=None, stats2=None, emit_q=False):
    """y1 = LN_a(x); y2 = LN_b(y1) in one pass.  stats*: (mean, rstd) tensors or None.  emit_q: returns (y1, y2, (bytes, scales) of y2)."""
    _req(x, BF16, "x")
    T, D = x.shape
    y1 = torch.empty_like(x)
    y2 = torch.empty_like(x)
    m1, r1 = stats1 if stats1 is not None else (None, None)
    m2, r2 = stats2 if stats2 is not None else (None, None)
    if emit_q:
        yq, ys = _mx_out(T, D, x.device)
        rc = lib().chadavit_layernorm_fwd2_q(_ptr(x), _ptr(ga), _ptr(ba), _ptr(gb), _ptr(bb), _ptr(y1), _ptr(y2), _ptr(m1), _ptr(r1), _ptr(m2),
                                             _ptr(r2), _ptr(yq), _ptr(ys), c_int(ys.stride(0)), c_int(T), c_int(D), c_float(eps_a),
                                             c_float(eps_b), _stream())
        _chk(rc, "chadavit_layernorm_fwd2_q")
        return y1, y2, (yq, ys)
    rc = lib().chadavit_layernorm_fwd2(_ptr(x), _ptr(ga), _ptr(ba), _ptr(gb), _ptr(bb), _ptr(y1), _ptr(y2), _ptr(m1), _ptr(r1), _ptr(m2),
                                       _ptr(r2), c_int(T), c_int(D), c_float(eps_a), c_float(eps_b), _stream())
    _chk(rc, "chadavit_layernorm_fwd2")
    return y1, y2


def layernorm_bwd_workspace(D, device):
    """Partials of one (layernorm_bwd) or two (layernorm_bwd_pair) LayerNorm backward passes."""
    return torch.empty(2 * lib().chadavit_layernorm_bwd_partials() * 2 * D, device=device, dtype=F32)


LN_PAIR_WIDTHS = (192, 384, 768)


@_timed(lambda dy, x, *a, **k: ("layernorm_bwd_pair", x.shape[0], x.shape[1]))
def layernorm_bwd_pair(dy, x, mean_a, rstd_a, gamma_a, dres, z, mean_b, rstd_b, gamma_b, dgamma_a, dbeta_a, dgamma_b, dbeta_b, workspace,
                       accumulate_a=False, accumulate_b=False, dz=None, beta_b=None):
    """dz = LN_b'(LN_a'(dy; x) + dres; z) in one sweep (the boundary between two blocks in the backward); returns dz.  With `beta_b`
    the kernel rebuilds x = LN_b(z) itself instead of reading it (x is then only used for its shape)."""
    for t, n in ((dy, "dy"), (x, "x"), (dres, "dres"), (z, "z")):
        _req(t, BF16, n)
    if beta_b is not None:
        _req(beta_b, F32, "beta_b")
    for t, n in ((mean_a, "mean_a"), (rstd_a, "rstd_a"), (gamma_a, "gamma_a"), (mean_b, "mean_b"), (rstd_b, "rstd_b"), (gamma_b, "gamma_b"),
                 (dgamma_a, "dgamma_a"), (dbeta_a, "dbeta_a"), (dgamma_b, "dgamma_b"), (dbeta_b, "dbeta_b"), (workspace, "workspace")):
        _req(t, F32, n)
    T, D = x.shape
    if workspace.numel() < 2 * lib().chadavit_layernorm_bwd_partials() * 2 * D:
        raise RuntimeError("layernorm_bwd_pair: workspace too small (ops.layernorm_bwd_workspace)")
    if dz is None:
        dz = torch.empty_like(z)
    rc = lib().chadavit_layernorm_bwd_pair(_ptr(dy), _ptr(x if beta_b is None else None), _ptr(mean_a), _ptr(rstd_a), _ptr(gamma_a), _ptr(dres), _ptr(z),
                                           _ptr(mean_b), _ptr(rstd_b), _ptr(gamma_b), _ptr(beta_b), _ptr(dz), _ptr(dgamma_a), _ptr(dbeta_a),
                                           c_int(1 if accumulate_a else 0),
                                           _ptr(dgamma_b), _ptr(dbeta_b), c_int(1 if accumulate_b else 0), c_int(T), c_int(D), _ptr(workspace),
                                           _stream())
    _chk(rc, "chadavit_layernorm_bwd_pair")
    return dz


@_timed(lambda dy, x, *a, **k: ("layernorm_bwd", x.shape[0], x.shape[1]))
def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, workspace, dres=None, dx=None, accumulate=False):
    _req(dy, BF16, "dy"); _req(x, BF16, "x"); _req(mean, F32, "mean"); _req(rstd, F32, "rstd")
    _req(dgamma, F32, "dgamma"); _req(dbeta, F32, "dbeta"); _req(workspace, F32, "workspace")
    T, D = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    rc = lib().chadavit_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dres), _ptr(dx), _ptr(dgamma),
                                      _ptr(dbeta), c_int(1 if accumulate else 0), c_int(T), c_int(D), _ptr(workspace), _stream())
    _chk(rc, "chadavit_layernorm_bwd")
    return dx


@_timed(lambda qkv, cu, work, H, *a, **k: ("attn_fwd", qkv.shape[0], qkv.shape[1] // 3, H, work.shape[0]))
def attn_fwd(qkv, cu, work, H, out=None, lse=None):
    _req(qkv, BF16, "qkv"); _req(cu, I32, "cu"); _req(work, I32, "work")
    T, D3 = qkv.shape
    D = D3 // 3
    if out is None:
        out = torch.empty((T, D), device=qkv.device, dtype=BF16)
    if lse is None:
        lse = torch.empty((H, T), device=qkv.device, dtype=F32)
    rc = lib().chadavit_attn_fwd(_ptr(qkv), _ptr(out), _ptr(lse), _ptr(cu), _ptr(work), c_int(work.shape[0]), c_int(T), c_int(D),
                                 c_int(H), _stream())
    _chk(rc, "chadavit_attn_fwd")
    return out, lse


@_timed(lambda qkv, cu, H, *a, **k: ("attn_cls_fwd", qkv.shape[0], qkv.shape[1] // 3, H, cu.shape[0] - 1))
def attn_cls_fwd(qkv, cu, H):
    """Attention of the CLS row (first row) of every sequence against all keys of its sequence: (out_cls [B, D] bf16, lse_cls [H, B])."""
    _req(qkv, BF16, "qkv"); _req(cu, I32, "cu")
    T, D3 = qkv.shape
    D, B = D3 // 3, cu.shape[0] - 1
    out = torch.empty((B, D), device=qkv.device, dtype=BF16)
    lse = torch.empty((H, B), device=qkv.device, dtype=F32)
    rc = lib().chadavit_attn_cls_fwd(_ptr(qkv), _ptr(cu), _ptr(out), _ptr(lse), c_int(B), c_int(D), c_int(H),
                                     c_float(float(D // H) ** -0.5), _stream())
    _chk(rc, "chadavit_attn_cls_fwd")
    return out, lse


@_timed(lambda qkv, cu, out_cls, dout_cls, lse_cls, H, *a, **k: ("attn_cls_bwd", qkv.shape[0], qkv.shape[1] // 3, H, cu.shape[0] - 1))
def attn_cls_bwd(qkv, cu, out_cls, dout_cls, lse_cls, H, dqkv=None):
    """dqkv [T, 3D] (every element written) for a loss that reads only the CLS rows of the attention output."""
    _req(qkv, BF16, "qkv"); _req(cu, I32, "cu"); _req(out_cls, BF16, "out_cls"); _req(dout_cls, BF16, "dout_cls"); _req(lse_cls, F32, "lse_cls")
    T, D3 = qkv.shape
    D, B = D3 // 3, cu.shape[0] - 1
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    _req(dqkv, BF16, "dqkv")
    rc = lib().chadavit_attn_cls_bwd(_ptr(qkv), _ptr(cu), _ptr(out_cls), _ptr(dout_cls), _ptr(lse_cls), _ptr(dqkv), c_int(B), c_int(D),
                                     c_int(H), c_float(float(D // H) ** -0.5), _stream())
    _chk(rc, "chadavit_attn_cls_bwd")
    return dqkv


@_timed(lambda qkv, out, dout, lse, cu, work, H, *a, **k: ("attn_bwd", qkv.shape[0], qkv.shape[1] // 3, H, work.shape[0]))
def attn_bwd(qkv, out, dout, lse, cu, work, H, dqkv=None, delta=None, side=None):
    """dqkv from dout.  With `side` (a HIP stream) the dK/dV kernel runs there while dQ runs on the current stream
    (they are independent once delta is known); the current stream waits for `side` before returning."""
    _req(qkv, BF16, "qkv"); _req(out, BF16, "out"); _req(dout, BF16, "dout"); _req(lse, F32, "lse")
    T, D3 = qkv.shape
    D = D3 // 3
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if delta is None:
        delta = torch.empty((H, T), device=qkv.device, dtype=F32)

    if D // H == 16:  # 12-head default constructor: widened-head path, caller-owned workspace
        ws = torch.empty(int(lib().chadavit_attn_bwd_dh16_workspace_bytes(c_int(T), c_int(H))), device=qkv.device, dtype=torch.uint8)
        rc = lib().chadavit_attn_bwd_dh16(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), _ptr(delta), _ptr(cu), _ptr(work),
                                          c_int(work.shape[0]), c_int(T), c_int(D), c_int(H), _ptr(ws), c_ll(ws.numel()), _stream())
        _chk(rc, "chadavit_attn_bwd_dh16")
        return dqkv

    def call(parts, stream):
        rc = lib().chadavit_attn_bwd_parts(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), _ptr(delta), _ptr(cu), _ptr(work),
                                           c_int(work.shape[0]), c_int(T), c_int(D), c_int(H), c_int(parts), stream)
        _chk(rc, "chadavit_attn_bwd_parts")

    if side is None:
        call(7, _stream())
        return dqkv
    main = torch.cuda.current_stream()
    call(1, _stream())
    side.wait_stream(main)
    call(4, side.cuda_stream)
    call(2, _stream())
    # every later use / free / reuse of these tensors happens on `main` after this wait -> no record_stream needed
    main.wait_stream(side)
    return dqkv


def attn_probs(qkv, cu, lens, H):
    """Per-image softmax(QK^T/sqrt(dh)) matrices.  Equal-length batch -> (B, H, N, N) fp32; ragged -> list of (H, N_i, N_i)."""
    _req(qkv, BF16, "qkv"); _req(cu, I32, "cu_seqlens")
    T, D3 = qkv.shape
    D = D3 // 3
    B = len(lens)
    sizes = [H * n * n for n in lens]
    offs = [0]
    for sz in sizes[:-1]:
        offs.append(offs[-1] + sz)
    probs = torch.empty(sum(sizes), device=qkv.device, dtype=F32)
    offs_t = torch.tensor(offs, dtype=I64, device=qkv.device)
    rc = lib().chadavit_attn_probs(_ptr(qkv), _ptr(probs), _ptr(cu), _ptr(offs_t), c_int(B), c_int(T), c_int(D), c_int(H),
                                   c_int(max(lens)), _stream())
    _chk(rc, "chadavit_attn_probs")
    if len(set(lens)) == 1:
        return probs.view(B, H, lens[0], lens[0])
    return [probs[o:o + sz].view(H, n, n) for o, sz, n in zip(offs, sizes, lens)]


def gather_rows(src, rows, out=None):
    _req(src, BF16, "src"); _req(rows, I32, "rows")
    n, D = rows.numel(), src.shape[1]
    if out is None:
        out = torch.empty((n, D), device=src.device, dtype=BF16)
    _chk(lib().chadavit_gather_rows(_ptr(src), _ptr(rows), _ptr(out), c_int(n), c_int(D), _stream()), "chadavit_gather_rows")
    return out


def scatter_rows_zero(src, rows, T, out=None):
    _req(src, BF16, "src"); _req(rows, I32, "rows")
    n, D = src.shape
    if out is None:
        out = torch.empty((T, D), device=src.device, dtype=BF16)
    _chk(lib().chadavit_scatter_rows_zero(_ptr(src), _ptr(rows), _ptr(out), c_int(n), c_int(T), c_int(D), _stream()),
         "chadavit_scatter_rows_zero")
    return out


def tokenizer_bwd(dtok, cu, chan_img, chan_idx, p, max_channels):
    _req(dtok, BF16, "dtok")
    B, n_chan, D = cu.numel() - 1, chan_img.numel(), dtok.shape[1]
    dev = dtok.device
    dpatch = torch.empty((n_chan * p, D), device=dev, dtype=BF16)
    dpos = torch.empty((p, D), device=dev, dtype=F32)
    dchan = torch.empty((max_channels, D), device=dev, dtype=F32)
    dcls = torch.empty((D,), device=dev, dtype=F32)
    ws = torch.empty(int(lib().chadavit_tokenizer_bwd_workspace_floats(c_int(p), c_int(D), c_int(max_channels))), device=dev, dtype=F32)
    rc = lib().chadavit_tokenizer_bwd(_ptr(dtok), _ptr(cu), _ptr(chan_img), _ptr(chan_idx), _ptr(dpatch), _ptr(dpos), _ptr(dchan),
                                      _ptr(dcls), _ptr(ws), c_int(B), c_int(n_chan), c_int(p), c_int(D), c_int(max_channels), _stream())
    _chk(rc, "chadavit_tokenizer_bwd")
    return dpatch, dpos, dchan, dcls


def l2norm_fwd(x):
    _req(x, F32, "x")
    M, N = x.shape
    y = torch.empty((M, N), device=x.device, dtype=BF16)
    inv = torch.empty((M,), device=x.device, dtype=F32)
    _chk(lib().chadavit_l2norm_fwd(_ptr(x), _ptr(y), _ptr(inv), c_int(M), c_int(N), _stream()), "chadavit_l2norm_fwd")
    return y, inv


def l2norm_bwd(dy, x, inv):
    _req(dy, F32, "dy"); _req(x, F32, "x"); _req(inv, F32, "inv")
    M, N = x.shape
    dx = torch.empty((M, N), device=x.device, dtype=BF16)
    _chk(lib().chadavit_l2norm_bwd(_ptr(dy), _ptr(x), _ptr(inv), _ptr(dx), c_int(M), c_int(N), _stream()), "chadavit_l2norm_bwd")
    return dx


def weightnorm_fwd(v, g, w=None, w_t=None, inv=None):
    _req(v, F32, "v"); _req(g, F32, "g")
    P, K = v.shape
    dev = v.device
    if w is None:
        w = torch.empty((P, K), device=dev, dtype=BF16)
    if w_t is None:
        w_t = torch.empty((K, P), device=dev, dtype=BF16)
    if inv is None:
        inv = torch.empty((P,), device=dev, dtype=F32)
    _chk(lib().chadavit_weightnorm_fwd(_ptr(v), _ptr(g), _ptr(w), _ptr(w_t), _ptr(inv), c_int(P), c_int(K), _stream()),
         "chadavit_weightnorm_fwd")
    return w, w_t, inv


def weightnorm_bwd(dw, v, g, inv, dv, accumulate=False, dg=None):
    """dv (+)= the gradient w.r.t. the directions v; dg (optional, [P]) (+)= the gradient w.r.t. the magnitudes g."""
    _req(dw, F32, "dw"); _req(v, F32, "v"); _req(dv, F32, "dv")
    if dg is not None:
        _req(dg, F32, "dg")
    P, K = v.shape
    _chk(lib().chadavit_weightnorm_bwd_g(_ptr(dw), _ptr(v), _ptr(g), _ptr(inv), _ptr(dv), _ptr(dg), c_int(1 if accumulate else 0), c_int(P),
                                         c_int(K), _stream()), "chadavit_weightnorm_bwd_g")
    return dv


def dino_loss(student, teacher, center, student_temp, teacher_temp, want_grad=True):
    """Returns (loss_rows [B], dstudent bf16 [V B,P] | None, teacher_colsum [P]).  teacher_temp: a float, or a device float32[1]
    tensor (read by the kernel: graph-captured training step).  student [V B, P] with V = 2 (the reference's DINO: global crops only)
    or V > 2 (standard-DINO multi-crop option: the local crops' logits follow the two global views)."""
    _req(student, F32, "student"); _req(teacher, F32, "teacher"); _req(center, F32, "center")
    B2, P = student.shape
    B = teacher.shape[0] // 2
    if teacher.shape[0] != 2 * B or B2 % B != 0 or B2 < 2 * B:
        raise RuntimeError(f"dino_loss: student rows {B2} are not a whole number (>= 2) of views of the teacher's {B} images")
    V = B2 // B
    dev = student.device
    loss_rows = torch.empty((B,), device=dev, dtype=F32)
    dstudent = torch.empty((B2, P), device=dev, dtype=BF16) if want_grad else None
    colsum = torch.empty((P,), device=dev, dtype=F32)
    if V > 2:
        if isinstance(teacher_temp, torch.Tensor):
            raise RuntimeError("dino_loss: the multi-crop form takes the teacher temperature by value (not captured in hipGraphs)")
        rc = lib().chadavit_dino_loss_multicrop(_ptr(student), _ptr(teacher), _ptr(center), c_float(student_temp), c_float(teacher_temp),
                                                _ptr(loss_rows), _ptr(dstudent), _ptr(colsum), c_int(B), c_int(V), c_int(P), _stream())
        _chk(rc, "chadavit_dino_loss_multicrop")
        return loss_rows, dstudent, colsum
    if isinstance(teacher_temp, torch.Tensor):
        _req(teacher_temp, F32, "teacher_temp")
        rc = lib().chadavit_dino_loss_dev(_ptr(student), _ptr(teacher), _ptr(center), c_float(student_temp), _ptr(teacher_temp),
                                          _ptr(loss_rows), _ptr(dstudent), _ptr(colsum), c_int(B), c_int(P), _stream())
        _chk(rc, "chadavit_dino_loss_dev")
        return loss_rows, dstudent, colsum
    rc = lib().chadavit_dino_loss(_ptr(student), _ptr(teacher), _ptr(center), c_float(student_temp), c_float(teacher_temp),
                                  _ptr(loss_rows), _ptr(dstudent), _ptr(colsum), c_int(B), c_int(P), _stream())
    _chk(rc, "chadavit_dino_loss")
    return loss_rows, dstudent, colsum


def center_ema(center, colsum, inv_count, momentum):
    _req(center, F32, "center"); _req(colsum, F32, "colsum")
    _chk(lib().chadavit_center_ema(_ptr(center), _ptr(colsum), c_float(inv_count), c_float(momentum), c_int(center.numel()), _stream()),
         "chadavit_center_ema")


def sum_rows_f32(x, scale=1.0):
    _req(x, F32, "x")
    out = torch.empty((x.shape[1],), device=x.device, dtype=F32)
    _chk(lib().chadavit_sum_rows_f32(_ptr(x), _ptr(out), c_int(x.shape[0]), c_int(x.shape[1]), c_float(scale), _stream()),
         "chadavit_sum_rows_f32")
    return out


def _req_z(z):
    if z.dtype == F32:
        _req(z, F32, "z")
    else:
        _req(z, BF16, "z")


@_timed(lambda z, *a, **k: ("bn_stats", z.shape[0], z.shape[1]))
def bn_stats(z, eps, running_mean=None, running_var=None, momentum=0.1):
    """Column statistics of z (N, C) bf16 for BatchNorm1d in training mode: (mean, rstd); the running estimates are updated in place
    when given (torch.nn.BatchNorm1d: momentum, unbiased variance)."""
    _req_z(z)
    N, C = z.shape
    mean = torch.empty((C,), device=z.device, dtype=F32)
    rstd = torch.empty((C,), device=z.device, dtype=F32)
    ws = torch.empty((2 * C,), device=z.device, dtype=F32)
    for t, nm in ((running_mean, "running_mean"), (running_var, "running_var")):
        if t is not None:
            _req(t, F32, nm)
    _chk(lib().chadavit_bn_stats(_ptr(z), c_int(int(z.dtype == F32)), c_int(N), c_int(C), c_float(eps), _ptr(mean), _ptr(rstd), _ptr(running_mean), _ptr(running_var),
                                 c_float(momentum), _ptr(ws), _stream()), "chadavit_bn_stats")
    return mean, rstd


def bn_apply_gelu(z, mean, rstd, gamma, beta):
    """(pre, act): pre = (z - mean) * rstd * gamma + beta, act = gelu(pre); both bf16 (N, C)."""
    _req_z(z)
    for t, nm in ((mean, "mean"), (rstd, "rstd"), (gamma, "gamma"), (beta, "beta")):
        _req(t, F32, nm)
    N, C = z.shape
    pre = torch.empty((N, C), device=z.device, dtype=BF16)
    act = torch.empty((N, C), device=z.device, dtype=BF16)
    _chk(lib().chadavit_bn_apply_gelu(_ptr(z), c_int(int(z.dtype == F32)), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(pre), _ptr(act), c_int(N), c_int(C),
                                      _stream()), "chadavit_bn_apply_gelu")
    return pre, act


@_timed(lambda dy, z, *a, **k: ("bn_bwd", z.shape[0], z.shape[1]))
def bn_bwd(dy, z, mean, rstd, gamma, dgamma, dbeta, accumulate=False):
    """BatchNorm1d backward (training-mode statistics): dgamma / dbeta written (or added to) in place, returns dz (bf16)."""
    _req(dy, BF16, "dy"); _req_z(z)
    for t, nm in ((mean, "mean"), (rstd, "rstd"), (gamma, "gamma"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _req(t, F32, nm)
    N, C = z.shape
    dz = torch.empty((N, C), device=z.device, dtype=BF16)
    ws = torch.empty((2 * C,), device=z.device, dtype=F32)
    _chk(lib().chadavit_bn_bwd(_ptr(dy), _ptr(z), c_int(int(z.dtype == F32)), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dgamma), _ptr(dbeta), c_int(1 if accumulate else 0),
                               _ptr(dz), c_int(N), c_int(C), _ptr(ws), _stream()), "chadavit_bn_bwd")
    return dz


def ema_update(teacher, student, tau):
    _req(teacher, F32, "teacher"); _req(student, F32, "student")
    _chk(lib().chadavit_ema_update(_ptr(teacher), _ptr(student), c_float(tau), c_ll(teacher.numel()), _stream()), "chadavit_ema_update")


def ema_update_dev(teacher, student, tau_dev):
    """tau read from device memory (one float): for a graph-captured training step."""
    _req(teacher, F32, "teacher"); _req(student, F32, "student"); _req(tau_dev, F32, "tau_dev")
    _chk(lib().chadavit_ema_update_dev(_ptr(teacher), _ptr(student), _ptr(tau_dev), c_ll(teacher.numel()), _stream()), "chadavit_ema_update_dev")


def adamw_step_dev(param, grad, exp_avg, exp_avg_sq, hyper_dev, beta1, beta2, eps, weight_decay):
    """hyper_dev: device float32[3] = {lr, 1 - beta1^t, sqrt(1 - beta2^t)} of this step."""
    _req(param, F32, "param"); _req(grad, F32, "grad"); _req(exp_avg, F32, "exp_avg"); _req(exp_avg_sq, F32, "exp_avg_sq")
    _req(hyper_dev, F32, "hyper_dev")
    rc = lib().chadavit_adamw_step_dev(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(hyper_dev), c_float(beta1),
                                       c_float(beta2), c_float(eps), c_float(weight_decay), c_ll(param.numel()), _stream())
    _chk(rc, "chadavit_adamw_step_dev")


def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step):
    _req(param, F32, "param"); _req(grad, F32, "grad"); _req(exp_avg, F32, "exp_avg"); _req(exp_avg_sq, F32, "exp_avg_sq")
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    rc = lib().chadavit_adamw_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), c_float(lr), c_float(beta1),
                                   c_float(beta2), c_float(eps), c_float(weight_decay), c_float(bc1), c_float(bc2),
                                   c_ll(param.numel()), _stream())
    _chk(rc, "chadavit_adamw_step")


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step):
    """torch.optim.Adam's update (L2 weight decay inside the gradient)."""
    _req(param, F32, "param"); _req(grad, F32, "grad"); _req(exp_avg, F32, "exp_avg"); _req(exp_avg_sq, F32, "exp_avg_sq")
    rc = lib().chadavit_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), c_float(lr), c_float(beta1), c_float(beta2),
                                  c_float(eps), c_float(weight_decay), c_float(1.0 - beta1 ** step), c_float(1.0 - beta2 ** step),
                                  c_ll(param.numel()), _stream())
    _chk(rc, "chadavit_adam_step")


def sgd_step(param, grad, buf, lr, momentum, dampening, weight_decay, nesterov, first):
    """torch.optim.SGD's update; buf (momentum buffer) may be None when momentum == 0."""
    _req(param, F32, "param"); _req(grad, F32, "grad")
    if buf is not None:
        _req(buf, F32, "buf")
    rc = lib().chadavit_sgd_step(_ptr(param), _ptr(grad), _ptr(buf), c_float(lr), c_float(momentum), c_float(dampening),
                                 c_float(weight_decay), c_int(1 if nesterov else 0), c_int(1 if first else 0), c_ll(param.numel()), _stream())
    _chk(rc, "chadavit_sgd_step")


def lars_step(params, grads, bufs, offsets, sizes, flags, lr, momentum, dampening, weight_decay, eta, eps, clip_lr, nesterov):
    _req(params, F32, "params"); _req(grads, F32, "grads"); _req(bufs, F32, "bufs")
    _req(offsets, I64, "offsets"); _req(sizes, I64, "sizes"); _req(flags, I32, "flags")
    rc = lib().chadavit_lars_step(_ptr(params), _ptr(grads), _ptr(bufs), _ptr(offsets), _ptr(sizes), _ptr(flags), c_int(offsets.numel()),
                                  c_float(lr), c_float(momentum), c_float(dampening), c_float(weight_decay), c_float(eta), c_float(eps),
                                  c_int(1 if clip_lr else 0), c_int(1 if nesterov else 0), _stream())
    _chk(rc, "chadavit_lars_step")


def knn_vote(sims, train_targets, k, temperature, use_exp, num_classes, top, want_votes=False):
    """top classes (n_test, top) int32 by weighted k-NN vote over the rows of `sims` (n_test, n_train) fp32."""
    _req(sims, F32, "sims"); _req(train_targets, I32, "train_targets")
    n_test, n_train = sims.shape
    out = torch.empty((n_test, top), device=sims.device, dtype=I32)
    votes = torch.empty((n_test, num_classes), device=sims.device, dtype=F32) if want_votes else None
    rc = lib().chadavit_knn_vote(_ptr(sims), c_ll(sims.stride(0)), _ptr(train_targets), c_int(n_test), c_int(n_train), c_int(k),
                                 c_float(temperature), c_int(1 if use_exp else 0), c_int(num_classes), c_int(top), _ptr(out),
                                 _ptr(votes), _stream())
    _chk(rc, "chadavit_knn_vote")
    return (out, votes) if want_votes else out


def cast_bf16(src, dst):
    _req(src, F32, "src"); _req(dst, BF16, "dst")
    _chk(lib().chadavit_cast_bf16(_ptr(src), _ptr(dst), c_ll(src.numel()), _stream()), "chadavit_cast_bf16")


def cast_transpose_bf16(src, dst, dst_t):
    _req(src, F32, "src"); _req(dst_t, BF16, "dst_t")
    rows, cols = src.shape
    _chk(lib().chadavit_cast_transpose_bf16(_ptr(src), _ptr(dst), _ptr(dst_t), c_int(rows), c_int(cols), _stream()),
         "chadavit_cast_transpose_bf16")


def cast_transpose_batched(src, dst_t, desc, n_mats, max_tiles):
    _req(src, F32, "src"); _req(dst_t, BF16, "dst_t"); _req(desc, I64, "desc")
    _chk(lib().chadavit_cast_transpose_batched(_ptr(src), _ptr(dst_t), _ptr(desc), c_int(n_mats), c_int(max_tiles), _stream()),
         "chadavit_cast_transpose_batched")


def ffn_pack_batched(slab, packed, desc, n_layers, D, FF):
    _req(slab, BF16, "slab"); _req(packed, BF16, "packed"); _req(desc, I64, "desc")
    _chk(lib().chadavit_ffn_pack_batched(_ptr(slab), _ptr(packed), _ptr(desc), c_int(n_layers), c_int(D), c_int(FF), _stream()),
         "chadavit_ffn_pack_batched")


def clip_tensors(grads, offsets, sizes, clip):
    _req(grads, F32, "grads"); _req(offsets, I64, "offsets"); _req(sizes, I64, "sizes")
    _chk(lib().chadavit_clip_tensors(_ptr(grads), _ptr(offsets), _ptr(sizes), c_int(offsets.numel()), c_float(clip), _stream()),
         "chadavit_clip_tensors")


# ---- fp8 (OCP MX) weight path: BASELINE configs[4] ---------------------------------------------------------------------------
U8 = torch.uint8


def mx8_quantize(x, relu=False, q=None, scales=None):
    """bf16 (R, K) -> (e4m3 bytes (R, K) uint8, E8M0 scale bytes (K/32, R) uint8 -- a view of a (K/32, ceil4(R)) buffer): one
    power-of-two scale per row and 32-k block."""
    _req(x, BF16, "x")
    R, K = x.shape
    if K % 32 != 0:
        raise RuntimeError("mx8_quantize: K must be a multiple of 32")
    if q is None:
        q = torch.empty((R, K), device=x.device, dtype=U8)
    if scales is None:
        scales = torch.empty((K // 32, (R + 3) // 4 * 4), device=x.device, dtype=U8)[:, :R]
    _req(q, U8, "q")
    if scales.dtype != U8 or scales.device != x.device or scales.stride(1) != 1 or tuple(scales.shape) != (K // 32, R):
        raise RuntimeError("mx8_quantize: scales must be a (K/32, R) uint8 view with unit column stride")
    _chk(lib().chadavit_mx8_quantize(_ptr(x), c_int(x.stride(0)), _ptr(q), _ptr(scales), c_int(scales.stride(0)), c_int(R), c_int(K),
                                     c_int(1 if relu else 0), _stream()), "chadavit_mx8_quantize")
    return q, scales


@_timed(lambda xq, xs, wq, ws, *a, **k: ("gemm_nt_mx8", xq.shape[0], wq.shape[0], xq.shape[1], k.get("epilogue", 0)))
def gemm_nt_mx8(xq, xs, wq, ws, bias=None, epilogue=EPI_NONE, aux=None, out=None, emit_q=False, want_out=True):
    """out[M,N] bf16 = epilogue(deq(xq, xs) @ deq(wq, ws)^T + bias) on the MX-scaled fp8 MFMA (epilogues NONE / RELU / RESID / RELUMASK).
    emit_q: also return (q, scales) = mx8_quantize(out) produced by the epilogue itself (the next GEMM's operand); with want_out=False
    the bf16 result is not written at all and None is returned in its place."""
    _req(xq, U8, "xq"); _req(wq, U8, "wq")
    M, K = xq.shape
    N = wq.shape[0]
    for t, nm, rows in ((xs, "xs", M), (ws, "ws", N)):
        if t.dtype != U8 or t.device != xq.device or t.stride(1) != 1 or tuple(t.shape) != (K // 32, rows):
            raise RuntimeError(f"gemm_nt_mx8: {nm} must be the (K/32, rows) uint8 scale view mx8_quantize returns")
    if wq.shape[1] != K:
        raise RuntimeError("gemm_nt_mx8: operand shapes do not match")
    if out is None and (want_out or not emit_q):
        out = torch.empty((M, N), device=xq.device, dtype=BF16)
    if out is not None:
        _req(out, BF16, "out")
    if bias is not None:
        _req(bias, F32, "bias")
    if aux is not None:
        _req(aux, BF16, "aux")
    if not emit_q:
        rc = lib().chadavit_gemm_nt_mx8(_ptr(xq), _ptr(xs), c_int(xs.stride(0)), _ptr(wq), _ptr(ws), c_int(ws.stride(0)), _ptr(out),
                                        c_int(out.stride(0)), c_int(M), c_int(N), c_int(K), _ptr(bias), c_int(epilogue), _ptr(aux),
                                        c_int(aux.stride(0) if aux is not None else 0), _stream())
        _chk(rc, "chadavit_gemm_nt_mx8")
        return out
    if N % 32 != 0:
        raise RuntimeError("gemm_nt_mx8: emit_q needs N % 32 == 0")
    oq = torch.empty((M, N), device=xq.device, dtype=U8)
    osc = torch.empty((N // 32, (M + 3) // 4 * 4), device=xq.device, dtype=U8)[:, :M]
    rc = lib().chadavit_gemm_nt_mx8_q(_ptr(xq), _ptr(xs), c_int(xs.stride(0)), _ptr(wq), _ptr(ws), c_int(ws.stride(0)), _ptr(out),
                                      c_int(out.stride(0) if out is not None else 0), _ptr(oq), _ptr(osc), c_int(osc.stride(0)), c_int(M), c_int(N),
                                      c_int(K), _ptr(bias), c_int(epilogue), _ptr(aux), c_int(aux.stride(0) if aux is not None else 0), _stream())
    _chk(rc, "chadavit_gemm_nt_mx8_q")
    return out, (oq, osc)


# ---- device augmentation ---------------------------------------------------------------------------------------
_SRC_KINDS = {torch.float32: 0, torch.uint8: 1, torch.uint16: 2}   # src_kind of chadavit_crop_resize_src


def crop_resize(src, desc, S, shift=None, gamma=None, out=None):
    """src: packed source planes, fp32 or as the image files store them (uint8 / uint16: converted on the device, exactly -- the reference
    reader's `.astype(np.float32)`); desc (n, 8) int64 {element offset, H, W, x0, y0, cw, ch, flip}; -> (n, 1, S, S) fp32 crops
    (bicubic as cv2.INTER_CUBIC, optional per-channel jitter clamp(gamma (x + shift), 0, 1), optional h-flip)."""
    kind = _SRC_KINDS.get(src.dtype)
    if kind is None:
        raise RuntimeError(f"crop_resize: source planes must be float32, uint8 or uint16, got {src.dtype}")
    _req(src, src.dtype, "src"); _req(desc, I64, "desc")
    n = desc.shape[0]
    if desc.dim() != 2 or desc.shape[1] != 8:
        raise RuntimeError("crop_resize: desc must be (n, 8) int64")
    if out is None:
        out = torch.empty((n, 1, S, S), device=src.device, dtype=F32)
    _req(out, F32, "out")
    if shift is not None:
        _req(shift, F32, "shift"); _req(gamma, F32, "gamma")
    _chk(lib().chadavit_crop_resize_src(_ptr(src), c_int(kind), _ptr(desc), _ptr(shift), _ptr(gamma), _ptr(out), c_int(n), c_int(S), _stream()),
         "chadavit_crop_resize_src")
    return out


def blur_finish(x, fin, out=None):
    """GaussianBlur -> Solarize -> Normalize per channel image of x (n, 1, S, S); fin (n, 12) fp32 (see include/chadavit_hip.h)."""
    _req(x, F32, "x"); _req(fin, F32, "fin")
    n, S = x.shape[0], x.shape[-1]
    if tuple(fin.shape) != (n, 12):
        raise RuntimeError("blur_finish: fin must be (n, 12) fp32")
    if out is None:
        out = torch.empty_like(x)
    _chk(lib().chadavit_blur_finish(_ptr(x), _ptr(fin), _ptr(out), c_int(n), c_int(S), _stream()), "chadavit_blur_finish")
    return out
