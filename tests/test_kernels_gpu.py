"""Per-kernel parity on a real MI355X: every C-ABI entry point against a plain PyTorch fp32 reference
of the same op on the same seeded inputs.  bf16 operands / fp32 accumulate -> tolerances are relative
to the fp32 result computed from the SAME bf16-rounded inputs (so only accumulation order differs)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def _close(a, b, rtol, atol, name=""):
    a = a.float().cpu()
    b = b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol)
    assert not bad.any(), f"{name}: {int(bad.sum())}/{bad.numel()} mismatches, max err {err.max().item():.4g} " \
                          f"(ref max {b.abs().max().item():.4g})"


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 192, 192), (257, 576, 192), (1000, 2048, 192), (130, 192, 2048), (64, 256, 2048),
                                   (128, 4096, 256), (5, 64, 64), (513, 1152, 384)])
def test_gemm_nt_epilogues(M, N, K):
    from chadavit_amd import ops
    dev = _dev()
    x = _rand((M, K), 1, 1.0).bfloat16().to(dev)
    w = _rand((N, K), 2, 1.0 / math.sqrt(K)).bfloat16().to(dev)
    bias = _rand((N,), 3, 0.5).to(dev)
    aux = _rand((M, N), 4, 1.0).bfloat16().to(dev)
    ref = x.float() @ w.float().t() + bias
    out = ops.gemm_nt(x, w, bias=bias)
    _close(out, ref, 1e-2, 1e-2, "none")
    out = ops.gemm_nt(x, w, bias=None, out_fp32=True)
    _close(out, x.float() @ w.float().t(), 1e-4, 1e-4, "fp32 out")
    out = ops.gemm_nt(x, w, bias=bias, epilogue=ops.EPI_RELU)
    _close(out, torch.relu(ref), 1e-2, 1e-2, "relu")
    pre = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    out = ops.gemm_nt(x, w, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    _close(out, torch.nn.functional.gelu(ref), 1e-2, 1e-2, "gelu")
    _close(pre, ref, 1e-2, 1e-2, "gelu pre-activation")
    out = ops.gemm_nt(x, w, bias=bias, epilogue=ops.EPI_RESID, aux=aux)
    _close(out, ref + aux.float(), 1e-2, 2e-2, "resid")
    out = ops.gemm_nt(x, w, bias=None, epilogue=ops.EPI_RELUMASK, aux=aux)
    _close(out, (x.float() @ w.float().t()) * (aux.float() > 0), 1e-2, 1e-2, "relumask")
    out = ops.gemm_nt(x, w, bias=None, epilogue=ops.EPI_GELUBWD, aux=aux)
    a = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(a).sum().backward()
    _close(out, (x.float() @ w.float().t()) * a.grad, 1e-2, 1e-2, "gelubwd")


@pytest.mark.parametrize("T,I,J,ws_splits", [(1000, 2048, 192, 16), (777, 192, 2048, 16), (300, 576, 192, 16), (256, 192, 192, 16),
                                             (130, 256, 2048, 16), (64, 4096, 256, 16), (5000, 192, 256, 16), (900, 1152, 384, 16),
                                             (333, 128, 64, 16),
                                             # the split rule of round 3: multiples of 8 that fill whole rounds of an XCD (many rows), the
                                             # 192 x 192 tile of wide outputs, a workspace with room for fewer than eight splits, for one
                                             (40000, 768, 768, 64), (20001, 2304, 768, 33), (9000, 384, 384, 64), (3000, 1152, 384, 5),
                                             (1024, 4096, 256, 1), (70000, 576, 192, 200)])
def test_gemm_tn(T, I, J, ws_splits):
    from chadavit_amd import ops
    dev = _dev()
    a = _rand((T, I), 5, 1.0).bfloat16().to(dev)
    b = _rand((T, J), 6, 1.0).bfloat16().to(dev)
    ws = torch.empty(ws_splits * (I * J + I), device=dev)
    c = torch.full((I, J), 7.0, device=dev)
    cs = torch.full((I,), 3.0, device=dev)
    ops.gemm_tn(a, b, c, colsum=cs, accumulate=False, workspace=ws)
    ref = a.float().t() @ b.float()
    _close(c, ref, 2e-3, 2e-2 * math.sqrt(T / 64), "tn")
    _close(cs, a.float().sum(0), 2e-3, 2e-2, "colsum")
    ops.gemm_tn(a, b, c, colsum=cs, accumulate=True, workspace=ws)
    _close(c, 2 * ref, 2e-3, 4e-2 * math.sqrt(T / 64), "tn accumulate")
    _close(cs, 2 * a.float().sum(0), 2e-3, 4e-2, "colsum accumulate")


@pytest.mark.parametrize("D", [192, 384])
@pytest.mark.parametrize("M,save,two", [(1000, True, True), (70001, True, True), (4097, False, True), (333, True, False)])
def test_proj_ffn_ln_block_kernel_matches_the_separate_kernels(M, save, two, D):
    """out-proj + residual + norm1 + FFN + norm2 (+ next norm1) in one launch vs out-proj GEMM, LayerNorm and the fused
    FFN + LayerNorm-tail kernel: y bit for bit, norm1 up to the summation order of its statistics, everything after it bit
    for bit given the same x1 (ragged last panel, both the 3-stage no-H and the 2-stage H instance)."""
    from chadavit_amd import ops
    dev = _dev()
    FF = 2048
    blk, npb = 64 * D, D // 64   # elements per stream block, blocks per [D x D] projection matrix
    a = _rand((M, D), 51, 1.0).bfloat16().to(dev)
    x = _rand((M, D), 52, 1.0).bfloat16().to(dev)
    wo = (_rand((D, D), 53, 1.0) / math.sqrt(D)).bfloat16().to(dev)
    w1 = (_rand((FF, D), 54, 1.0) / math.sqrt(D)).bfloat16().to(dev)
    w2 = (_rand((D, FF), 55, 1.0) / math.sqrt(FF)).bfloat16().to(dev)
    bo, b1, b2 = _rand((D,), 56, 0.1).to(dev), _rand((FF,), 57, 0.1).to(dev), _rand((D,), 58, 0.1).to(dev)
    lns = [((1 + _rand((D,), 60 + i, 0.2)).to(dev), _rand((D,), 70 + i, 0.2).to(dev), 1e-5) for i in range(3)]
    # reference: the three launches it replaces
    y_r = ops.gemm_nt(a, wo, bias=bo, epilogue=ops.EPI_RESID, aux=x)
    m1_r, r1_r = torch.empty(M, device=dev), torch.empty(M, device=dev)
    x1_r = ops.layernorm_fwd(y_r, lns[0][0], lns[0][1], lns[0][2], mean=m1_r, rstd=r1_r)
    pk = ops.ffn_pack(w1, w2)
    z_r = torch.empty((M, D), device=dev, dtype=torch.bfloat16) if save else None
    h_r = torch.empty((M, FF), device=dev, dtype=torch.bfloat16) if save else None
    sa_r = (torch.empty(M, device=dev), torch.empty(M, device=dev))
    sb_r = (torch.empty(M, device=dev), torch.empty(M, device=dev))
    x2_r, hn_r = ops.ffn_ln_fwd(x1_r, pk, b1, b2, lns[1], resid=x1_r, z=z_r, h=h_r, ln_b=lns[2] if two else None, stats_a=sa_r,
                                stats_b=sb_r if two else None)
    # the [Wo | FFN] stream, packed from a bf16 "slab" holding the three matrices
    slab = torch.cat([w1.reshape(-1), w2.reshape(-1), wo.reshape(-1)])
    n = ops.ffn_proj_packed_bytes(D, FF) // 2
    pkp = torch.empty(n, device=dev, dtype=torch.bfloat16)
    desc = torch.tensor([0, w1.numel(), w1.numel() + w2.numel(), -1, 0], device=dev, dtype=torch.int64)
    ops.ffn_pack_proj_batched(slab, pkp, desc, 1, D, FF)
    assert torch.equal(pkp[npb * blk:pkp.numel() - 3 * npb * blk], pk)
    y = torch.empty((M, D), device=dev, dtype=torch.bfloat16) if save else None
    z = torch.empty((M, D), device=dev, dtype=torch.bfloat16) if save else None
    h = torch.empty((M, FF), device=dev, dtype=torch.bfloat16) if save else None
    s1 = (torch.empty(M, device=dev), torch.empty(M, device=dev))
    sa = (torch.empty(M, device=dev), torch.empty(M, device=dev))
    sb = (torch.empty(M, device=dev), torch.empty(M, device=dev))
    x1, x2, hn = ops.proj_ffn_ln_fwd(a, x, pkp, bo, lns[0], b1, b2, lns[1], y=y, stats1=s1, z=z, h=h, ln_b=lns[2] if two else None,
                                     stats_a=sa, stats_b=sb if two else None)
    if save:
        assert torch.equal(y, y_r), float((y.float() - y_r.float()).abs().max())  # the projection + residual: bit for bit
    # norm1: equal up to the fp32 summation order of the row statistics (<= 1 bf16 ulp on a few elements)
    assert torch.allclose(s1[0], m1_r, atol=1e-5) and torch.allclose(s1[1], r1_r, rtol=1e-5)
    d = (x1.float() - x1_r.float()).abs()
    assert d.max().item() <= 3.2e-2 and (d > 0).float().mean().item() < 0.02
    # everything after norm1: bit for bit against the stand-alone fused FFN fed with THIS x1
    z_c = torch.empty((M, D), device=dev, dtype=torch.bfloat16) if save else None
    h_c = torch.empty((M, FF), device=dev, dtype=torch.bfloat16) if save else None
    sa_c = (torch.empty(M, device=dev), torch.empty(M, device=dev))
    sb_c = (torch.empty(M, device=dev), torch.empty(M, device=dev))
    x2_c, hn_c = ops.ffn_ln_fwd(x1, pk, b1, b2, lns[1], resid=x1, z=z_c, h=h_c, ln_b=lns[2] if two else None, stats_a=sa_c,
                                stats_b=sb_c if two else None)
    if save:
        assert torch.equal(z, z_c) and torch.equal(h, h_c)
    assert torch.equal(x2, x2_c) and torch.equal(sa[0], sa_c[0]) and torch.equal(sa[1], sa_c[1])
    if two:
        assert torch.equal(hn, hn_c) and torch.equal(sb[0], sb_c[0]) and torch.equal(sb[1], sb_c[1])
    else:
        assert hn is None
    # and the whole chain stays within a bf16 ulp or two of the three-launch reference
    d2 = (x2.float() - x2_r.float()).abs()
    assert d2.max().item() <= 6.3e-2 and (d2 > 0).float().mean().item() < 0.03
    if two:
        # postlogue: the NEXT block's QKV projection of hn -- bit for bit against the GEMM on the hn this kernel produces,
        # with and without hn itself being written
        wq = (_rand((3 * D, D), 59, 1.0) / math.sqrt(D)).bfloat16().to(dev)
        bq = _rand((3 * D,), 61, 0.1).to(dev)
        slab2 = torch.cat([slab, wq.reshape(-1)])
        pkq = torch.empty(n, device=dev, dtype=torch.bfloat16)
        ops.ffn_pack_proj_batched(slab2, pkq, torch.tensor([0, w1.numel(), w1.numel() + w2.numel(), slab.numel(), 0], device=dev,
                                                           dtype=torch.int64), 1, D, FF)
        assert torch.equal(pkq[:n - 3 * npb * blk], pkp[:n - 3 * npb * blk])
        for want_hn in (True, False):
            x1q, x2q, hnq, qkv = ops.proj_ffn_ln_fwd(a, x, pkq, bo, lns[0], b1, b2, lns[1], ln_b=lns[2], qkv_bias=bq, want_hn=want_hn,
                                                     h=torch.empty_like(h) if save else None)
            assert torch.equal(x2q, x2) and torch.equal(x1q, x1) and (hnq is None) == (not want_hn)
            if want_hn:
                assert torch.equal(hnq, hn)
            assert torch.equal(qkv, ops.gemm_nt(hn, wq, bias=bq)), float((qkv.float() - ops.gemm_nt(hn, wq, bias=bq).float()).abs().max())


@pytest.mark.parametrize("M,D", [(1000, 192), (70001, 192), (333, 192), (128, 192), (1000, 384), (33001, 384), (129, 384)])
def test_ffn_bwd_dx_from_relu_bits(M, D):
    """Backward dX pass of the FFN without a hidden-wide tensor: the forward records 1 bit per hidden activation (relu_bits), the
    backward kernel computes dx1 = dz + ((dz W2) * [H > 0]) W1 in one launch.  Checked against fp32 torch on the same bf16 inputs
    (the mask taken from the H the forward also wrote here), and bit for bit on dpre against the mask itself; recording the bits
    must not change any forward output."""
    from chadavit_amd import ops
    dev = _dev()
    FF = 2048
    x = _rand((M, D), 81, 1.0).bfloat16().to(dev)
    dz = _rand((M, D), 82, 1.0).bfloat16().to(dev)
    w1 = (_rand((FF, D), 83, 1.0) / math.sqrt(D)).bfloat16().to(dev)
    w2 = (_rand((D, FF), 84, 1.0) / math.sqrt(FF)).bfloat16().to(dev)
    b1, b2 = _rand((FF,), 85, 0.3).to(dev), _rand((D,), 86, 0.1).to(dev)
    pk = ops.ffn_pack(w1, w2)
    h0 = torch.empty((M, FF), device=dev, dtype=torch.bfloat16)
    out0 = ops.ffn_fwd(x, pk, b1, b2, resid=x, h=h0)
    bits = ops.relu_bits_buffer(M, FF, dev)
    bits.fill_(0xA5)
    for with_h in (True, False):   # the two instances that record bits: beside H (2 LDS stages) and instead of it (3 stages)
        h = torch.empty_like(h0) if with_h else None
        out = ops.ffn_fwd(x, pk, b1, b2, resid=x, h=h, relu_bits=bits)
        assert torch.equal(out, out0) and (h is None or torch.equal(h, h0))
        pkb = ops.ffn_pack(w2.t().contiguous(), w1.t().contiguous())   # W2^T in the W1 slot, W1^T in the W2 slot
        dpre = torch.empty((M, FF), device=dev, dtype=torch.bfloat16)
        dx1 = ops.ffn_bwd_dx(dz, pkb, bits, dpre=dpre)
        dx1_b = ops.ffn_bwd_dx(dz, pkb, bits)              # the instance without the dpre output
        mask = (h0.float() > 0)
        dh_ref = (dz.float() @ w2.float()) * mask
        _close(dpre, dh_ref, 1e-2, 1e-2, "dpre")
        assert torch.equal(dpre.float() != 0, (dh_ref.bfloat16().float() != 0) & mask) or \
            float(((dpre.float() != 0) != ((dh_ref != 0) & mask)).float().mean()) < 1e-5   # the mask itself: exact (up to dH == 0)
        dx_ref = dz.float() + dpre.float() @ w1.float()
        _close(dx1, dx_ref, 1e-2, 2e-2, "dx1")
        assert torch.equal(dx1, dx1_b)
        # and against the two-GEMM path it replaces
        dhid = ops.gemm_nt(dz, w2.t().contiguous(), epilogue=ops.EPI_RELUMASK, aux=h0)
        dx_two = ops.gemm_nt(dhid, w1.t().contiguous(), epilogue=ops.EPI_RESID, aux=dz)
        assert float((dpre.float() - dhid.float()).abs().max()) <= 2e-2 * float(dhid.float().abs().max())
        assert float((dx1.float() - dx_two.float()).abs().max()) <= 3e-2 * float(dx_two.float().abs().max())


@pytest.mark.parametrize("T,D", [(1000, 192), (70001, 192), (333, 384), (70, 768), (3, 192)])
@pytest.mark.parametrize("acc", [False, True])
def test_layernorm_bwd_pair_equals_two_calls(T, D, acc):
    """The block-boundary sweep dz = LN_b'(LN_a'(dy; x) + dres; z) against the two layernorm_bwd launches it replaces: the same
    arithmetic with dx rounded to bf16 in between; the compiler contracts the multiply-adds of the two kernels differently, so dz
    agrees to one bf16 ulp and the four column sums to fp32 rounding."""
    from chadavit_amd import ops
    dev = _dev()
    z = _rand((T, D), 31, 2.0).bfloat16().to(dev)
    gb, bb = (1 + _rand((D,), 32, 0.2)).to(dev), _rand((D,), 33, 0.2).to(dev)
    ga = (1 + _rand((D,), 34, 0.2)).to(dev)
    st = torch.empty((4, T), device=dev)
    x = ops.layernorm_fwd(z, gb, bb, 1e-5, mean=st[2], rstd=st[3])          # x = LN_b(z): the next block's input
    ops.layernorm_fwd(x, ga, torch.zeros_like(ga), 1e-5, mean=st[0], rstd=st[1])
    dy = _rand((T, D), 35, 1.0).bfloat16().to(dev)
    dres = _rand((T, D), 36, 1.0).bfloat16().to(dev)
    ws = ops.layernorm_bwd_workspace(D, dev)
    init = 0.25 if acc else float("nan")
    ref = [torch.full((D,), init, device=dev) for _ in range(4)]
    dx = ops.layernorm_bwd(dy, x, st[0], st[1], ga, ref[0], ref[1], ws, dres=dres, accumulate=acc)
    dz_ref = ops.layernorm_bwd(dx, z, st[2], st[3], gb, ref[2], ref[3], ws, accumulate=acc)
    got = [torch.full((D,), init, device=dev) for _ in range(4)]
    dz = ops.layernorm_bwd_pair(dy, x, st[0], st[1], ga, dres, z, st[2], st[3], gb, got[0], got[1], got[2], got[3], ws,
                                accumulate_a=acc, accumulate_b=acc)
    # the instance that rebuilds x from z instead of reading it: x = bf16(LN_b(z)) is reproduced bit for bit (ln_apply, explicit FMA);
    # the two template instances are separate compilations whose implicit multiply-add contractions may differ (they did once packed
    # f32 ops were switched off, round 3), so dz agrees to one bf16 ulp and the column sums to fp32 rounding, as for the two-call path
    got2 = [torch.full((D,), init, device=dev) for _ in range(4)]
    dz2 = ops.layernorm_bwd_pair(dy, x, st[0], st[1], ga, dres, z, st[2], st[3], gb, got2[0], got2[1], got2[2], got2[3], ws,
                                 accumulate_a=acc, accumulate_b=acc, beta_b=bb)
    torch.cuda.synchronize()
    e2 = (dz2.float() - dz.float()).abs()
    assert bool((e2 <= 2 ** -7 * dz.float().abs() + 2e-3 * float(dz.float().abs().max())).all()) and float((e2 > 0).float().mean()) < 0.01, float(e2.max())
    for a, b in zip(got2, got):
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1.0), float((a - b).abs().max())
    err = (dz.float() - dz_ref.float()).abs()
    # one bf16 ulp of the element, or of the intermediate dx spread over its row (small |dz| next to large ones)
    assert bool((err <= 2 ** -7 * dz_ref.float().abs() + 2e-3 * float(dz_ref.float().abs().max())).all()), float(err.max())
    assert float((err > 0).float().mean()) < 0.02   # and almost everywhere identical
    for a, b in zip(got, ref):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1.0), float((a - b).abs().max())


@pytest.mark.parametrize("T,D", [(1000, 192), (333, 384), (70, 768), (5, 1024)])
def test_layernorm_fwd_bwd(T, D):
    from chadavit_amd import ops
    dev = _dev()
    x = _rand((T, D), 7, 2.0).bfloat16().to(dev)
    gamma = (1 + _rand((D,), 8, 0.2)).to(dev)
    beta = _rand((D,), 9, 0.2).to(dev)
    dy = _rand((T, D), 10, 1.0).bfloat16().to(dev)
    dres = _rand((T, D), 11, 1.0).bfloat16().to(dev)
    mean = torch.empty(T, device=dev)
    rstd = torch.empty(T, device=dev)
    y = ops.layernorm_fwd(x, gamma, beta, 1e-5, mean=mean, rstd=rstd)
    xf = x.float().requires_grad_(True)
    gf = gamma.clone().requires_grad_(True)
    bf = beta.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xf, (D,), gf, bf, 1e-5)
    _close(y, yr, 1e-2, 1e-2, "ln fwd")
    _close(mean, xf.mean(1), 1e-5, 1e-5, "mean")
    yr.backward(dy.float())
    dg = torch.zeros(D, device=dev)
    db = torch.zeros(D, device=dev)
    ws = ops.layernorm_bwd_workspace(D, dev)
    dx = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, ws, dres=dres)
    _close(dx, xf.grad + dres.float(), 1e-2, 2e-2, "ln dx")
    _close(dg, gf.grad, 1e-3, 5e-3 * math.sqrt(T), "ln dgamma")
    _close(db, bf.grad, 1e-3, 5e-3 * math.sqrt(T), "ln dbeta")
    dx2 = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, ws, accumulate=True)
    _close(dx2, xf.grad, 1e-2, 2e-2, "ln dx no-res")
    _close(dg, 2 * gf.grad, 1e-3, 1e-2 * math.sqrt(T), "ln dgamma acc")


def _attn_ref(qkv, cu, H):
    T, D3 = qkv.shape
    D = D3 // 3
    dh = D // H
    q, k, v = qkv.split(D, dim=1)
    outs = []
    for i in range(len(cu) - 1):
        s, e = cu[i], cu[i + 1]
        qi = q[s:e].reshape(e - s, H, dh).transpose(0, 1)
        ki = k[s:e].reshape(e - s, H, dh).transpose(0, 1)
        vi = v[s:e].reshape(e - s, H, dh).transpose(0, 1)
        att = torch.softmax(qi @ ki.transpose(1, 2) / math.sqrt(dh), dim=-1)
        outs.append((att @ vi).transpose(0, 1).reshape(e - s, D))
    return torch.cat(outs, 0)


@pytest.mark.parametrize("nch,p,D,H", [([3, 1, 10, 5], 196, 192, 2), ([1, 2], 36, 192, 2), ([2, 1, 1], 36, 384, 2),
                                       ([1], 4, 64, 2), ([3, 2], 36, 128, 2), ([1, 3], 196, 384, 2), ([2, 1], 36, 768, 2),
                                       ([1, 2], 196, 768, 2), ([2, 3], 36, 192, 12), ([1, 2], 196, 192, 12),
                                       # the longest sequence the model can see (10 channels x 196 patches + CLS = 1961 tokens: cfg5's
                                       # "max-token stress"), at all three head widths
                                       ([10, 1], 196, 192, 2), ([10], 196, 384, 2), ([1, 10], 196, 768, 2),
                                       # embed_dim 256 / 512 with the factory's two heads (dh 128 / 256: the register-staged kernels)
                                       ([3, 2], 36, 256, 2), ([1, 3, 10], 196, 256, 2), ([2, 1], 36, 512, 2), ([1, 10], 196, 512, 2)])
def test_attention_fwd_bwd(nch, p, D, H):
    from chadavit_amd import ops
    from chadavit_amd.ragged import RaggedBatch
    dev = _dev()
    rb = RaggedBatch(nch, p, dev)
    T = rb.T
    qkv = _rand((T, 3 * D), 12, 1.0).bfloat16().to(dev)
    # one spiky query/key pair to exercise the running-max rescale across KV tiles
    qkv[min(T - 1, 70), :D] *= 6.0
    dout = _rand((T, D), 13, 1.0).bfloat16().to(dev)
    out, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    qf = qkv.float().requires_grad_(True)
    ref = _attn_ref(qf, rb.host_cu_seqlens, H)
    _close(out, ref, 2e-2, 2e-2, "attn fwd")
    ref.backward(dout.float())
    dqkv = ops.attn_bwd(qkv, out, dout, lse, rb.cu_seqlens, rb.work, H)
    g = qf.grad
    scale = g.abs().max().item()
    _close(dqkv[:, 2 * D:], g[:, 2 * D:], 3e-2, 3e-2 * scale, "dV")
    _close(dqkv[:, D:2 * D], g[:, D:2 * D], 3e-2, 3e-2 * scale, "dK")
    _close(dqkv[:, :D], g[:, :D], 3e-2, 3e-2 * scale, "dQ")


def test_attention_dh384_round5_kernels_are_bit_identical_to_the_ones_they_replaced(tmp_path):
    """dh 384 dispatches attn_fwd_pair_kernel (the two waves of every SIMD in complementary phases, K ring of two / V ring of three, two barriers per key
    tile): same arithmetic in the same order as attn_fwd_dma_kernel<384> -- outputs and LSE must agree BIT FOR BIT with that kernel
    (the A/B side build with CHADAVIT_ATTN_FWD_PAIR=0, read once per process: child process) on ragged batches: 1961-token sequences, a last tile of one key (len 33 = 32 + 1),
    a single tile, sequences shorter than a tile, waves without query rows."""
    import subprocess, sys, os
    from chadavit_amd import ops
    from chadavit_amd.ragged import RaggedBatch
    dev = _dev()
    cases = [([10, 1, 3], 196), ([2, 10, 1], 36), ([1], 4), ([5, 7], 196), ([3, 7, 1, 10, 2, 5], 7), ([1, 2, 3, 4, 5, 6, 7, 8, 9, 10], 33), ([10, 9, 1], 121),
             ([4, 8, 2], 64)]   # (lengths 8 ... 1961: every parity of the tile count, remainders 1 ... 32; scratch/r5/pair_sweep.py runs 18 more sets)
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from chadavit_amd import ops
from chadavit_amd.ragged import RaggedBatch
dev = torch.device("cuda:0"); outs = []
for i, (nch, p) in enumerate(%r):
    rb = RaggedBatch(nch, p, dev)
    qkv = torch.randn((rb.T, 3 * 768), generator=torch.Generator(device="cpu").manual_seed(40 + i)).bfloat16().to(dev)
    o, l = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
    do = torch.randn((rb.T, 768), generator=torch.Generator(device="cpu").manual_seed(80 + i)).bfloat16().to(dev)
    delta = torch.empty((2, rb.T), device=dev)
    dqkv = ops.attn_bwd(qkv, o, do, l, rb.cu_seqlens, rb.work, 2, delta=delta)
    outs.append((o.cpu(), l.cpu(), dqkv.cpu(), delta.cpu()))
torch.save(outs, %r)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), cases, str(tmp_path / "old.pt"))
    # (the replaced kernels and their switches live in the A/B side build only -- chadavit_amd.build.build_ab, -DCHADA_AB_SWITCHES=1 -- not in the product)
    from chadavit_amd.build import build_ab
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CHADAVIT_ATTN_FWD_PAIR="0", CHADAVIT_ATTN_DQ_RM="0", CHADAVIT_HIP_LIB=build_ab(),
                                                              CHADAVIT_ALLOW_FOREIGN_LIB="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    old = torch.load(tmp_path / "old.pt")
    for i, (nch, p) in enumerate(cases):
        rb = RaggedBatch(nch, p, dev)
        qkv = torch.randn((rb.T, 3 * 768), generator=torch.Generator(device="cpu").manual_seed(40 + i)).bfloat16().to(dev)
        o, l = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
        assert torch.equal(o.cpu().view(torch.int16), old[i][0].view(torch.int16)) and torch.equal(l.cpu(), old[i][1]), (nch, p)
        _close(o, _attn_ref(qkv.float(), rb.host_cu_seqlens, 2), 2e-2, 2e-2, f"paired forward {nch}")
        # dQ at dh 384: the row-major-stage kernel as eight waves x 16 rows against the fragment-major one it replaced (dK / dV: same kernel in both)
        do = torch.randn((rb.T, 768), generator=torch.Generator(device="cpu").manual_seed(80 + i)).bfloat16().to(dev)
        delta = torch.empty((2, rb.T), device=dev)
        dqkv = ops.attn_bwd(qkv, o, do, l, rb.cu_seqlens, rb.work, 2, delta=delta)
        assert torch.equal(dqkv.cpu().view(torch.int16), old[i][2].view(torch.int16)) and torch.equal(delta.cpu(), old[i][3]), (nch, p)


def test_attention_fwd_row_major_stages_and_paired_schedule_are_bit_identical_to_the_fragment_major_kernel():
    """Round 6.  (a) The 32x32x16 forward stages its K / V tiles as row-major images since this round (CHADA_M32_RM = 1: whole-row pieces per LDS-DMA
    instruction, chunks XOR-swizzled on the source side); a side build with the fragment-major records of rounds 3-5 (-DCHADA_M32_RM=0) must give the
    same outputs and LSE BIT FOR BIT on ragged batches at dh 96 / 192 incl. the lean softmax's re-run (a spiked score), single-tile and edge-length
    sequences.  (b) The paired (ping-pong) forward -- two heads per 512-thread block, the halves one segment apart; built, measured, not adopted: a
    side-build kernel, variant 6 of chadavit_attn_fwd_m32 -- must reproduce the unpaired kernel bit for bit, re-run included; (c) likewise the paired dK/dV kernel at dh 96."""
    import subprocess, sys, os
    from chadavit_amd.build import build, build_ab
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fm = build(verbose=False, side="m32fm", extra_flags=["-DCHADA_M32_RM=0"], units=("attention_m32.hip",))
    dumps = []
    for lib in (None, fm):
        env = dict(os.environ) if lib is None else dict(os.environ, CHADAVIT_HIP_LIB=lib, CHADAVIT_ALLOW_FOREIGN_LIB="1")
        r = subprocess.run([sys.executable, os.path.join(root, "scratch", "r6", "fwd_dump.py")], env=env, capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        dumps.append([ln for ln in r.stdout.splitlines() if ln and ln[0].isdigit()])
    assert len(dumps[0]) == 22 and dumps[0] == dumps[1], [(a, b) for a, b in zip(*dumps) if a != b][:4]
    assert all(ln.endswith("True") for ln in dumps[0])   # finite everywhere
    r = subprocess.run([sys.executable, os.path.join(root, "scratch", "r6", "p32_time.py"), "check"], capture_output=True, text=True, timeout=900, cwd=root,
                       env=dict(os.environ, CHADAVIT_HIP_LIB=build_ab(), CHADAVIT_ALLOW_FOREIGN_LIB="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if "identical=" in ln]
    assert len(lines) == 18 and all("identical=True" in ln for ln in lines), [ln for ln in lines if "identical=True" not in ln]
    # (c) the paired dK/dV at dh 96 (side-build kernel, CHADAVIT_ATTN_DKV_PAIR=1: built, register-bound, not adopted) reproduces the product's dqkv and delta
    dumps = []
    for env in (dict(os.environ), dict(os.environ, CHADAVIT_HIP_LIB=build_ab(), CHADAVIT_ALLOW_FOREIGN_LIB="1", CHADAVIT_ATTN_DKV_PAIR="1")):
        r = subprocess.run([sys.executable, os.path.join(root, "scratch", "r6", "bwd_dump.py")], env=env, capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        dumps.append([ln for ln in r.stdout.splitlines() if ln and ln[0].isdigit()])
    assert len(dumps[0]) == 12 and dumps[0] == dumps[1], [(a, b) for a, b in zip(*dumps) if a != b][:4]


def test_attention_fwd_row_major_stages():
    """A forward instance of the A/B side build (not in the product library), kept for its measurement (profiles/r05o_*): the 16x16x32 forward on
    ROW-MAJOR LDS stages (CHADAVIT_ATTN_FWD_RM=1: whole 128-byte lines per LDS-DMA instruction, chunks swizzled on the source side; the switch is read
    once per process, hence the child process), dh 96 / 192 / 384, against fp32 torch on ragged batches incl. the 1961-token sequence and a last
    tile of 13 keys, LSE included."""
    import subprocess, sys, os
    code = r"""
import ctypes, math, sys, torch
sys.path.insert(0, %r)
from chadavit_amd import ops
from chadavit_amd._lib import lib
from chadavit_amd.ragged import RaggedBatch
dev = torch.device("cuda:0")
def ref(qkv, cu, H):
    D = qkv.shape[1] // 3; dh = D // H
    q, k, v = qkv.float().split(D, dim=1)
    outs, lses = [], []
    for i in range(len(cu) - 1):
        s, e = cu[i], cu[i + 1]
        qi, ki, vi = (x[s:e].reshape(e - s, H, dh).transpose(0, 1) for x in (q, k, v))
        sc = qi @ ki.transpose(1, 2) / math.sqrt(dh)
        outs.append((torch.softmax(sc, -1) @ vi).transpose(0, 1).reshape(e - s, D)); lses.append(torch.logsumexp(sc, -1))
    return torch.cat(outs, 0), torch.cat(lses, 1)
for D, nch, p in ((192, [3, 1, 10, 5], 196), (192, [1, 3, 2], 36), (384, [10, 2, 3], 196), (768, [1, 10, 3], 196), (768, [2, 10], 36)):
    rb = RaggedBatch(nch, p, dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    qkv = torch.randn((rb.T, 3 * D), generator=g).bfloat16().to(dev)
    o_ref, l_ref = ref(qkv, rb.host_cu_seqlens, 2)
    o, l = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, 2)
    assert float((o.float() - o_ref).abs().max()) < 2e-2 and float((l - l_ref).abs().max()) < 2e-3, ("row-major", D, nch)
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from chadavit_amd.build import build_ab   # (a side-build instance: the product library has neither the kernel nor the switch)
    env = dict(os.environ, CHADAVIT_ATTN_FWD_RM="1", CHADAVIT_HIP_LIB=build_ab(), CHADAVIT_ALLOW_FOREIGN_LIB="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("D", [192, 384, 768])
def test_attention_lengths_around_the_tile_boundaries(D):
    """Sequences of 2 .. 257 tokens straddling every 16 / 32 / 64 / 128-row boundary: the last tile of the LDS-DMA kernels
    skips the 16-row blocks without a valid key / query and whole waves without a valid row (dh = 96, 192 and -- forward: eight
    waves x 16 query rows -- 384)."""
    from chadavit_amd import ops
    from chadavit_amd.ragged import RaggedBatch
    dev = _dev()
    H = 2
    rb = RaggedBatch([1, 14, 15, 16, 30, 31, 32, 62, 63, 64, 95, 96, 126, 127, 128, 191, 192, 256], 1, dev)
    T = rb.T
    qkv = _rand((T, 3 * D), 41, 1.0).bfloat16().to(dev)
    dout = _rand((T, D), 42, 1.0).bfloat16().to(dev)
    out, lse = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    qf = qkv.float().requires_grad_(True)
    ref = _attn_ref(qf, rb.host_cu_seqlens, H)
    _close(out, ref, 2e-2, 2e-2, "attn fwd")
    ref.backward(dout.float())
    dqkv = ops.attn_bwd(qkv, out, dout, lse, rb.cu_seqlens, rb.work, H)
    g = qf.grad
    scale = g.abs().max().item()
    _close(dqkv[:, 2 * D:], g[:, 2 * D:], 3e-2, 3e-2 * scale, "dV")
    _close(dqkv[:, D:2 * D], g[:, D:2 * D], 3e-2, 3e-2 * scale, "dK")
    _close(dqkv[:, :D], g[:, :D], 3e-2, 3e-2 * scale, "dQ")
    # nothing leaks between neighbouring sequences or beyond the last row
    out2, _ = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("D", [192, 384])
def test_attention_fwd_m32_variants_and_out_of_range_rerun(D):
    """csrc/attention_m32.hip (the forward chadavit_attn_fwd dispatches to at dh = 96 / 192): the lean softmax keeps the FIRST key
    tile's row maximum as the exponent reference of the whole row.  (a) its three variants against fp32 torch, LSE included;
    (b) a row whose later scores exceed that reference by more than 2^64 -- a query / key pair with a raw score of ~600 placed in
    a later tile, and in the LAST (masked) tile -- must take the block-wide re-run with the online recurrence and still match;
    (c) the default dispatch IS that kernel (bit-identical to variant 0), and the 16x16x32 kernels agree with it."""
    import ctypes
    from chadavit_amd import ops
    from chadavit_amd._lib import lib
    from chadavit_amd.ragged import RaggedBatch
    dev = _dev()
    H = 2
    dh = D // H

    def fwd(qkv, rb, variant):
        out = torch.empty((rb.T, D), device=dev, dtype=torch.bfloat16)
        lse = torch.empty((H, rb.T), device=dev, dtype=torch.float32)
        rc = lib().chadavit_attn_fwd_m32(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(lse.data_ptr()),
                                         ctypes.c_void_p(rb.cu_seqlens.data_ptr()), ctypes.c_void_p(rb.work.data_ptr()),
                                         ctypes.c_int(rb.n_work), ctypes.c_int(rb.T), ctypes.c_int(D), ctypes.c_int(H), ctypes.c_int(variant),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        return out, lse

    def ref_lse(qkv, cu):
        q, k, _ = qkv.float().split(D, dim=1)
        out = torch.empty((H, qkv.shape[0]), device=dev)
        for i in range(len(cu) - 1):
            a, b = cu[i], cu[i + 1]
            for h in range(H):
                out[h, a:b] = torch.logsumexp(q[a:b, h * dh:(h + 1) * dh] @ k[a:b, h * dh:(h + 1) * dh].T / math.sqrt(dh), dim=1)
        return out

    rb = RaggedBatch([3, 1, 3], 196, dev)   # 589, 197, 589 tokens
    for spike in (None, (300, 500), (10, 588), (786 + 5, 786 + 588)):   # (query row, key row): later tile / last masked tile / third image
        qkv = _rand((rb.T, 3 * D), 12, 1.0).bfloat16().to(dev)
        if spike is not None:
            qkv[spike[0], :dh] = 2.5
            qkv[spike[1], D:D + dh] = 2.5     # raw score 6.25 dh = 600 (dh 96) / 1200 (dh 192): far beyond tile 0's maximum + 2^64
        ref = _attn_ref(qkv.float(), rb.host_cu_seqlens, H)
        lref = ref_lse(qkv, rb.host_cu_seqlens)
        for variant in (0, 1, 2):
            out, lse = fwd(qkv, rb, variant)
            _close(out, ref, 2e-2, 2e-2, f"attn_fwd_m32 variant {variant} spike {spike}")
            assert torch.isfinite(lse).all()
            if variant != 2:   # (variant 2 rounds the scaled Q a second time: its LSE carries that rounding, |score| * 2^-9)
                assert float((lse - lref).abs().max()) < 2e-3, (variant, spike, float((lse - lref).abs().max()))
        out_d, lse_d = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
        out_0, lse_0 = fwd(qkv, rb, 0)
        assert torch.equal(out_d, out_0) and torch.equal(lse_d, lse_0)


def test_tokenizer_path():
    from chadavit_amd import ops
    from chadavit_amd.ragged import RaggedBatch
    dev = _dev()
    nch, S, P, D = [3, 1, 4], 96, 16, 192
    g = S // P
    p = g * g
    rb = RaggedBatch(nch, p, dev)
    x = _rand((sum(nch), 1, S, S), 14).to(dev)
    W = _rand((D, 1, P, P), 15, 1 / 16).to(dev)
    bias = _rand((D,), 16, 0.1).to(dev)
    pos = _rand((p, D), 17, 0.5).to(dev)
    chan = _rand((10, D), 18, 0.5).to(dev)
    cls = _rand((D,), 19, 0.5).to(dev)
    pos0 = _rand((D,), 20, 0.5).to(dev)
    patches = ops.im2col(x.view(-1, S, S), P)
    ref_patches = torch.nn.functional.unfold(x, P, stride=P).transpose(1, 2).reshape(-1, P * P)
    _close(patches, ref_patches.bfloat16(), 0, 0, "im2col")
    tokens = torch.zeros((rb.T, D), device=dev, dtype=torch.bfloat16)
    ops.tokenizer_gemm(patches, W.view(D, -1).bfloat16().contiguous(), bias, pos, chan, rb.chan_img, rb.chan_idx, tokens, p)
    # the same GEMM with the unfold folded into its operand staging (no patch buffer): bit-identical tokens
    tokens_f = torch.zeros((rb.T, D), device=dev, dtype=torch.bfloat16)
    ops.tokenizer_fused(x.view(-1, S, S).contiguous(), W.view(D, -1).bfloat16().contiguous(), bias, pos, chan, rb.chan_img, rb.chan_idx,
                        tokens_f, p)
    assert torch.equal(tokens_f, tokens)
    ops.write_cls(tokens, rb.cu_seqlens, cls, pos0)
    conv = torch.nn.functional.conv2d(x.bfloat16().float(), W.bfloat16().float(), bias, stride=P).flatten(2).transpose(1, 2)
    rows, off = [], 0
    for c in nch:
        rows.append((cls + pos0)[None])
        rows.append((conv[off:off + c] + pos[None] + chan[:c, None]).reshape(c * p, D))
        off += c
    _close(tokens, torch.cat(rows), 1e-2, 1e-2, "tokens")
    # backward reductions
    dtok = _rand((rb.T, D), 21).bfloat16().to(dev)
    dpatch, dpos, dchan, dcls = ops.tokenizer_bwd(dtok, rb.cu_seqlens, rb.chan_img, rb.chan_idx, p, 10)
    cu = rb.host_cu_seqlens
    keep = torch.ones(rb.T, dtype=torch.bool)
    keep[cu[:-1]] = False
    dpt = dtok[keep.to(dev)].float()
    _close(dpatch, dpt, 0, 0, "dpatch")
    _close(dpos, dpt.view(-1, p, D).sum(0), 1e-3, 1e-2, "dpos")
    ci = torch.tensor([k for c in nch for k in range(c)], device=dev)
    ref_dchan = torch.zeros(10, D, device=dev).index_add_(0, ci, dpt.view(-1, p, D).sum(1))
    _close(dchan, ref_dchan, 1e-3, 2e-2, "dchan")
    _close(dcls, dtok[torch.tensor(cu[:-1], device=dev)].float().sum(0), 1e-3, 1e-2, "dcls")
    # gather / scatter
    gth = ops.gather_rows(dtok, rb.cls_rows)
    _close(gth, dtok[torch.tensor(cu[:-1], device=dev)], 0, 0, "gather")
    sc = ops.scatter_rows_zero(gth, rb.cls_rows, rb.T)
    ref_sc = torch.zeros_like(dtok)
    ref_sc[torch.tensor(cu[:-1], device=dev)] = gth
    _close(sc, ref_sc, 0, 0, "scatter")


def test_head_ops_and_loss():
    from chadavit_amd import ops
    dev = _dev()
    M, K, P = 10, 256, 4096
    x = _rand((M, K), 22).to(dev)
    dy = _rand((M, K), 23).to(dev)
    y, inv = ops.l2norm_fwd(x)
    xf = x.clone().requires_grad_(True)
    yr = torch.nn.functional.normalize(xf, dim=-1)
    _close(y, yr, 1e-2, 1e-3, "l2norm")
    yr.backward(dy)
    dx = ops.l2norm_bwd(dy, x, inv)
    _close(dx, xf.grad, 1e-2, 1e-3, "l2norm bwd")
    v = _rand((P, K), 24, 0.05).to(dev)
    g = torch.ones(P, 1, device=dev)
    w, wt, winv = ops.weightnorm_fwd(v, g.view(-1))
    vf = v.clone().requires_grad_(True)
    wr = g * vf / vf.norm(dim=1, keepdim=True)
    _close(w, wr, 1e-2, 1e-4, "weightnorm")
    _close(wt, wr.t(), 1e-2, 1e-4, "weightnorm T")
    dw = _rand((P, K), 25).to(dev)
    wr.backward(dw)
    dv = torch.zeros_like(v)
    ops.weightnorm_bwd(dw, v, g.view(-1), winv, dv)
    _close(dv, vf.grad, 1e-3, 1e-3 * vf.grad.abs().max().item(), "weightnorm bwd")
    # loss
    B = 5
    s = _rand((2 * B, P), 26).to(dev)
    t = _rand((2 * B, P), 27).to(dev)
    c = _rand((1, P), 28, 0.05).to(dev)
    loss_rows, ds, colsum = ops.dino_loss(s, t, c.view(-1), 0.1, 0.055)
    sf = s.clone().requires_grad_(True)
    so = (sf / 0.1).chunk(2)
    q = torch.softmax((t - c) / 0.055, dim=-1).chunk(2)
    loss = 0.5 * (torch.sum(-q[0] * torch.log_softmax(so[1], -1), -1).mean() + torch.sum(-q[1] * torch.log_softmax(so[0], -1), -1).mean())
    loss.backward()
    assert abs(loss_rows.mean().item() - loss.item()) < 1e-4 * abs(loss.item())
    _close(ds, sf.grad, 2e-2, 1e-2 * sf.grad.abs().max().item(), "dstudent")
    _close(colsum, t.sum(0), 1e-5, 1e-4, "colsum")
    cc = c.view(-1).clone()
    ops.center_ema(cc, colsum, 1.0 / (2 * B), 0.9)
    _close(cc, (c * 0.9 + t.sum(0, keepdim=True) / (2 * B) * 0.1).view(-1), 1e-5, 1e-6, "center")


def test_flat_param_kernels():
    from chadavit_amd import ops
    from oracle import chada_ref as R
    dev = _dev()
    n = 100003
    p = _rand((n,), 29).to(dev)
    g = _rand((n,), 30, 0.1).to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    for step in (1, 2, 3):
        ops.adamw_step(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step)
        pr, mr, vr = R.adamw_step(pr, g, mr, vr, step, 1e-3, 1e-2)
    _close(p, pr, 1e-5, 1e-6, "adamw")
    t = _rand((n,), 31).to(dev)
    tr = 0.99 * t + 0.01 * p
    ops.ema_update(t, p, 0.99)
    _close(t, tr, 1e-6, 1e-7, "ema")
    d = torch.empty(n, device=dev, dtype=torch.bfloat16)
    ops.cast_bf16(p, d)
    _close(d, p.bfloat16(), 0, 0, "cast")
    src = _rand((70, 300), 32).to(dev)
    a = torch.empty((70, 300), device=dev, dtype=torch.bfloat16)
    b = torch.empty((300, 70), device=dev, dtype=torch.bfloat16)
    ops.cast_transpose_bf16(src, a, b)
    _close(a, src.bfloat16(), 0, 0, "cast_t a")
    _close(b, src.bfloat16().t(), 0, 0, "cast_t b")
    grads = _rand((1000,), 33).to(dev)
    offs = torch.tensor([0, 100, 600], device=dev)
    sizes = torch.tensor([100, 500, 400], device=dev)
    ref = grads.clone()
    for o, s in ((0, 100), (100, 500), (600, 400)):
        nrm = ref[o:o + s].norm()
        coef = 3.0 / (nrm + 1e-6)
        if coef < 1:
            ref[o:o + s] *= coef
    ops.clip_tensors(grads, offs, sizes, 3.0)
    _close(grads, ref, 1e-5, 1e-6, "clip")
    x = _rand((37, 5000), 34).to(dev)
    _close(ops.sum_rows_f32(x, 0.5), x.sum(0) * 0.5, 1e-5, 1e-5, "sum_rows")
    # the centre's shapes (2B x P: 2-D decomposition, eight loads in flight per row group) and the scalar path (cols % 4 != 0)
    for rows, cols in ((1024, 4096), (2048, 4096), (130, 65536), (300, 1027), (3, 64)):
        x = _rand((rows, cols), 35).to(dev)
        _close(ops.sum_rows_f32(x), x.double().sum(0).float(), 1e-5, 2e-4, f"sum_rows {rows}x{cols}")
        assert torch.equal(ops.sum_rows_f32(x), ops.sum_rows_f32(x)), "sum_rows must be deterministic"


@pytest.mark.parametrize("kind,kw", [("adam", dict(lr=1e-3, weight_decay=1e-2)), ("adam", dict(lr=3e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=0.0)),
                                      ("sgd", dict(lr=0.05, weight_decay=1e-3)), ("sgd", dict(lr=0.05, momentum=0.9, weight_decay=1e-3)),
                                      ("sgd", dict(lr=0.05, momentum=0.9, dampening=0.1)), ("sgd", dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-2))])
def test_fused_adam_and_sgd_match_torch_optim(kind, kw):
    """The reference's other optimiser choices (base.py:67-72: torch.optim.Adam / torch.optim.SGD) as one launch per slab run, through
    the module's own `configure_optimizers` plumbing (FlatParams slabs, parameters that receive no gradient are skipped): four steps
    against torch.optim on a copy of the same parameters and gradients."""
    from chadavit_amd.optim import FusedAdam, FusedSGD
    dev = _dev()
    m = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.LayerNorm(53), torch.nn.Linear(53, 11)).to(dev)
    ref = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    params = list(m.parameters())
    fused = (FusedAdam if kind == "adam" else FusedSGD)([{"params": params[:2]}, {"params": params[2:], "weight_decay": 0.0}], **kw)
    topt = (torch.optim.Adam if kind == "adam" else torch.optim.SGD)([{"params": ref[:2]}, {"params": ref[2:], "weight_decay": 0.0}], **kw)
    for step in range(4):
        for i, (p, r) in enumerate(zip(params, ref)):
            g = _rand(tuple(p.shape), 400 + 10 * step + i, 0.3).to(dev)
            skip = (i == 3 and step < 2)   # a parameter without a gradient for the first steps (frozen prototypes): its counters lag
            p.grad = None if skip else g.clone()
            r.grad = None if skip else g.clone()
        fused.step()
        topt.step()
    for p, r in zip(params, ref):
        _close(p.detach(), r.detach(), 2e-6, 2e-6, f"{kind} {kw}")


def test_layernorm_fwd2_is_two_layernorms():
    from chadavit_amd import ops
    dev = _dev()
    T, D = 777, 192
    x = _rand((T, D), 50, 2.0).bfloat16().to(dev)
    ga, ba = (1 + _rand((D,), 51, 0.2)).to(dev), _rand((D,), 52, 0.2).to(dev)
    gb, bb = (1 + _rand((D,), 53, 0.2)).to(dev), _rand((D,), 54, 0.2).to(dev)
    st1 = (torch.empty(T, device=dev), torch.empty(T, device=dev))
    st2 = (torch.empty(T, device=dev), torch.empty(T, device=dev))
    y1, y2 = ops.layernorm_fwd2(x, ga, ba, gb, bb, 1e-5, 1e-6, stats1=st1, stats2=st2)
    m1, r1, m2, r2 = (torch.empty(T, device=dev) for _ in range(4))
    z1 = ops.layernorm_fwd(x, ga, ba, 1e-5, mean=m1, rstd=r1)
    z2 = ops.layernorm_fwd(z1, gb, bb, 1e-6, mean=m2, rstd=r2)
    assert torch.equal(y1, z1) and torch.equal(y2, z2)
    assert torch.equal(st1[0], m1) and torch.equal(st1[1], r1) and torch.equal(st2[0], m2) and torch.equal(st2[1], r2)


def test_lars_matches_oracle_and_golden():
    import os
    import numpy as np
    from chadavit_amd import ops
    from oracle import chada_ref as R
    from oracle import procedural as P
    dev = _dev()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lars.npz"))
    shapes = {"w0": (64, 48), "b0": (64,), "w1": (16, 64, 3), "g1": (16,), "z": (8, 8)}
    combos = {"plain": dict(), "excl": dict(exclude_bias_n_norm=True), "clip_nest": dict(clip_lr=True, nesterov=True),
              "wd0": dict(weight_decay=0.0)}
    for cname, kw in combos.items():
        args = dict(lr=0.3, momentum=0.9, dampening=0.0, weight_decay=1e-2, eta=1e-3, eps=1e-8, clip_lr=False, nesterov=False,
                    exclude_bias_n_norm=False)
        args.update(kw)
        offs, sizes, vals = [], [], []
        off = 0
        for n, s in shapes.items():
            p = P.tensor(s, "lars." + n, 0.5, seed=61)
            if n == "z":
                p = torch.zeros_like(p)
            offs.append(off); sizes.append(p.numel()); vals.append(p.flatten())
            off += (p.numel() + 63) // 64 * 64
        flat = torch.zeros(off)
        for o, v in zip(offs, vals):
            flat[o:o + v.numel()] = v
        flat = flat.to(dev)
        buf = torch.zeros_like(flat)
        offs_t = torch.tensor(offs, dtype=torch.int64, device=dev)
        sizes_t = torch.tensor(sizes, dtype=torch.int64, device=dev)
        for step in range(2):
            gr = torch.zeros(off)
            for (n, s), o in zip(shapes.items(), offs):
                gg = P.tensor(s, f"lars.g{step}." + n, 0.2, seed=62).flatten()
                gr[o:o + gg.numel()] = gg
            flags = torch.tensor([(1 if (len(s) != 1 or not args["exclude_bias_n_norm"]) else 0) | (2 if step else 0) for s in shapes.values()],
                                 dtype=torch.int32, device=dev)
            ops.lars_step(flat, gr.to(dev), buf, offs_t, sizes_t, flags, args["lr"], args["momentum"], args["dampening"],
                          args["weight_decay"], args["eta"], args["eps"], args["clip_lr"], args["nesterov"])
        out = flat.cpu()
        for (n, s), o in zip(shapes.items(), offs):
            np.testing.assert_allclose(out[o:o + int(np.prod(s))].numpy().reshape(s), g[f"{cname}::{n}"], rtol=2e-5, atol=1e-6,
                                       err_msg=f"{cname} {n}")


@pytest.mark.parametrize("M,rpw,write_h,D", [(77, 32, True, 192), (1000, 32, False, 192), (4099, 32, True, 192),
                                             (77, 32, True, 384), (1000, 32, False, 384), (4099, 32, True, 384)])
def test_fused_ffn_matches_fp64_and_two_gemm_path(M, rpw, write_h, D):
    """ops.ffn_fwd (one kernel, hidden activation on chip) vs fp64 math and vs linear1 -> relu -> linear2 + residual as two
    GEMM launches (torch.nn.TransformerEncoderLayer feed-forward, chada_vit.py:256-264).  D = 384 (Small) is the second build of
    the kernel: 8 waves x 16 rows per block."""
    from chadavit_amd import ops
    dev = _dev()
    FF = 2048
    gen = torch.Generator(device="cpu").manual_seed(M)
    bf = torch.bfloat16
    x = torch.randn((M, D), generator=gen).to(dev).to(bf)
    w1 = (torch.randn((FF, D), generator=gen) / D ** 0.5).to(dev).to(bf)
    w2 = (torch.randn((D, FF), generator=gen) / FF ** 0.5).to(dev).to(bf)
    b1 = (torch.randn(FF, generator=gen) * 0.1).to(dev)
    b2 = (torch.randn(D, generator=gen) * 0.1).to(dev)
    res = torch.randn((M, D), generator=gen).to(dev).to(bf)
    pk = ops.ffn_pack(w1, w2)
    assert pk.numel() * 2 == ops.ffn_packed_bytes(D, FF)
    h = torch.full((M, FF), float("nan"), device=dev, dtype=bf) if write_h else None
    out = ops.ffn_fwd(x, pk, b1, b2, resid=res, h=h, rows_per_wave=rpw)
    h64 = torch.relu(x.double() @ w1.double().T + b1.double())
    o64 = h64.to(bf).double() @ w2.double().T + b2.double() + res.double()
    assert (out.double() - o64).abs().max().item() <= 2.5e-2      # one bf16 ulp at |out| ~ 4
    assert float((out.double().flatten() @ o64.flatten()) / (out.double().norm() * o64.norm())) >= 0.99999
    h2 = ops.gemm_nt(x, w1, bias=b1, epilogue=ops.EPI_RELU)
    o2 = ops.gemm_nt(h2, w2, bias=b2, epilogue=ops.EPI_RESID, aux=res)
    assert (out.float() - o2.float()).abs().max().item() <= 3.2e-2  # both are roundings of the same fp32 sums
    if write_h:
        assert not torch.isnan(h.float()).any()
        assert (h.double() - h64).abs().max().item() <= 2e-2
        assert (h.float() - h2.float()).abs().max().item() <= 1.6e-2
    assert ops.ffn_packed_bytes(768, 2048) < 0 and ops.ffn_packed_bytes(192, 2000) < 0 and ops.ffn_packed_bytes(384, 2048) > 0
    # no residual
    out_nr = ops.ffn_fwd(x, pk, b1, b2, rows_per_wave=rpw)
    assert (out_nr.double() - (o64 - res.double())).abs().max().item() <= 2.5e-2


def test_weighted_knn_matches_reference_golden_and_oracle():
    """chadavit_amd.utils.knn.WeightedKNNClassifier (HIP vote kernel) vs the reference classifier's golden accuracies and the
    oracle's per-sample class ranking (src/utils/knn.py:96-177).  Index work: the predicted classes must be identical."""
    import os
    import numpy as np
    from chadavit_amd import ops
    from chadavit_amd.utils.knn import WeightedKNNClassifier
    from oracle import chada_ref as R
    from tests.test_oracle_golden import KNN_CASES, _knn_data
    dev = _dev()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_knn_ckpt.npz"))
    xtr, ytr, xte, yte = _knn_data(71, 1500, 400, 64, 10, 4.5)
    for tag, k, T, fx in KNN_CASES:
        m = WeightedKNNClassifier(k=k, T=T, distance_fx=fx, max_distance_matrix_size=150 * 1500)  # 3 chunks of test samples
        m.update(train_features=xtr[:700].to(dev), train_targets=ytr[:700].to(dev))
        m.update(train_features=xtr[700:].to(dev), train_targets=ytr[700:].to(dev), test_features=xte.to(dev), test_targets=yte.to(dev))
        pred, _ = m.predict(top=5)
        rank, votes = R.knn_predict(xtr, ytr, xte, k, T, fx, num_classes=10)
        agree = (pred.cpu().long() == rank[:, :5]).float().mean().item()
        assert agree >= 0.995, (tag, agree)  # identical up to fp32 near-ties in the vote mass
        top1, top5 = m.compute()
        assert abs(top1 - float(g[f"{tag}::acc"][0])) <= 0.25 + 1e-9 and abs(top5 - float(g[f"{tag}::acc"][1])) <= 0.25 + 1e-9, (tag, top1, top5)
        assert m.compute() == (-1, -1)  # banks were reset
    # the vote kernel alone, exact ties at the k-th similarity: the first ones in index order are taken
    sims = torch.tensor([[0.5, 0.9, 0.5, 0.5, 0.1, 0.5]], device=dev)
    tt = torch.tensor([0, 1, 2, 3, 4, 5], dtype=torch.int32, device=dev)
    top, votes = ops.knn_vote(sims, tt, 3, 1.0, False, 6, 6, want_votes=True)
    assert torch.equal(votes[0].cpu(), torch.tensor([0.5, 0.9, 0.5, 0.0, 0.0, 0.0]))
    assert top[0].tolist()[:3] == [1, 0, 2]


def test_channel_jitter_matches_reference_golden():
    """chadavit_channel_jitter vs the reference's CustomColorJitter.apply output (golden) on the collated layout, plus the
    fused horizontal flip and the identity draw."""
    import os
    import numpy as np
    from chadavit_amd.data.gpu_augment import ChannelJitter
    from oracle import procedural as P
    dev = _dev()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "jitter.npz"))
    H, W, C = int(g["H"]), int(g["W"]), int(g["C"])
    img = (P.tensor((H, W, C), "jitter.img", 0.5, seed=int(g["seed_img"])).numpy() * 0.5 + 0.5).astype(np.float32)
    sq = img[:, :H, :]                                              # the kernel works on square crops
    for k in range(2):
        x = torch.from_numpy(np.ascontiguousarray(sq.transpose(2, 0, 1)))[:, None].contiguous().to(dev)   # (C, 1, S, S)
        jit = ChannelJitter(p=1.0)
        out = jit(x.clone(), [C], params=(g[f"shifts{k}"], g[f"gammas{k}"], None))
        ref = g[f"out{k}"][:, :H, :].transpose(2, 0, 1)
        np.testing.assert_allclose(out[:, 0].cpu().numpy(), ref, atol=1e-6)
        flipped = jit(x.clone(), [C], params=(g[f"shifts{k}"], g[f"gammas{k}"], np.array([1, 0, 1, 0, 1], dtype=np.uint8)))
        exp = ref.copy()
        exp[[0, 2, 4]] = exp[[0, 2, 4]][:, :, ::-1]
        np.testing.assert_allclose(flipped[:, 0].cpu().numpy(), exp, atol=1e-6)
    ident = ChannelJitter(p=0.0)(x.clone(), [C], rng=np.random.RandomState(0))
    assert torch.equal(ident, x.clamp(0, 1))
    rs = np.random.RandomState(3)
    sh, gm, fl = ChannelJitter(p=1.0).sample([2, 3], rs)
    rs2 = np.random.RandomState(3)
    rs2.uniform(); a = rs2.uniform(-0.3, 0.3, 2); b = rs2.uniform(0.5, 1.5, 2)
    assert np.allclose(sh[:2], a) and np.allclose(gm[:2], b) and fl.sum() == 0


@pytest.mark.parametrize("M,save,two,D", [(77, True, True, 192), (1000, False, True, 192), (4099, True, False, 192), (4099, False, False, 192),
                                          (77, True, True, 384), (4099, False, True, 384), (1000, True, False, 384)])
def test_fused_ffn_layernorm_tail(M, save, two, D):
    """ops.ffn_ln_fwd = ops.ffn_fwd followed by the stand-alone LayerNorm kernels (norm2, and the next block's norm1): same
    z, H bit for bit; X2 / Hn equal up to the fp32 summation order of the row statistics (<= 1 bf16 ulp on a few elements)."""
    from chadavit_amd import ops
    dev = _dev()
    FF = 2048
    gen = torch.Generator(device="cpu").manual_seed(M + 5)
    bf = torch.bfloat16
    x = torch.randn((M, D), generator=gen).to(dev).to(bf)
    w1 = (torch.randn((FF, D), generator=gen) / D ** 0.5).to(dev).to(bf)
    w2 = (torch.randn((D, FF), generator=gen) / FF ** 0.5).to(dev).to(bf)
    b1 = (torch.randn(FF, generator=gen) * 0.1).to(dev)
    b2 = (torch.randn(D, generator=gen) * 0.1).to(dev)
    ga, ba = (1 + 0.2 * torch.randn(D, generator=gen)).to(dev), (0.2 * torch.randn(D, generator=gen)).to(dev)
    gb, bb = (1 + 0.2 * torch.randn(D, generator=gen)).to(dev), (0.2 * torch.randn(D, generator=gen)).to(dev)
    pk = ops.ffn_pack(w1, w2)
    h_ref = torch.empty((M, FF), device=dev, dtype=bf)
    z_ref = ops.ffn_fwd(x, pk, b1, b2, resid=x, h=h_ref)
    m1, r1, m2, r2 = (torch.empty(M, device=dev) for _ in range(4))
    if two:
        x2_ref, hn_ref = ops.layernorm_fwd2(z_ref, ga, ba, gb, bb, 1e-5, 1e-6, stats1=(m1, r1), stats2=(m2, r2))
    else:
        x2_ref, hn_ref = ops.layernorm_fwd(z_ref, ga, ba, 1e-5, mean=m1, rstd=r1), None
    z = torch.full((M, D), float("nan"), device=dev, dtype=bf) if save else None
    h = torch.full((M, FF), float("nan"), device=dev, dtype=bf) if save else None
    sa = (torch.empty(M, device=dev), torch.empty(M, device=dev)) if save else None
    sb = (torch.empty(M, device=dev), torch.empty(M, device=dev)) if (save and two) else None
    x2, hn = ops.ffn_ln_fwd(x, pk, b1, b2, (ga, ba, 1e-5), resid=x, z=z, h=h, ln_b=(gb, bb, 1e-6) if two else None, stats_a=sa, stats_b=sb)
    if save:
        assert torch.equal(z, z_ref) and torch.equal(h, h_ref)
        assert torch.allclose(sa[0], m1, atol=1e-5) and torch.allclose(sa[1], r1, rtol=1e-5)
        if two:
            assert torch.allclose(sb[0], m2, atol=1e-5) and torch.allclose(sb[1], r2, rtol=1e-5)
    d = (x2.float() - x2_ref.float()).abs()
    assert d.max().item() <= 3.2e-2 and (d > 0).float().mean().item() < 0.02
    if two:
        d2 = (hn.float() - hn_ref.float()).abs()
        assert d2.max().item() <= 3.2e-2 and (d2 > 0).float().mean().item() < 0.03
    else:
        assert hn is None


def _mx_dequant(q, scales):
    """(R, K) e4m3 bytes + (K/32, R) E8M0 bytes -> fp32 (R, K): the value the MFMA multiplies.  Power-of-two scaling done on the
    CPU in float64 (exact); a GPU pow / ldexp is not."""
    vals = q.cpu().view(torch.float8_e4m3fn).double()
    ex = (scales.cpu().double() - 127.0).t().repeat_interleave(32, dim=1)   # (R, K)
    return (vals * torch.pow(torch.tensor(2.0, dtype=torch.float64), ex)).float().to(q.device)


@pytest.mark.parametrize("R,K", [(300, 768), (129, 2048), (5, 128)])
def test_mx8_quantize(R, K):
    """OCP-MX fp8 quantiser: per row and 32-k block one power-of-two scale 2^(floor(log2 amax) - 8) (+1 if the largest element would
    exceed 448), elements rounded (RNE) to e4m3: checked against that definition evaluated exactly (float64, CPU), incl. an
    all-zero block, a block at the 448 limit and the ReLU variant."""
    from chadavit_amd import ops
    dev = _dev()
    x = (_rand((R, K), 91, 1.0) * torch.exp(_rand((R, 1), 92, 2.0))).bfloat16().to(dev)
    x[0, :32] = 0
    x[1, 32:64] = 448.0
    two = torch.tensor(2.0, dtype=torch.float64)
    for relu in (False, True):
        q, sc = ops.mx8_quantize(x, relu=relu)
        xf = (torch.relu(x.float()) if relu else x.float()).cpu().double()
        blk = xf.view(R, K // 32, 32)
        amax = blk.abs().amax(-1)
        ex = torch.floor(torch.log2(amax.clamp_min(1e-300))) - 8
        ex = torch.where(amax * torch.pow(two, -ex) > 448.0, ex + 1, ex)
        ex = torch.where(amax > 0, ex, torch.zeros_like(ex))
        assert torch.equal(sc.cpu().t().double() - 127, ex), float((sc.cpu().t().double() - 127 - ex).abs().max())
        scaled = (blk * torch.pow(two, -ex)[..., None]).view(R, K).float()   # exact: a bf16 value times a power of two
        ref = scaled.to(torch.float8_e4m3fn).float()
        got = q.cpu().view(torch.float8_e4m3fn).float()
        bad = got != ref
        assert not bool(bad.any()), (int(bad.sum()), scaled[bad][:8].tolist(), got[bad][:8].tolist(), ref[bad][:8].tolist())
        deq = _mx_dequant(q, sc).cpu().double()
        err = (deq - xf).abs().view(R, K // 32, 32)
        assert bool((err <= amax[..., None] * 2.0 ** -4 + 1e-30).all())   # half an e4m3 ulp of the block's largest binade


@pytest.mark.parametrize("M,N,K", [(1000, 2304, 768), (257, 768, 768), (300, 2048, 768), (130, 768, 2048), (64, 128, 128),
                                   (125504, 2048, 768), (125504, 768, 2048)])   # (the last two: cfg5's FFN shapes at bench rows)
def test_gemm_nt_mx8(M, N, K):
    """MX-scaled fp8 MFMA GEMM vs fp32 torch on the DEQUANTISED operands (so only the accumulation order differs), all epilogues;
    and vs the unquantised bf16 product within the quantisation noise (rel-L2 <= 6e-2: two e4m3 operands, 3 mantissa bits each)."""
    from chadavit_amd import ops
    dev = _dev()
    x = _rand((M, K), 93, 1.0).bfloat16().to(dev)
    w = (_rand((N, K), 94, 1.0) / math.sqrt(K)).bfloat16().to(dev)
    bias = _rand((N,), 95, 0.5).to(dev)
    aux = _rand((M, N), 96, 1.0).bfloat16().to(dev)
    xq, xs = ops.mx8_quantize(x)
    wq, ws = ops.mx8_quantize(w)
    ref = _mx_dequant(xq, xs) @ _mx_dequant(wq, ws).t()
    out = ops.gemm_nt_mx8(xq, xs, wq, ws)
    _close(out, ref, 1e-2, 1e-2, "none")
    out = ops.gemm_nt_mx8(xq, xs, wq, ws, bias=bias, epilogue=ops.EPI_RELU)
    _close(out, torch.relu(ref + bias), 1e-2, 1e-2, "relu")
    out = ops.gemm_nt_mx8(xq, xs, wq, ws, bias=bias, epilogue=ops.EPI_RESID, aux=aux)
    _close(out, ref + bias + aux.float(), 1e-2, 2e-2, "resid")
    # threshold_backward (the dX GEMM behind the ReLU): bit-exact masking of the plain product; its quantised copy bit-identical to
    # mx8_quantize of the masked bf16 result
    plain = ops.gemm_nt_mx8(xq, xs, wq, ws)
    out, (oq, osc) = ops.gemm_nt_mx8(xq, xs, wq, ws, epilogue=ops.EPI_RELUMASK, aux=aux, emit_q=True)
    assert torch.equal(out, torch.where(aux > 0, plain, torch.zeros_like(plain))), "relumask"
    rq, rs = ops.mx8_quantize(out)
    assert torch.equal(oq, rq) and torch.equal(osc, rs)
    exact = x.float() @ w.float().t()
    rel = float((ref - exact).norm() / exact.norm())
    assert rel <= 6e-2, rel


@pytest.mark.parametrize("M", [1000, 257])
def test_gemm_nt_mx8_emits_the_next_operand(M):
    """The ReLU epilogue of the MX-fp8 GEMM hands the next GEMM its operand: (bytes, scales) bit-identical to mx8_quantize applied to the
    bf16 result, with and without the bf16 result being written."""
    from chadavit_amd import ops
    dev = _dev()
    N, K = 3072, 768
    x = _rand((M, K), 201, 1.0).bfloat16().to(dev)
    w = (_rand((N, K), 202, 1.0) / math.sqrt(K)).bfloat16().to(dev)
    bias = _rand((N,), 203, 0.2).to(dev)
    xq, xs = ops.mx8_quantize(x)
    wq, ws = ops.mx8_quantize(w)
    ref = ops.gemm_nt_mx8(xq, xs, wq, ws, bias=bias, epilogue=ops.EPI_RELU)
    rq, rs = ops.mx8_quantize(ref)
    out, (q, sc) = ops.gemm_nt_mx8(xq, xs, wq, ws, bias=bias, epilogue=ops.EPI_RELU, emit_q=True)
    assert torch.equal(out, ref) and torch.equal(q, rq) and torch.equal(sc, rs)
    none, (q2, sc2) = ops.gemm_nt_mx8(xq, xs, wq, ws, bias=bias, epilogue=ops.EPI_RELU, emit_q=True, want_out=False)
    assert none is None and torch.equal(q2, rq) and torch.equal(sc2, rs)


@pytest.mark.parametrize("T,D", [(1000, 768), (333, 384), (77, 192)])
def test_layernorm_fwd_emits_fp8_operand(T, D):
    """LayerNorm forward kernels that also emit their output as an OCP-MX fp8 operand: bit-identical to mx8_quantize of the bf16 output
    (single norm and the second output of the chained pair), the bf16 outputs and statistics unchanged."""
    from chadavit_amd import ops
    dev = _dev()
    x = _rand((T, D), 301, 2.0).bfloat16().to(dev)
    ga, ba = (1 + _rand((D,), 302, 0.2)).to(dev), _rand((D,), 303, 0.2).to(dev)
    gb, bb = (1 + _rand((D,), 304, 0.2)).to(dev), _rand((D,), 305, 0.2).to(dev)
    y = ops.layernorm_fwd(x, ga, ba, 1e-5)
    rq, rs = ops.mx8_quantize(y)
    y2, (q, sc) = ops.layernorm_fwd(x, ga, ba, 1e-5, emit_q=True)
    assert torch.equal(y2, y) and torch.equal(q, rq) and torch.equal(sc, rs)
    a1, a2 = ops.layernorm_fwd2(x, ga, ba, gb, bb, 1e-5, 1e-5)
    r2q, r2s = ops.mx8_quantize(a2)
    b1, b2, (q2, s2) = ops.layernorm_fwd2(x, ga, ba, gb, bb, 1e-5, 1e-5, emit_q=True)
    assert torch.equal(b1, a1) and torch.equal(b2, a2) and torch.equal(q2, r2q) and torch.equal(s2, r2s)


def test_tokenizer_bwd_more_images_than_one_list_window():
    """Tokenizer backward reductions with more sequences than one 4096-row list window (the CLS share and every channel slot go through
    in windows): against plain torch sums over the packed gradient rows."""
    from chadavit_amd import ops
    from chadavit_amd.ragged import RaggedBatch
    dev = _dev()
    p, D, max_c = 4, 192, 10
    nch = [1 + (i * 7) % 3 for i in range(4500)]          # 4500 images of 1-3 channels: 4500 CLS rows, ~3000 rows in slot 1
    rb = RaggedBatch(nch, p, dev)
    dtok = _rand((rb.T, D), 401, 1.0).bfloat16().to(dev)
    dpatch, dpos, dchan, dcls = ops.tokenizer_bwd(dtok, rb.cu_seqlens, rb.chan_img, rb.chan_idx, p, max_c)
    g = dtok.float().cpu()
    cu = rb.host_cu_seqlens
    ref_cls = torch.zeros(D); ref_pos = torch.zeros(p + 1, D); ref_chan = torch.zeros(max_c, D)
    for b, c in enumerate(nch):
        r0 = cu[b]
        ref_cls += g[r0]
        ref_pos[0] += g[r0]
        for k in range(c):
            blk = g[r0 + 1 + k * p:r0 + 1 + (k + 1) * p]
            ref_pos[1:] += blk
            ref_chan[k] += blk.sum(0)
    def rel(a, b):
        return float((a.cpu().float() - b).abs().max() / b.abs().max())
    assert rel(dcls.view(-1), ref_cls) < 1e-4
    assert rel(dchan.view(max_c, D)[:3], ref_chan[:3]) < 1e-4 and float(dchan.view(max_c, D)[3:].abs().max()) == 0.0
    assert rel(dpos.view(p, D), ref_pos[1:]) < 1e-4


# (the fourth case: 2.3 x the bench's rows -- the hidden activation is 2.9 G elements / 5.8 GB, past 32-bit element AND byte offsets)
@pytest.mark.parametrize("D,M,ch,n_seq", [(192, 603136, 3, 1024), (384, 278462, 10, 142), (768, 125504, 10, 64), (192, 1413600, 3, 2400)])
def test_hot_kernels_at_bench_rows_equal_their_small_runs(D, M, ch, n_seq):
    """BASELINE-size launches (cfg2 at 512 images per GPU: 603 136 token rows, 1024 sequences of 589 tokens) through a
    size-independent property: every row / sequence is processed independently of its position, so a launch over R copies
    of a small input must reproduce the small launch's output in every copy BIT FOR BIT -- and the small launches are the ones
    the tests above hold against fp32 torch.  Covers the whole-block forward kernel (training instance with every output), the
    fused FFN backward from its ReLU bits, the attention forward / backward, and (sum over rows: fp32 tolerance) the
    weight-gradient TN GEMM at 603 136 x 192 x 2048; likewise at Small's (cfg3: 142 sequences of 1961 tokens, the D = 384 builds of the
    kernels) and Base's (cfg5: 64 x 1961; no whole-block kernel at D = 768, attention + TN only) bench shapes."""
    from chadavit_amd import ops
    from chadavit_amd.ragged import RaggedBatch
    dev = _dev()
    FF, M0 = 2048, 1024
    assert M == n_seq * (1 + 196 * ch)   # cfg2: 603136 = 589 * 1024 rows = 4712 blocks of 128 rows
    rep = lambda t: t.repeat((M + M0 - 1) // M0, 1)[:M].contiguous()
    dz0 = _rand((M0, D), 181, 1.0).bfloat16().to(dev)
    if D <= 384:
        a0 = _rand((M0, D), 151, 1.0).bfloat16().to(dev)
        x0 = _rand((M0, D), 152, 1.0).bfloat16().to(dev)
        wo = (_rand((D, D), 153, 1.0) / math.sqrt(D)).bfloat16().to(dev)
        w1 = (_rand((FF, D), 154, 1.0) / math.sqrt(D)).bfloat16().to(dev)
        w2 = (_rand((D, FF), 155, 1.0) / math.sqrt(FF)).bfloat16().to(dev)
        wq = (_rand((3 * D, D), 159, 1.0) / math.sqrt(D)).bfloat16().to(dev)
        bo, b1, b2, bq = _rand((D,), 156, 0.1).to(dev), _rand((FF,), 157, 0.1).to(dev), _rand((D,), 158, 0.1).to(dev), _rand((3 * D,), 161, 0.1).to(dev)
        lns = [((1 + _rand((D,), 160 + i, 0.2)).to(dev), _rand((D,), 170 + i, 0.2).to(dev), 1e-5) for i in range(3)]
        slab = torch.cat([w1.reshape(-1), w2.reshape(-1), wo.reshape(-1), wq.reshape(-1)])
        pkq = torch.empty(ops.ffn_proj_packed_bytes(D, FF) // 2, device=dev, dtype=torch.bfloat16)
        o1, o2 = w1.numel(), w1.numel() + w2.numel()
        ops.ffn_pack_proj_batched(slab, pkq, torch.tensor([0, o1, o2, o2 + wo.numel(), 0], device=dev, dtype=torch.int64), 1, D, FF)

        def block(a, x):
            m = a.shape[0]
            e = lambda *s: torch.empty(s, device=dev, dtype=torch.bfloat16)
            st = [(torch.empty(m, device=dev), torch.empty(m, device=dev)) for _ in range(3)]
            y, z, h, bits = e(m, D), e(m, D), e(m, FF), ops.relu_bits_buffer(m, FF, dev)
            x1, x2, hn, qkv = ops.proj_ffn_ln_fwd(a, x, pkq, bo, lns[0], b1, b2, lns[1], y=y, stats1=st[0], z=z, h=h, ln_b=lns[2],
                                                  stats_a=st[1], stats_b=st[2], qkv_bias=bq, want_hn=True, relu_bits=bits)
            return dict(x1=x1, x2=x2, hn=hn, qkv=qkv, y=y, z=z, h=h, m1=st[0][0], r1=st[0][1], ma=st[1][0], ra=st[1][1], mb=st[2][0],
                        rb=st[2][1]), bits

        small, bits0 = block(a0, x0)
        big, bits = block(rep(a0), rep(x0))
        for k, v in big.items():
            ref = rep(small[k]) if small[k].dim() == 2 else small[k].repeat((M + M0 - 1) // M0)[:M]
            assert torch.equal(v, ref), k
        del big
        # fused FFN backward from the bits the big launch recorded
        pkb = ops.ffn_pack(w2.t().contiguous(), w1.t().contiguous())
        dpre0 = torch.empty((M0, FF), device=dev, dtype=torch.bfloat16)
        dx0 = ops.ffn_bwd_dx(dz0, pkb, bits0, dpre=dpre0)
        dpre = torch.empty((M, FF), device=dev, dtype=torch.bfloat16)
        dx = ops.ffn_bwd_dx(rep(dz0), pkb, bits, dpre=dpre)
        assert torch.equal(dx, rep(dx0)) and torch.equal(dpre, rep(dpre0))
    else:
        dpre0 = _rand((M0, FF), 182, 1.0).bfloat16().to(dev)
        dpre = rep(dpre0)
    # weight-gradient TN GEMM over all rows: (M / M0) x the small product (fp32 partial sums in another order)
    ws = torch.empty(24 * 1024 * 1024, device=dev)   # the backbone's own split-T workspace size
    c_big = torch.zeros((D, FF), device=dev)
    c_small = torch.zeros((D, FF), device=dev)
    cs_big, cs_small = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ops.gemm_tn(rep(dz0), dpre, c_big, colsum=cs_big, workspace=ws)
    ops.gemm_tn(dz0, dpre0, c_small, colsum=cs_small, workspace=ws)
    c_ref, cs_ref = (M // M0) * c_small, (M // M0) * cs_small
    if M % M0:   # the partial last copy
        c_t, cs_t = torch.zeros((D, FF), device=dev), torch.zeros(D, device=dev)
        ops.gemm_tn(dz0[:M % M0].contiguous(), dpre0[:M % M0].contiguous(), c_t, colsum=cs_t, workspace=ws)
        c_ref, cs_ref = c_ref + c_t, cs_ref + cs_t
    assert float((c_big - c_ref).norm() / c_ref.norm()) < 1e-4
    assert float((cs_big - cs_ref).norm() / cs_ref.norm()) < 1e-4
    del dpre, c_big
    # attention: n_seq copies of one sequence (cfg2: 1024 x 589 tokens, 2 heads of 96)
    rb1, rbn = RaggedBatch([ch], 196, dev), RaggedBatch([ch] * n_seq, 196, dev)
    assert rbn.T == M
    n1 = 1 + 196 * ch
    q0 = _rand((n1, 3 * D), 191, 1.0).bfloat16().to(dev)
    do0 = _rand((n1, D), 192, 1.0).bfloat16().to(dev)
    out0, lse0 = ops.attn_fwd(q0, rb1.cu_seqlens, rb1.work, 2)
    dq0 = ops.attn_bwd(q0, out0, do0, lse0, rb1.cu_seqlens, rb1.work, 2)
    qn, don = q0.repeat(n_seq, 1), do0.repeat(n_seq, 1)
    outn, lsen = ops.attn_fwd(qn, rbn.cu_seqlens, rbn.work, 2)
    dqn = ops.attn_bwd(qn, outn, don, lsen, rbn.cu_seqlens, rbn.work, 2)
    assert torch.equal(outn, out0.repeat(n_seq, 1)) and torch.equal(lsen, lse0.repeat(1, n_seq))
    assert torch.equal(dqn, dq0.repeat(n_seq, 1))


@pytest.mark.parametrize("nch,p,D,H", [([3, 1, 10, 5], 196, 192, 2), ([1, 2], 36, 192, 2), ([2, 1, 1], 36, 384, 2), ([1], 4, 64, 2),
                                       ([3, 2], 36, 128, 2), ([2, 1], 36, 768, 2), ([1, 10], 196, 768, 2), ([2, 3], 36, 192, 12),
                                       ([1] * 70, 1, 192, 2), ([2, 1, 3], 36, 256, 2), ([1, 2], 196, 512, 2)])
def test_attention_of_the_cls_rows(nch, p, D, H):
    """chadavit_attn_cls_fwd / _bwd against fp32 torch attention over all rows: the CLS rows of the output, and the gradient of a loss
    that reads only those rows (dQ exactly zero elsewhere, dK / dV of every row), and against the flash kernels' CLS rows."""
    from chadavit_amd import ops
    from chadavit_amd.ragged import RaggedBatch
    dev = _dev()
    rb = RaggedBatch(nch, p, dev)
    T = rb.T
    cls = rb.cls_rows.long()
    qkv = _rand((T, 3 * D), 12, 1.0).bfloat16().to(dev)
    qkv[0, :D] *= 4.0   # a peaky softmax for the first image's CLS query
    dout_cls = _rand((len(nch), D), 13, 1.0).bfloat16().to(dev)
    out_cls, lse_cls = ops.attn_cls_fwd(qkv, rb.cu_seqlens, H)
    qf = qkv.float().requires_grad_(True)
    ref = _attn_ref(qf, rb.host_cu_seqlens, H)
    _close(out_cls, ref[cls], 1e-2, 1e-2, "attn cls fwd")
    full, lse_full = ops.attn_fwd(qkv, rb.cu_seqlens, rb.work, H)
    _close(out_cls, full[cls].float(), 2e-2, 2e-2, "attn cls fwd vs flash")
    assert torch.allclose(lse_cls, lse_full[:, cls], atol=2e-2)
    dfull = torch.zeros((T, D), device=dev)
    dfull[cls] = dout_cls.float()
    ref.backward(dfull)
    dqkv = ops.attn_cls_bwd(qkv, rb.cu_seqlens, out_cls, dout_cls, lse_cls, H)
    g = qf.grad
    scale = g.abs().max().item()
    _close(dqkv[:, 2 * D:], g[:, 2 * D:], 2e-2, 2e-2 * scale, "dV")
    _close(dqkv[:, D:2 * D], g[:, D:2 * D], 2e-2, 2e-2 * scale, "dK")
    _close(dqkv[cls, :D], g[cls, :D], 2e-2, 2e-2 * scale, "dQ cls")
    notcls = torch.ones(T, dtype=torch.bool, device=dev)
    notcls[cls] = False
    assert float(dqkv[notcls, :D].float().abs().max()) == 0.0 if notcls.any() else True


@pytest.mark.parametrize("D,M", [(192, 603136), (384, 278462), (768, 125504)])
def test_row_wise_kernels_at_bench_rows_equal_their_small_runs(D, M):
    """The HBM-bound row-wise kernels at the bench's row counts, by the same replica property as the hot kernels above: LayerNorm
    forward / backward (+ residual), the chained LayerNorm pair backward, the QKV-backward / out-proj-backward GEMMs and the CLS gather
    / scatter reproduce their 1024-row launches bit for bit in every copy; the column sums (dgamma / dbeta) scale with the copies."""
    from chadavit_amd import ops
    dev = _dev()
    M0 = 1024
    reps = (M + M0 - 1) // M0
    rep = lambda t: t.repeat(reps, 1)[:M].contiguous() if t.dim() == 2 else t.repeat(reps)[:M].contiguous()
    x0 = _rand((M0, D), 301, 2.0).bfloat16().to(dev)
    dy0 = _rand((M0, D), 302, 1.0).bfloat16().to(dev)
    dres0 = _rand((M0, D), 303, 1.0).bfloat16().to(dev)
    gamma, beta = (1 + _rand((D,), 304, 0.2)).to(dev), _rand((D,), 305, 0.2).to(dev)
    ws = ops.layernorm_bwd_workspace(D, dev)

    def ln(x, dy, dres):
        m = x.shape[0]
        mean, rstd = torch.empty(m, device=dev), torch.empty(m, device=dev)
        y = ops.layernorm_fwd(x, gamma, beta, 1e-5, mean=mean, rstd=rstd)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        dx = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, ws, dres=dres)
        return y, mean, rstd, dx, dg, db

    ys, ms, rs, dxs, dgs, dbs = ln(x0, dy0, dres0)
    yb, mb, rb_, dxb, dgb, dbb = ln(rep(x0), rep(dy0), rep(dres0))
    assert torch.equal(yb, rep(ys)) and torch.equal(mb, rep(ms)) and torch.equal(rb_, rep(rs)) and torch.equal(dxb, rep(dxs))
    full, tail = M // M0, M % M0
    if tail == 0:
        assert float((dgb - full * dgs).norm() / (full * dgs).norm()) < 1e-4 and float((dbb - full * dbs).norm() / (full * dbs).norm()) < 1e-4
    del yb, dxb
    # GEMMs of the backward: dh = dqkv Wqkv ([M, 3D] x [3D, D]) and da = dy Wo ([M, D] x [D, D])
    wq = (_rand((D, 3 * D), 306, 1.0) / math.sqrt(3 * D)).bfloat16().to(dev)   # = in_proj_weight^T, [N = D, K = 3D]
    wo = (_rand((D, D), 307, 1.0) / math.sqrt(D)).bfloat16().to(dev)
    dqkv0 = _rand((M0, 3 * D), 308, 1.0).bfloat16().to(dev)
    assert torch.equal(ops.gemm_nt(rep(dqkv0), wq), rep(ops.gemm_nt(dqkv0, wq)))
    assert torch.equal(ops.gemm_nt(rep(dy0), wo), rep(ops.gemm_nt(dy0, wo)))
    assert torch.equal(ops.gemm_nt(rep(dy0), wo, epilogue=ops.EPI_RESID, aux=rep(dres0)), rep(ops.gemm_nt(dy0, wo, epilogue=ops.EPI_RESID, aux=dres0)))
    # CLS gather / scatter over the bench's sequence count
    n_seq = M // 1961 if D > 192 else M // 589
    rows = (torch.arange(n_seq, device=dev, dtype=torch.int32) * (M // n_seq)).contiguous()
    big = rep(x0)
    g = ops.gather_rows(big, rows)
    assert torch.equal(g, big[rows.long()])
    sc = ops.scatter_rows_zero(g, rows, M)
    ref = torch.zeros_like(big)
    ref[rows.long()] = g
    assert torch.equal(sc, ref)

@pytest.mark.parametrize("N,f32in", [(4, True), (64, True), (1000, False), (5120, True)])
def test_batchnorm_kernels_match_torch(N, f32in):
    """BatchNorm1d of the head's projector (use_bn_in_head; reference src/methods/dino.py:59-77): statistics, running estimates,
    normalise + affine + GELU, and the backward, against torch.nn.functional.batch_norm in fp32 on the same inputs."""
    from chadavit_amd import ops
    dev = _dev()
    C = 2048
    z = _rand((N, C), 401, 1.5).to(dev) + _rand((C,), 402, 2.0).to(dev)
    if not f32in:
        z = z.bfloat16()
    zf = z.float()
    ga, be = (1 + _rand((C,), 403, 0.3)).to(dev), _rand((C,), 404, 0.3).to(dev)
    rm, rv = _rand((C,), 405, 0.1).to(dev), (1 + _rand((C,), 406, 0.1)).to(dev)
    rm_t, rv_t = rm.clone(), rv.clone()
    mean, rstd = ops.bn_stats(z, 1e-5, rm, rv, 0.1)
    ref = torch.nn.functional.batch_norm(zf, rm_t, rv_t, ga, be, True, 0.1, 1e-5)
    assert torch.allclose(mean, zf.mean(0), atol=2e-6, rtol=1e-5)
    assert torch.allclose(rstd, (zf.var(0, unbiased=False) + 1e-5).rsqrt(), rtol=2e-4)
    assert torch.allclose(rm, rm_t, atol=1e-6, rtol=1e-5) and torch.allclose(rv, rv_t, atol=1e-6, rtol=2e-4)
    pre, act = ops.bn_apply_gelu(z, mean, rstd, ga, be)
    _close(pre, ref, 1e-2, 1e-2, "bn output")
    _close(act, torch.nn.functional.gelu(pre.float()), 1e-2, 1e-2, "gelu")
    dy = _rand((N, C), 407, 1.0).bfloat16().to(dev)
    zr, gar, ber = zf.clone().requires_grad_(True), ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
    torch.nn.functional.batch_norm(zr, None, None, gar, ber, True, 0.1, 1e-5).backward(dy.float())
    dga, dbe = torch.full((C,), 3.0, device=dev), torch.full((C,), -2.0, device=dev)
    dz = ops.bn_bwd(dy, z, mean, rstd, ga, dga, dbe, accumulate=False)
    assert float((dz.float() - zr.grad).norm() / zr.grad.norm()) <= 5e-3
    assert torch.allclose(dga, gar.grad, rtol=1e-3, atol=1e-3 * float(gar.grad.abs().max()))
    assert torch.allclose(dbe, ber.grad, rtol=1e-3, atol=1e-3 * float(ber.grad.abs().max()))
    ops.bn_bwd(dy, z, mean, rstd, ga, dga, dbe, accumulate=True)
    assert torch.allclose(dga, 2 * gar.grad, rtol=1e-3, atol=2e-3 * float(gar.grad.abs().max()))


def _same(a, b):
    if isinstance(a, (tuple, list)):
        return all(_same(x, y) for x, y in zip(a, b) if x is not None)
    return a is None or torch.equal(a, b)


@pytest.mark.parametrize("D,M,ch,n_seq", [(192, 603136, 3, 1024), (192, 26282, 10, 0), (384, 278462, 10, 142), (768, 125504, 10, 64)])
def test_hot_kernels_are_deterministic(D, M, ch, n_seq):
    """Every hot entry point, launched five times on the same inputs at the bench's row counts, must reproduce its first result BIT
    FOR BIT: the kernels order their LDS-DMA rings with hand-counted `s_waitcnt vmcnt(N)`, and a count that is one stage too
    generous reads a weight block that is still landing -- silently, only at sizes where HBM is busy, and differently every run
    (this test found exactly that in the no-grad block kernel's QKV postlogue).  The no-grad block kernel's fused QKV is also held
    against the stand-alone GEMM of its hn output (bit-identical by construction)."""
    from chadavit_amd import ops
    from chadavit_amd.ragged import RaggedBatch
    dev = _dev()
    FF = 2048
    g = torch.Generator(device=dev); g.manual_seed(1234)
    rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc)
    a, x = rnd(M, D).bfloat16(), rnd(M, D).bfloat16()
    wq = rnd(3 * D, D, sc=1 / math.sqrt(D)).bfloat16()
    bq = rnd(3 * D, sc=0.1)
    cases = {}
    if D <= 384:
        wo, w1, w2 = rnd(D, D, sc=1 / math.sqrt(D)).bfloat16(), rnd(FF, D, sc=1 / math.sqrt(D)).bfloat16(), rnd(D, FF, sc=1 / math.sqrt(FF)).bfloat16()
        bo, b1, b2 = rnd(D, sc=0.1), rnd(FF, sc=0.1), rnd(D, sc=0.1)
        lns = [(1 + rnd(D, sc=0.2), rnd(D, sc=0.2), 1e-5) for _ in range(3)]
        slab = torch.cat([w1.reshape(-1), w2.reshape(-1), wo.reshape(-1), wq.reshape(-1)])
        pkq = torch.empty(ops.ffn_proj_packed_bytes(D, FF) // 2, device=dev, dtype=torch.bfloat16)
        o1, o2 = w1.numel(), w1.numel() + w2.numel()
        ops.ffn_pack_proj_batched(slab, pkq, torch.tensor([0, o1, o2, o2 + wo.numel(), 0], device=dev, dtype=torch.int64), 1, D, FF)
        cases["block no-grad + qkv"] = lambda: ops.proj_ffn_ln_fwd(a, x, pkq, bo, lns[0], b1, b2, lns[1], ln_b=lns[2], want_x1=False, want_hn=False, qkv_bias=bq)
        cases["block no-grad"] = lambda: ops.proj_ffn_ln_fwd(a, x, pkq, bo, lns[0], b1, b2, lns[1], ln_b=lns[2], want_x1=False, want_hn=True)

        def train():
            e = lambda *s: torch.empty(s, device=dev, dtype=torch.bfloat16)
            st = [(torch.empty(M, device=dev), torch.empty(M, device=dev)) for _ in range(3)]
            y, z, h, bits = e(M, D), e(M, D), e(M, FF), ops.relu_bits_buffer(M, FF, dev)
            r = ops.proj_ffn_ln_fwd(a, x, pkq, bo, lns[0], b1, b2, lns[1], y=y, stats1=st[0], z=z, h=h, ln_b=lns[2], stats_a=st[1], stats_b=st[2],
                                    qkv_bias=bq, want_hn=True, relu_bits=bits)
            return tuple(r) + (y, z, h, bits, st[0][0], st[2][1])
        cases["block training"] = train
        pkb = ops.ffn_pack(w2.t().contiguous(), w1.t().contiguous())
        bits0 = train()[7]   # the ReLU record of the forward (a random byte pattern would also put bits where no row is)

        def bwd():
            dpre = torch.empty((M, FF), device=dev, dtype=torch.bfloat16)
            return ops.ffn_bwd_dx(a, pkb, bits0, dpre=dpre), dpre
        cases["ffn_bwd_dx"] = bwd
    else:
        xq, xs = ops.mx8_quantize(a)
        wq8, ws8 = ops.mx8_quantize(wq)
        cases["gemm_nt_mx8 qkv"] = lambda: ops.gemm_nt_mx8(xq, xs, wq8, ws8, bias=bq)
        cases["gemm_nt_mx8 resid + q"] = lambda: ops.gemm_nt_mx8(xq, xs, wq8[:D], ws8[:, :D], bias=bq[:D].contiguous(), epilogue=ops.EPI_RESID, aux=x, emit_q=True)
    cases["gemm_nt qkv"] = lambda: ops.gemm_nt(a, wq, bias=bq)
    big = rnd(M, 3 * D).bfloat16()
    cases["gemm_nt dX of qkv"] = lambda: ops.gemm_nt(big, wq.t().contiguous())
    ws = torch.empty(max(24 * 1024 * 1024, 32 * (3 * D * D + 3 * D)), device=dev)

    def tn():
        c, cs = torch.empty((3 * D, D), device=dev), torch.empty(3 * D, device=dev)
        ops.gemm_tn(big, a, c, colsum=cs, workspace=ws)
        return c, cs
    cases["gemm_tn dW_qkv"] = tn
    ga, be = 1 + rnd(D, sc=0.2), rnd(D, sc=0.2)
    cases["layernorm_fwd"] = lambda: ops.layernorm_fwd(a, ga, be, 1e-5)
    nch = [ch] * n_seq if n_seq else [10, 10, 10, 10, 10, 8, 5, 3, 1] * 2
    rb = RaggedBatch(nch, 196, dev)
    assert rb.T == M
    out0, lse0 = ops.attn_fwd(big, rb.cu_seqlens, rb.work, 2)
    cases["attn_fwd"] = lambda: ops.attn_fwd(big, rb.cu_seqlens, rb.work, 2)
    cases["attn_bwd"] = lambda: ops.attn_bwd(big, out0, x, lse0, rb.cu_seqlens, rb.work, 2)
    for name, fn in cases.items():
        def flat(r):
            out = []
            for t in (r if isinstance(r, (tuple, list)) else (r,)):
                out += flat(t) if isinstance(t, (tuple, list)) else [t]
            return out
        ref = flat(fn())
        torch.cuda.synchronize()
        ref = [t.clone() if t is not None else None for t in ref]
        for it in range(4):
            junk = torch.full((1 << 26,), float(it), device=dev)   # (another allocator state, a little other traffic)
            got = flat(fn())
            torch.cuda.synchronize()
            for k, (u, v) in enumerate(zip(got, ref)):
                assert u is None or torch.equal(u, v), f"{name}: output {k} of run {it + 1} differs from the first run"
            del junk, got
        if name == "block no-grad + qkv":
            _, x2n, hn = cases["block no-grad"]()
            assert torch.equal(ref[3], ops.gemm_nt(hn, wq, bias=bq)), "fused QKV differs from the GEMM of hn"
            tr_out = cases["block training"]()   # (x1, x2, hn, qkv, ...): the instance the step goldens hold against the reference
            assert torch.equal(ref[1], x2n) and torch.equal(ref[1], tr_out[1]) and torch.equal(hn, tr_out[2]) and torch.equal(ref[3], tr_out[3]), \
                "the no-grad instances' x2 / hn / qkv differ from the training instance's"
            del tr_out
        del ref


def test_gemms_are_deterministic_at_every_shape_of_the_three_models():
    """The GEMM entry points at every (M, N, K) the three model widths launch at bench size (forward, dX, the heads), every epilogue,
    and the weight-gradient GEMM at every (T, I, J): four launches each, bit-identical (see test_hot_kernels_are_deterministic)."""
    from chadavit_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev); g.manual_seed(4321)
    rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc)

    def repeat_equal(name, fn, n=4):
        ref = fn()
        torch.cuda.synchronize()
        ref = [t.clone() for t in (ref if isinstance(ref, (tuple, list)) else (ref,))]
        for it in range(n - 1):
            junk = torch.full((1 << 25,), float(it), device=dev)
            got = fn()
            got = got if isinstance(got, (tuple, list)) else (got,)
            torch.cuda.synchronize()
            for k, (u, v) in enumerate(zip(got, ref)):
                assert torch.equal(u, v), f"{name}: output {k} of launch {it + 2} differs from the first"
            del junk, got

    nt_shapes = [(603136, 576, 192), (603136, 192, 576), (603136, 192, 192), (603136, 2048, 192), (603136, 192, 2048),
                 (278462, 1152, 384), (278462, 384, 1152), (278462, 384, 384),
                 (125504, 2304, 768), (125504, 768, 2304), (125504, 768, 768), (125504, 2048, 768), (125504, 768, 2048),
                 (1024, 2048, 192), (1024, 2048, 2048), (1024, 256, 2048), (1024, 4096, 256), (1024, 65536, 256), (70001, 576, 192)]
    for M, N, K in nt_shapes:
        x, w, bias = rnd(M, K).bfloat16(), rnd(N, K, sc=1 / math.sqrt(K)).bfloat16(), rnd(N, sc=0.1)
        aux = rnd(M, N).bfloat16()
        repeat_equal(f"gemm_nt {M}x{N}x{K}", lambda: ops.gemm_nt(x, w, bias=bias))
        if N <= 4096:
            repeat_equal(f"gemm_nt {M}x{N}x{K} resid", lambda: ops.gemm_nt(x, w, bias=bias, epilogue=ops.EPI_RESID, aux=aux))
            repeat_equal(f"gemm_nt {M}x{N}x{K} relumask", lambda: ops.gemm_nt(x, w, epilogue=ops.EPI_RELUMASK, aux=aux))
        if M == 1024:
            pre = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
            repeat_equal(f"gemm_nt {M}x{N}x{K} gelu", lambda: (ops.gemm_nt(x, w, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre), pre))
            repeat_equal(f"gemm_nt {M}x{N}x{K} gelubwd", lambda: ops.gemm_nt(x, w, epilogue=ops.EPI_GELUBWD, aux=aux))
            repeat_equal(f"gemm_nt {M}x{N}x{K} fp32", lambda: ops.gemm_nt(x, w, bias=bias, out_fp32=True))
        del x, w, aux
    ws = torch.empty(32 * (2304 * 768 + 2304), device=dev)
    for T, I, J in [(603136, 576, 192), (603136, 192, 192), (603136, 2048, 192), (603136, 192, 2048), (278462, 1152, 384), (278462, 2048, 384),
                    (278462, 384, 2048), (125504, 2304, 768), (125504, 768, 768), (125504, 2048, 768), (125504, 768, 2048),
                    (1024, 4096, 256), (1024, 2048, 2048), (1024, 256, 2048), (70001, 192, 2048)]:
        a, b = rnd(T, I).bfloat16(), rnd(T, J).bfloat16()

        def tn():
            c, cs = torch.empty((I, J), device=dev), torch.empty(I, device=dev)
            ops.gemm_tn(a, b, c, colsum=cs, workspace=ws)
            return c, cs
        repeat_equal(f"gemm_tn {T}x{I}x{J}", tn)
        del a, b
    for M, N, K in [(125504, 2304, 768), (125504, 2048, 768), (125504, 768, 2048), (125504, 768, 768), (1000, 768, 768)]:
        x, w, bias = rnd(M, K).bfloat16(), rnd(N, K, sc=1 / math.sqrt(K)).bfloat16(), rnd(N, sc=0.1)
        aux = rnd(M, N).bfloat16()
        xq, xs = ops.mx8_quantize(x)
        wq, wsq = ops.mx8_quantize(w)
        flat = lambda r: (r[0],) + tuple(r[1])
        repeat_equal(f"mx8 {M}x{N}x{K}", lambda: ops.gemm_nt_mx8(xq, xs, wq, wsq, bias=bias))
        repeat_equal(f"mx8 {M}x{N}x{K} relu + q", lambda: flat(ops.gemm_nt_mx8(xq, xs, wq, wsq, bias=bias, epilogue=ops.EPI_RELU, emit_q=True)))
        repeat_equal(f"mx8 {M}x{N}x{K} resid", lambda: ops.gemm_nt_mx8(xq, xs, wq, wsq, bias=bias, epilogue=ops.EPI_RESID, aux=aux))
        repeat_equal(f"mx8 {M}x{N}x{K} relumask + q", lambda: flat(ops.gemm_nt_mx8(xq, xs, wq, wsq, epilogue=ops.EPI_RELUMASK, aux=aux, emit_q=True)))
        del x, w, aux, xq, wq


@pytest.mark.parametrize("B,V,PR", [(5, 3, 1000), (4, 10, 4096), (3, 2, 640)])
def test_dino_loss_over_more_than_two_student_views(B, V, PR):
    """chadavit_dino_loss_multicrop (the standard-DINO multi-crop option) against the reference loss arithmetic restated by the oracle
    (oracle.chada_ref.dino_loss with n_student_views = V; losses/dino.py:69-100 with chunk(V)): loss, dL/dstudent, teacher column sum.
    V = 2 must agree with the two-view kernel bit for bit in the loss."""
    import numpy as np
    from oracle import chada_ref as R
    from chadavit_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    s = torch.randn(V * B, PR) * 1.5
    t = torch.randn(2 * B, PR) * 1.5
    c = torch.randn(1, PR) * 0.1
    sr = s.clone().requires_grad_(True)
    loss_ref = R.dino_loss(sr, t, c, 0.05, 0.1, n_student_views=V)
    loss_ref.backward()
    if V == 2:
        rows2, d2, cs2 = ops.dino_loss(s.to(dev), t.to(dev), c.view(-1).to(dev), 0.1, 0.05)
        assert abs(rows2.mean().item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
        return
    rows, dstu, colsum = ops.dino_loss(s.to(dev), t.to(dev), c.view(-1).to(dev), 0.1, 0.05)
    assert rows.shape == (B,) and dstu.shape == (V * B, PR)
    assert abs(rows.mean().item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item()), (rows.mean().item(), loss_ref.item())
    g = dstu.float().cpu()
    assert float((g - sr.grad).norm() / sr.grad.norm()) <= 5e-3          # bf16-stored gradient
    np.testing.assert_allclose(colsum.cpu().numpy(), t.sum(0).numpy(), rtol=1e-5, atol=1e-4)


