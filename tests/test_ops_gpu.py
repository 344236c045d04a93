"""GPU parity of the backbone building blocks (implicit-GEMM conv family, BatchNorm pieces, pooling,
layout bridges) through the C ABI, against float64 torch-CPU evaluations of the same operators."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def _has_study():
    import awr_amd  # noqa: F401
    from awr_amd import _lib
    return _lib.HAS_STUDY


# measured-and-rejected forms of earlier rounds live in study builds only (AWR_BUILD_STUDY=1 python -m awr_amd.build --force): their tests run there
study_only = pytest.mark.skipif(not _has_study(), reason="study form: needs libawr_hip.so built with -DAWR_STUDY")

_KEEP = []


def DP(L, t, dev):
    """device pointer of a host tensor; the device copy is kept alive until the module is torn down
    (a temporary freed right after data_ptr() would be recycled by the caching allocator)."""
    d = t.to(dev)
    _KEEP.append(d)
    return L.ptr(d)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    import awr_amd  # noqa: F401
    from awr_amd import ops as o
    return o


@pytest.fixture(scope="module")
def L():
    import awr_amd  # noqa: F401
    from awr_amd import _lib
    return _lib


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + 1000 * len(shape) + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def rel_err(a, ref):
    ref = ref.double()
    return float((a.double() - ref).abs().max() / (ref.abs().max() + 1e-30))


# kind, cin, cout, k, stride, pad, B, H
CONV_CASES = [
    ("conv", 64, 64, 3, 1, 1, 2, 32),
    ("conv", 64, 128, 3, 2, 1, 3, 20),      # ragged M (not a tile multiple)
    ("conv", 128, 256, 3, 2, 1, 2, 16),
    ("conv", 512, 512, 3, 1, 1, 2, 8),      # K = 4608
    ("conv", 64, 128, 1, 2, 0, 2, 16),      # downsample 1x1 s2
    ("conv", 256, 64, 1, 1, 0, 2, 16),
    ("conv", 128, 128, 3, 1, 1, 5, 4),      # tiny spatial (hourglass bottom)
    ("deconv", 128, 64, 4, 2, 1, 2, 8),
    ("deconv", 512, 256, 4, 2, 1, 1, 8),
]


def _torch_fwd(kind, x, w, b, stride, pad):
    return (TF.conv2d if kind == "conv" else TF.conv_transpose2d)(x, w, b, stride, pad)


@pytest.mark.parametrize("kind,cin,cout,k,stride,pad,B,H", CONV_CASES)
def test_conv_forward_dgrad_wgrad(ops, dev, kind, cin, cout, k, stride, pad, B, H):
    spec = ops.ConvSpec(kind, cin, cout, k, stride, pad)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w = rnd(*wshape, seed=1, scale=(cin * k * k) ** -0.5)
    x = rnd(B, cin, H, H, seed=2)
    bias = rnd(cout, seed=3)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y_ref = _torch_fwd(kind, xd, wd, bias.double(), stride, pad)
    gy = rnd(*y_ref.shape, seed=4)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [xd, wd], gy.double())

    wg = w.to(dev)
    wp = ops.pack_weight(wg, spec.fwd_pack())
    y = ops.conv_forward(spec, ops.nhwc(x).to(dev), wp, bias=bias.to(dev))
    # fp32 MFMA == k-ordered fmaf chain: rounding error grows ~ sqrt(K) * 2^-24 relative to the row scale
    tol = 5e-6
    assert rel_err(ops.nchw(y).cpu(), y_ref.detach()) < tol
    wpd = ops.pack_weight(wg, spec.dgrad_pack())
    gx = ops.conv_dgrad(spec, ops.nhwc(gy).to(dev), wpd, H, H)
    assert rel_err(ops.nchw(gx).cpu(), gx_ref) < tol
    bg = torch.empty(cout, device=dev) if kind == "conv" else None
    gw = ops.conv_wgrad(spec, ops.nhwc(x).to(dev), ops.nhwc(gy).to(dev), bias_grad=bg)
    assert rel_err(gw.cpu(), gw_ref) < 1e-5
    if bg is not None:       # the bias gradient falls out of the same kernel (column sums of dY)
        assert rel_err(bg.cpu(), gy.double().sum((0, 2, 3))) < 1e-5
    # accumulate paths: dgrad onto an existing gradient, wgrad onto an existing gradient
    base = rnd(B, cin, H, H, seed=5)
    acc = ops.nhwc(base).to(dev)
    ops.conv_dgrad(spec, ops.nhwc(gy).to(dev), wpd, H, H, out=acc, res=acc)
    assert rel_err(ops.nchw(acc).cpu(), gx_ref + base.double()) < tol
    gw2 = ops.conv_wgrad(spec, ops.nhwc(x).to(dev), ops.nhwc(gy).to(dev), grad=gw.clone(), accumulate=True)
    assert rel_err(gw2.cpu(), 2 * gw_ref) < 1e-5


@pytest.mark.parametrize("products,staging", [(1, 2), (1, 0), (6, 2)])
@pytest.mark.parametrize("tm,tn", [(1, 1), (1, 2), (2, 1), (2, 2)])
@pytest.mark.parametrize("kind,cin,cout,k,stride,pad,B,H", [("conv", 64, 96, 3, 1, 1, 3, 18), ("conv", 128, 160, 3, 2, 1, 2, 20),
                                                              ("deconv", 64, 96, 4, 2, 1, 3, 10), ("conv", 32, 64, 1, 1, 0, 2, 12)])
def test_conv_every_tile_shape(ops, L, dev, products, staging, tm, tn, kind, cin, cout, k, stride, pad, B, H):
    """All four workgroup tiles (64/128 x 64/128) in both product modes (f32 MFMA, 6-product bf16 split) and, for the f32 mode, both
    operand staging paths (2 = LDS-DMA, the default; 0 = through registers) on ragged M and N (not multiples of any tile), incl. a
    single-K-slice problem."""
    L.call("awr_debug_force_tile", tm, tn)
    L.call("awr_set_gemm_products", products)
    L.call("awr_set_gemm_staging", staging)
    try:
        test_conv_forward_dgrad_wgrad(ops, dev, kind, cin, cout, k, stride, pad, B, H)
    finally:
        L.call("awr_debug_force_tile", 0, 0)
        L.call("awr_set_gemm_products", 1)
        L.call("awr_set_gemm_staging", 2)


@study_only
def test_split_act_image_is_the_exact_three_way_cut(ops, L, dev):
    """awr_split_act: every element of [relu](x * scale + shift) as three bf16 pieces whose sum is the fp32 value EXACTLY, in the layout of the
    weights' split image (element idx -> shorts (idx / 32) * 96 + idx % 32 + {0, 32, 64})."""
    x = rnd(3, 5, 7, 96, seed=5).to(dev) * 3.0
    sc, sh = (rnd(96, seed=6) + 1.5).to(dev), rnd(96, seed=7).to(dev)
    for affine, relu in ((False, False), (True, True), (True, False)):
        img = ops.split_act(x, sc if affine else None, sh if affine else None, relu)
        ref = x * sc + sh if affine else x
        if relu:
            ref = torch.relu(ref)
        pl = img.view(-1, 3, 32)                                             # [slice][plane][32]
        pieces = (pl.to(torch.int32) << 16).view(torch.float32)              # bf16 bits -> fp32
        h, m, l = pieces[:, 0].reshape(-1), pieces[:, 1].reshape(-1), pieces[:, 2].reshape(-1)
        tot = (h.double() + m.double() + l.double()).float()                 # (three 8-bit pieces: the double sum is exact)
        assert torch.equal(tot.double(), h.double() + m.double() + l.double())
        if affine:      # the kernel's multiply-add is fused (one rounding): compared with the float64 expression, within one ulp of the larger term
            r64 = x.double() * sc.double() + sh.double()
            r64 = torch.relu(r64) if relu else r64
            bound = ((x.double() * sc.double()).abs() + sh.double().abs()).reshape(-1) * 2.0 ** -23
            assert bool(((tot.double() - r64.reshape(-1)).abs() <= bound).all()), (affine, relu)
        else:
            assert torch.equal(tot, ref.reshape(-1)), (affine, relu)
        assert torch.equal(h, (tot.view(torch.int32) & -65536).view(torch.float32))     # h = the top 16 bits of the value (truncation)
        r1 = tot - h
        assert torch.equal(m, (r1.view(torch.int32) & -65536).view(torch.float32)) and torch.equal(l, r1 - m)


@pytest.mark.parametrize("tm,tn", [(1, 1), (1, 2), (2, 1), (2, 2)])
@pytest.mark.parametrize("kind,cin,cout,k,stride,pad,B,H", [("conv", 64, 96, 3, 1, 1, 3, 18), ("conv", 128, 160, 3, 2, 1, 2, 20),
                                                              ("deconv", 64, 96, 4, 2, 1, 3, 10), ("conv", 96, 64, 1, 1, 0, 2, 12)])
@study_only
def test_split_mode_with_precut_activations_is_bit_identical(ops, L, dev, tm, tn, kind, cin, cout, k, stride, pad, B, H):
    """Split-operand mode, both operands by LDS-DMA (awr_conv_args.in_split: the activation image cut ONCE by awr_split_act) against the
    kernel that cuts every staged row itself: same pieces, same six products per 16 k, same order -> the SAME BITS, on ragged M / N, padding
    taps (out-of-range DMA sources land zero pieces), strided and transposed phases; with bias / residual / ReLU / statistics epilogues.
    And a BatchNorm + ReLU input: cut after the affine by the producer vs applied by the consumer's loader."""
    import ctypes as C
    spec = ops.ConvSpec(kind, cin, cout, k, stride, pad)
    x = rnd(B, cin, H, H, seed=1)
    w = rnd(*((cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)), seed=2, scale=0.05)
    prob = spec.fwd_problem(H, H)
    xin = ops.nhwc(x).to(dev)
    if prob["Cin"] != cin:
        xin = torch.nn.functional.pad(xin, (0, prob["Cin"] - cin))
    xin = xin.contiguous()
    L.call("awr_set_gemm_products", 6)
    try:
        wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
        ops.split_packed(wp)
        bias = rnd(prob["N"], seed=3).to(dev)
        res = rnd(B, prob["Hout"], prob["Wout"], prob["N"], seed=4).to(dev)
        sc, sh = (rnd(prob["Cin"], seed=5) + 1.5).to(dev), rnd(prob["Cin"], seed=6).to(dev)
        for form in ("plain", "epilogue", "stats", "bn_input"):
            outs = []
            for pre in (False, True):
                out = torch.full((B, prob["Hout"], prob["Wout"], prob["N"]), float("nan"), device=dev)
                kw = {}
                if form in ("epilogue", "stats"):
                    kw = dict(bias=bias, res=res, relu_out=True)
                st = torch.zeros(16, 2, prob["N"], device=dev, dtype=torch.float64) if form == "stats" else None
                if form == "bn_input":
                    if pre:
                        kw["in_split"] = ops.split_act(xin, sc, sh, True)
                    else:
                        kw.update(in_scale=sc, in_shift=sh, relu_in=True)
                elif pre:
                    kw["in_split"] = ops.split_act(xin)
                a = ops.make_conv_args(prob, B, xin, wp, out, stats=st, T=spec.T, **kw)
                a.tile_m, a.tile_n = tm, tn
                L.call("awr_conv_gemm", C.byref(a), L.stream())
                torch.cuda.synchronize()
                outs.append((out.clone(), None if st is None else st.sum(0).float()))
            (o0, s0), (o1, s1) = outs
            assert not torch.isnan(o1).any(), form
            assert torch.equal(o0, o1), (form, float((o0 - o1).abs().max()))
            if s0 is not None:
                assert rel_err(s1.cpu(), s0.cpu()) < 1e-6
        # and against float64
        ref = TF.conv2d(x.double(), w.double(), None, stride, pad) if kind == "conv" else TF.conv_transpose2d(x.double(), w.double(), None, stride, pad)
        out = torch.empty(B, prob["Hout"], prob["Wout"], prob["N"], device=dev)
        a = ops.make_conv_args(prob, B, xin, wp, out, T=spec.T, in_split=ops.split_act(xin))
        a.tile_m, a.tile_n = tm, tn
        L.call("awr_conv_gemm", C.byref(a), L.stream())
        assert rel_err(ops.nchw(out)[:, :cout].cpu(), ref) < 2e-6
    finally:
        L.call("awr_set_gemm_products", 1)


@pytest.mark.parametrize("tm,tn", [(1, 1), (1, 2), (2, 1), (2, 2)])
def test_lds_dma_staging_is_bit_identical_to_register_staging(ops, L, dev, tm, tn):
    """The LDS-DMA kernel (buffer_load ... lds into swizzled unpadded rows, 16-float stages) walks K in the same order as the register-staged
    one, so every fused form must give the SAME BITS: 3x3 with the fused input affine + ReLU (activation rows through registers, weights by
    DMA) + bias / affine / residual / ReLU / statistics epilogue; a strided 3x3 and a transposed conv (padding taps = out-of-range DMA
    sources that must land zeros); the two-tensor K extent of conv3 + skip_layer (hourglass.py:44-59); the short-K operand-prefetch form;
    a data gradient with the fused BatchNorm-backward reduction, plain and accumulating onto an earlier contribution with the mask from a
    stored activation."""
    import ctypes as C
    outs, keep = {}, []

    def D(t):      # device copy kept alive until the test ends (make_conv_args only stores raw pointers)
        keep.append(t.to(dev).contiguous())
        return keep[-1]
    # 0 = register staging; 2 = LDS-DMA: these small launches take the deep pipeline (four stage buffers, round 5); "2s" = LDS-DMA with the
    # deep pipeline switched off (two stage buffers, what the chip-filling launches run)
    for staging in (0, 2, "2s"):
        L.call("awr_set_gemm_staging", 2 if staging == "2s" else staging)
        L.call("awr_debug_set_knob", b"deep", 0 if staging == "2s" else 1)
        try:
            got = []
            # (a) fused prologue + full epilogue, ragged N
            B, H, cin, cout = 3, 18, 64, 96
            spec = ops.ConvSpec("conv", cin, cout, 3, 1, 1)
            x, w = rnd(B, cin, H, H, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.05)
            s_, t_ = rnd(cin, seed=3) + 1.5, rnd(cin, seed=4)
            so, to, bias, res = rnd(cout, seed=5) + 1.5, rnd(cout, seed=6), rnd(cout, seed=7), rnd(B, cout, H, H, seed=8)
            wp = ops.pack_weight(D(w), spec.fwd_pack())
            stats = torch.zeros(16, 2, cout, device=dev, dtype=torch.float64)
            prob = spec.fwd_problem(H, H)
            out = torch.full((B, H, H, prob["N"]), float("nan"), device=dev)
            a = ops.make_conv_args(prob, B, D(ops.nhwc(x)), wp, out, in_scale=D(s_), in_shift=D(t_), relu_in=True, bias=D(bias),
                                   out_scale=D(so), out_shift=D(to), res=D(ops.nhwc(res)), relu_out=True, stats=stats, T=spec.T)
            a.tile_m, a.tile_n = tm, tn
            L.call("awr_conv_gemm", C.byref(a), L.stream())
            got += [out.clone(), stats.sum(0).float()]
            # (a2) the same without the statistics: the reduction-free epilogue (bias, folded BatchNorm, residual, ReLU)
            out2 = torch.full((B, H, H, prob["N"]), float("nan"), device=dev)
            a2 = ops.make_conv_args(prob, B, D(ops.nhwc(x)), wp, out2, in_scale=D(s_), in_shift=D(t_), relu_in=True, bias=D(bias),
                                    out_scale=D(so), out_shift=D(to), res=D(ops.nhwc(res)), relu_out=True, T=spec.T)
            a2.tile_m, a2.tile_n = tm, tn
            L.call("awr_conv_gemm", C.byref(a2), L.stream())
            got.append(out2.clone())
            # (b) strided conv, transposed conv, plain activations (both operands by DMA), ragged M
            for kind, ci, co, k, st, p, hh in (("conv", 128, 160, 3, 2, 1, 20), ("deconv", 64, 96, 4, 2, 1, 10), ("conv", 96, 64, 1, 1, 0, 12)):
                sp = ops.ConvSpec(kind, ci, co, k, st, p)
                xx = rnd(2, ci, hh, hh, seed=11)
                ww = rnd(*((co, ci, k, k) if kind == "conv" else (ci, co, k, k)), seed=12, scale=0.05)
                pr = sp.fwd_problem(hh, hh)
                xin = ops.nhwc(xx)
                if pr["Cin"] != ci:
                    xin = torch.nn.functional.pad(xin, (0, pr["Cin"] - ci))
                oo = torch.full((2, pr["Hout"], pr["Wout"], pr["N"]), float("nan"), device=dev)
                aa = ops.make_conv_args(pr, 2, D(xin), D(ops.pack_weight(D(ww), sp.fwd_pack())), oo, T=sp.T)
                aa.tile_m, aa.tile_n = tm, tn
                L.call("awr_conv_gemm", C.byref(aa), L.stream())
                got.append(oo.clone())
            # (c) two input tensors (K = [in | in2]) with the affine on the first
            c1, c2, co = 64, 128, 160
            xa, xb = rnd(2, c1, 12, 12, seed=21), rnd(2, c2, 12, 12, seed=22)
            wab = rnd(co, c1 + c2, 1, 1, seed=23, scale=0.1)
            sp = ops.ConvSpec("conv", c1 + c2, co, 1, 1, 0)
            pr = sp.fwd_problem(12, 12)
            oo = torch.full((2, 12, 12, pr["N"]), float("nan"), device=dev)
            st2 = torch.zeros(16, 2, pr["N"], device=dev, dtype=torch.float64)
            sa, ta = rnd(c1, seed=24) + 1.2, rnd(c1, seed=25)
            aa = ops.make_conv_args(pr, 2, D(ops.nhwc(xa)), D(ops.pack_weight(D(wab), sp.fwd_pack())), oo, in_scale=D(sa), in_shift=D(ta),
                                    relu_in=True, stats=st2, T=sp.T)
            xb_d = D(ops.nhwc(xb))
            aa.in2, aa.Cin1, aa.tile_m, aa.tile_n = L.ptr(xb_d), c1, tm, tn
            L.call("awr_conv_gemm", C.byref(aa), L.stream())
            got += [oo.clone(), st2.sum(0).float()]
            ref_c = TF.conv2d(torch.cat([TF.relu(xa.double() * sa.double().view(1, -1, 1, 1) + ta.double().view(1, -1, 1, 1)), xb.double()], 1), wab.double())
            assert rel_err(ops.nchw(oo)[:, :co].cpu(), ref_c) < 2e-6
            # (d) short K + one epilogue operand (prefetch form) and (e) BatchNorm-backward reduction, plain and accumulating with a stored mask
            cin, cout, B, H = 128, 160, 3, 10
            sp = ops.ConvSpec("conv", cin, cout, 1, 1, 0)
            x, w, res = rnd(B, cin, H, H, seed=31), rnd(cout, cin, 1, 1, seed=32, scale=0.1), rnd(B, cout, H, H, seed=33)
            pr = sp.fwd_problem(H, H)
            oo = torch.full((B, H, H, pr["N"]), float("nan"), device=dev)
            aa = ops.make_conv_args(pr, B, D(ops.nhwc(x)), D(ops.pack_weight(D(w), sp.fwd_pack())), oo, res=D(ops.nhwc(res)), T=sp.T)
            aa.tile_m, aa.tile_n = tm, tn
            L.call("awr_conv_gemm", C.byref(aa), L.stream())
            got.append(oo.clone())
            sp3 = ops.ConvSpec("conv", 64, 96, 3, 1, 1)
            gy, y, w3 = rnd(2, 96, 14, 14, seed=41), rnd(2, 64, 14, 14, seed=42), rnd(96, 64, 3, 3, seed=43, scale=0.05)
            dp = sp3.dgrad_problem(14, 14)
            coef4 = torch.zeros(4, dp["N"], device=dev)
            coef4[:, :64] = torch.stack([rnd(64, seed=44) + 1.2, rnd(64, seed=45) * 0.3, rnd(64, seed=46) * 0.2, rnd(64, seed=47) + 1.5]).to(dev)
            yg = torch.zeros(2, 14, 14, dp["N"], device=dev)
            yg[..., :64] = D(ops.nhwc(y))
            gin = ops.nhwc(gy)
            if dp["Cin"] != 96:
                gin = torch.nn.functional.pad(gin, (0, dp["Cin"] - 96))
            for accumulate in (False, True):
                g = torch.full((2, 14, 14, dp["N"]), float("nan"), device=dev)
                sums = torch.zeros(16, 2, dp["N"], device=dev, dtype=torch.float64)
                d = ops.make_conv_args(dp, 2, D(gin), D(ops.pack_weight(D(w3), sp3.dgrad_pack())), g, stats=sums, T=sp3.T)
                d.bnr_y, d.bnr_coef, d.tile_m, d.tile_n = L.ptr(yg), L.ptr(coef4), tm, tn
                if accumulate:
                    act = D(ops.nhwc(rnd(2, dp["N"], 14, 14, seed=48)))
                    g.copy_(ops.nhwc(rnd(2, dp["N"], 14, 14, seed=49)).to(dev))
                    d.bnr_act, d.res = L.ptr(act), L.ptr(g)
                L.call("awr_conv_gemm", C.byref(d), L.stream())
                torch.cuda.synchronize()
                got += [g.clone(), sums.sum(0).float()]
            outs[staging] = got
        finally:
            L.call("awr_set_gemm_staging", 2)
            L.call("awr_debug_set_knob", b"deep", 1)
    for other in (2, "2s"):
        assert len(outs[0]) == len(outs[other])
        for i, (p, q) in enumerate(zip(outs[0], outs[other])):
            if p.dtype == torch.float32 and p.dim() == 2:        # statistics: fp64 atomics in launch order, compared after rounding to fp32
                assert rel_err(q.cpu(), p.cpu()) < 1e-6, (other, i)
            else:
                assert torch.equal(torch.nan_to_num(p, nan=-1234.0), torch.nan_to_num(q, nan=-1234.0)), (other, i)


@pytest.mark.parametrize("accum", [0, 1])
@pytest.mark.parametrize("tm,tn", [(1, 1), (1, 2), (2, 1), (2, 2)])
@pytest.mark.parametrize("kind,cin,cout,k,stride,pad,B,H", [("conv", 64, 96, 3, 1, 1, 3, 18), ("conv", 128, 160, 3, 2, 1, 2, 20),
                                                              ("deconv", 64, 96, 4, 2, 1, 3, 10), ("conv", 96, 64, 1, 1, 0, 2, 12)])
def test_dgrad_with_unmaterialised_batchnorm_backward(ops, L, dev, accum, tm, tn, kind, cin, cout, k, stride, pad, B, H):
    """awr_conv_args.in_bnb_y / in_bnb_coef: the data gradient of a conv whose output y feeds a BatchNorm takes g = d(loss)/d(bn(y)) and y
    and forms d(y) = a1 g + a2 (y - mean) + a3 per channel on the operand's way to the matrix pipe (padding taps stay zero), instead of
    reading a d(y) that a separate pass wrote.  Against float64 (conv_transpose / conv of the float64 d(y)), and against the same GEMM fed
    with d(y) materialised by awr_bn_bwd_apply_only from the coefficients awr_bn_bwd_finalize_lin derives from reduction sums.
    accum = 1: the blocked-accumulation instantiation of the same launches."""
    import ctypes as C
    spec = ops.ConvSpec(kind, cin, cout, k, stride, pad)
    ho = spec.out_hw(H, H)[0]
    w = rnd(*((cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)), seed=2, scale=0.05)
    g, y = rnd(B, cout, ho, ho, seed=3), rnd(B, cout, ho, ho, seed=4) * 2.0 + 0.7
    gamma = rnd(cout, seed=5) + 1.5
    # reduction sums as the fused epilogue leaves them: sum g and sum g * xhat per channel (one slot)
    mean = y.double().mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(y.double().var((0, 2, 3), unbiased=False) + 1e-5)
    xhat = (y.double() - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    npix = B * ho * ho
    dp = spec.dgrad_problem(H, H)
    Cp = dp["Cin"]
    sums = torch.zeros(16, 2, Cp, dtype=torch.float64)
    sums[0, 0, :cout] = g.double().sum((0, 2, 3))
    sums[0, 1, :cout] = (g.double() * xhat).sum((0, 2, 3))
    dy_ref = (gamma.double() * invstd).view(1, -1, 1, 1) * (g.double() - (sums[0, 0, :cout] / npix).view(1, -1, 1, 1) - xhat * (sums[0, 1, :cout] / npix).view(1, -1, 1, 1))
    gx_ref = (TF.conv_transpose2d(dy_ref, w.double(), None, stride, pad, output_padding=(H + 2 * pad - k) % stride) if kind == "conv"
              else TF.conv2d(dy_ref, w.double(), None, stride, pad))

    def padc(t, fill=0.0):      # NHWC with the channel padding of the GEMM's K extent
        t = ops.nhwc(t)
        return (torch.nn.functional.pad(t, (0, Cp - cout), value=fill) if Cp != cout else t).contiguous().to(dev)
    g_d, y_d = padc(g), padc(y)
    vec = lambda v, fill: torch.nn.functional.pad(v.float(), (0, Cp - cout), value=fill).to(dev)
    mean_d, invstd_d, gamma_d = vec(mean, 0.0), vec(invstd, 1.0), vec(gamma, 1.0)
    sums_d, coef, lin4 = sums.to(dev), torch.zeros(3, Cp, device=dev), torch.zeros(4, Cp, device=dev)
    dgam, dbet = torch.zeros(Cp, device=dev), torch.zeros(Cp, device=dev)
    L.call("awr_bn_bwd_finalize_lin", L.ptr(sums_d), Cp, npix, L.ptr(gamma_d), L.ptr(mean_d), L.ptr(invstd_d), L.ptr(coef), L.ptr(lin4), L.ptr(dgam), L.ptr(dbet), 0, 16, L.stream())
    assert rel_err(dgam[:cout].cpu(), (g.double() * xhat).sum((0, 2, 3))) < 1e-5 and rel_err(dbet[:cout].cpu(), g.double().sum((0, 2, 3))) < 1e-5
    assert float(sums_d.abs().max()) == 0.0          # the accumulator is re-armed
    dy_mat = torch.empty_like(g_d)
    L.call("awr_bn_bwd_apply_only", L.ptr(g_d), None, L.ptr(y_d), L.ptr(mean_d), L.ptr(invstd_d), None, None, L.ptr(coef), npix, Cp, L.ptr(dy_mat), None, None, L.stream())
    assert rel_err(ops.nchw(dy_mat)[:, :cout].cpu(), dy_ref) < 2e-6
    wd = ops.pack_weight(w.to(dev), spec.dgrad_pack())
    outs = []
    for lazy in (False, True):
        out = torch.full((B, H, H, dp["N"]), float("nan"), device=dev)
        if not dp["full"]:
            out.zero_()
        a = ops.make_conv_args(dp, B, g_d if lazy else dy_mat, wd, out, res=None if dp["full"] else out, T=spec.T)
        a.tile_m, a.tile_n, a.accum = tm, tn, accum
        if lazy:
            a.in_bnb_y, a.in_bnb_coef = L.ptr(y_d), L.ptr(lin4)
        L.call("awr_conv_gemm", C.byref(a), L.stream())
        torch.cuda.synchronize()
        outs.append(ops.nchw(out)[:, :cin].cpu())
        assert rel_err(outs[-1], gx_ref) < 3e-6, lazy
    assert rel_err(outs[1], outs[0]) < 2e-6


def test_conv_fused_prologue_epilogue_stats(ops, dev):
    """conv( relu(x*s+t) ) * so + to + res -> relu, with per-channel statistics of the pre-ReLU value."""
    B, H, cin, cout = 2, 16, 64, 96           # N = 96: not a multiple of the 64/128 tile
    spec = ops.ConvSpec("conv", cin, cout, 3, 1, 1)
    x, w = rnd(B, cin, H, H, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.05)
    s, t = rnd(cin, seed=3) + 1.5, rnd(cin, seed=4)
    so, to, bias = rnd(cout, seed=5) + 1.5, rnd(cout, seed=6), rnd(cout, seed=7)
    res = rnd(B, cout, H, H, seed=8)
    a = TF.relu(x.double() * s.double().view(1, -1, 1, 1) + t.double().view(1, -1, 1, 1))
    pre = (TF.conv2d(a, w.double(), bias.double(), 1, 1)) * so.double().view(1, -1, 1, 1) + to.double().view(1, -1, 1, 1) + res.double()
    ref = TF.relu(pre)
    stats = torch.zeros(16, 2, cout, device=dev, dtype=torch.float64)
    wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
    y = ops.conv_forward(spec, ops.nhwc(x).to(dev), wp, in_scale=s.to(dev), in_shift=t.to(dev), relu_in=True, bias=bias.to(dev),
                         out_scale=so.to(dev), out_shift=to.to(dev), res=ops.nhwc(res).to(dev), relu_out=True, stats=stats)
    assert rel_err(ops.nchw(y).cpu(), ref) < 2e-6
    assert rel_err(stats.sum(0)[0].cpu(), pre.sum((0, 2, 3))) < 1e-5          # [AWR_STAT_SLOTS][2][N]
    assert rel_err(stats.sum(0)[1].cpu(), (pre * pre).sum((0, 2, 3))) < 1e-5


def test_stem_im2col_gemm(ops, L, dev):
    """5x5 stem (Cin=1) as im2col + the same MFMA GEMM; forward and weight gradient."""
    B, H = 2, 32
    img, w = rnd(B, 1, H, H, seed=1), rnd(64, 1, 5, 5, seed=2, scale=0.2)
    wd = w.double().requires_grad_(True)
    ref = TF.conv2d(img.double(), wd, None, 1, 2)
    gy = rnd(*ref.shape, seed=3)
    (gw_ref,) = torch.autograd.grad(ref, wd, gy.double())
    cols = torch.empty(B, H, H, 32, device=dev)
    L.call("awr_stem_im2col", DP(L, img, dev), B, H, H, L.ptr(cols), L.stream())
    spec = ops.ConvSpec("conv", 25, 64, 1, 1, 0, cin_pad=32)
    wp = ops.pack_weight(w.to(dev).view(64, 25, 1, 1), spec.fwd_pack())
    y = ops.conv_forward(spec, cols, wp)
    assert rel_err(ops.nchw(y).cpu(), ref.detach()) < 2e-6
    gw = ops.conv_wgrad(spec, cols, ops.nhwc(gy).to(dev))
    assert rel_err(gw.cpu().view(64, 1, 5, 5), gw_ref) < 5e-6


def test_batch_statistics_of_nearly_constant_channels(ops, L, dev):
    """sum x / sum x^2 of channels whose spread is tiny next to their mean (dead or saturated channels, a constant image
    background): a plain one-pass fp32 accumulation loses the variance (|mean| = 5, std = 1e-3: x^2 carries the spread in its last
    two bits).  All three producers -- GEMM epilogue, stand-alone channel statistics, fused stem -- accumulate shifted sums and must
    give the float64 variance to 1e-4 relative."""
    B, H, cin, cout = 2, 16, 64, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * cin ** -0.5
    w[:16] *= 1e-3                                     # channels 0..15: conv output = bias + 1e-3 * noise
    bias = torch.randn(cout, generator=g)
    bias[:16] = torch.linspace(-8, 8, 16)
    spec = ops.ConvSpec("conv", cin, cout, 1, 1, 0)
    wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
    stats = torch.zeros(16, 2, cout, device=dev, dtype=torch.float64)
    y = ops.conv_forward(spec, ops.nhwc(x).to(dev), wp, bias=bias.to(dev), stats=stats)
    n = B * H * H

    def var_of(st):
        st = st.sum(0).cpu()
        return st[0] / n, st[1] / n - (st[0] / n) ** 2

    yd = ops.nchw(y).cpu().double()                    # statistics of the values the kernel stored (fp32), evaluated in float64
    mean_ref, var_ref = yd.mean((0, 2, 3)), yd.var((0, 2, 3), unbiased=False)
    assert float(var_ref[:16].max()) < 1e-5 and float(var_ref[16:].min()) > 0.1
    mean, var = var_of(stats)
    assert float(((mean - mean_ref).abs() / (mean_ref.abs() + 1e-3)).max()) < 1e-6
    assert float(((var - var_ref).abs() / var_ref).max()) < 1e-4, ((var - var_ref).abs() / var_ref)[:16]
    stats2 = torch.zeros(16, 2, cout, device=dev, dtype=torch.float64)
    L.call("awr_channel_stats", L.ptr(y), n, cout, L.ptr(stats2), 0, L.stream())
    mean2, var2 = var_of(stats2)
    assert float(((var2 - var_ref).abs() / var_ref).max()) < 1e-4
    # fused stem: a constant image -> every interior conv pixel of a channel has the same value (bias + sum of taps)
    Hs = 64
    img = torch.full((B, 1, Hs, Hs), 1.0)
    img[:, :, 20:30, 20:30] += torch.randn(B, 1, 10, 10, generator=g) * 1e-3
    ws = torch.randn(64, 1, 5, 5, generator=g) * 0.2
    bs = torch.randn(64, generator=g) * 3
    ys = TF.conv2d(img.double(), ws.double(), bs.double(), 1, 2)
    stats3 = torch.zeros(16, 2, 64, device=dev, dtype=torch.float64)
    imgd, wsd, bsd = img.to(dev), ws.to(dev).contiguous(), bs.to(dev)      # (named: the launch is asynchronous, temporaries would be recycled)
    L.call("awr_stem_stats", L.ptr(imgd), L.ptr(wsd), L.ptr(bsd), B, Hs, Hs, L.ptr(stats3), 0, L.stream())
    st = stats3.sum(0).cpu()
    ns = B * Hs * Hs
    var3 = st[1] / ns - (st[0] / ns) ** 2
    assert float(((var3 - ys.var((0, 2, 3), unbiased=False)).abs() / ys.var((0, 2, 3), unbiased=False)).max()) < 1e-4


@pytest.mark.parametrize("B,H,C", [(2, 16, 64), (3, 10, 96), (2, 8, 512), (4, 32, 128), (2, 4, 2048)])      # 2048: the Bottleneck ResNets' layer4 (channel chunks)
def test_batchnorm_train_forward_backward(L, dev, B, H, C):
    x = rnd(B, C, H, H, seed=1) * 2 + 0.3
    gamma, beta = rnd(C, seed=2) + 1.5, rnd(C, seed=3)
    res = rnd(B, C, H, H, seed=4)
    rm, rv = rnd(C, seed=5), rnd(C, seed=6) + 1.5
    xd, gd, bd = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    y_ref = TF.relu(TF.batch_norm(xd, rm_ref, rv_ref, gd, bd, True, 0.1, 1e-5) + res.double())
    gout = rnd(B, C, H, H, seed=7)
    gx_ref, gg_ref, gb_ref = torch.autograd.grad(y_ref, [xd, gd, bd], gout.double())

    from awr_amd import ops
    npix = B * H * H
    xg = ops.nhwc(x).to(dev)
    stats = torch.zeros(16, 2, C, device=dev, dtype=torch.float64)
    L.call("awr_channel_stats", L.ptr(xg), npix, C, L.ptr(stats), 0, L.stream())
    scale, shift, mean, invstd = (torch.empty(C, device=dev) for _ in range(4))
    rmg, rvg = rm.to(dev), rv.to(dev)
    L.call("awr_bn_finalize", L.ptr(stats), C, npix, DP(L, gamma, dev), DP(L, beta, dev), L.ptr(rmg), L.ptr(rvg), 0.1, 1e-5,
           L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), 0, L.stream())
    assert float(stats.abs().max()) == 0.0          # finalize re-arms the accumulator
    out = torch.empty_like(xg)
    resg = ops.nhwc(res).to(dev)
    L.call("awr_bn_apply", L.ptr(xg), L.ptr(scale), L.ptr(shift), L.ptr(resg), 1, L.ptr(out), npix, C, L.stream())
    assert rel_err(ops.nchw(out).cpu(), y_ref.detach()) < 3e-6
    assert rel_err(rmg.cpu(), rm_ref) < 1e-6 and rel_err(rvg.cpu(), rv_ref) < 1e-6
    # backward
    sums = torch.zeros(16, 2, C, device=dev, dtype=torch.float64)
    coef = torch.empty(3, C, device=dev)
    goutg = ops.nhwc(gout).to(dev)
    L.call("awr_bn_bwd_reduce", L.ptr(goutg), L.ptr(out), L.ptr(xg), L.ptr(mean), L.ptr(invstd), None, None, npix, C, L.ptr(sums), 0, L.stream())
    dy, g = torch.empty_like(xg), torch.empty_like(xg)
    dgam, dbet = torch.empty(C, device=dev), torch.empty(C, device=dev)
    L.call("awr_bn_bwd_apply", L.ptr(goutg), L.ptr(out), L.ptr(xg), L.ptr(mean), L.ptr(invstd), DP(L, gamma, dev), None, None, L.ptr(sums), L.ptr(coef), npix, C,
           L.ptr(dy), None, L.ptr(g), L.ptr(dgam), L.ptr(dbet), 0, 0, L.stream())
    assert rel_err(ops.nchw(dy).cpu(), gx_ref) < 2e-5
    assert rel_err(dgam.cpu(), gg_ref) < 2e-5 and rel_err(dbet.cpu(), gb_ref) < 2e-5
    assert rel_err(ops.nchw(g).cpu(), gout.double() * (y_ref.detach() > 0)) < 1e-6
    assert float(sums.abs().max()) == 0.0
    # no-residual variant: the ReLU mask is re-derived from y with the forward scale/shift instead of reading the activation
    out2 = torch.empty_like(xg)
    L.call("awr_bn_apply", L.ptr(xg), L.ptr(scale), L.ptr(shift), None, 1, L.ptr(out2), npix, C, L.stream())
    sa, sb = torch.zeros(16, 2, C, device=dev, dtype=torch.float64), torch.zeros(16, 2, C, device=dev, dtype=torch.float64)
    L.call("awr_bn_bwd_reduce", L.ptr(goutg), L.ptr(out2), L.ptr(xg), L.ptr(mean), L.ptr(invstd), None, None, npix, C, L.ptr(sa), 0, L.stream())
    L.call("awr_bn_bwd_reduce", L.ptr(goutg), None, L.ptr(xg), L.ptr(mean), L.ptr(invstd), L.ptr(scale), L.ptr(shift), npix, C, L.ptr(sb), 0, L.stream())
    assert rel_err(sb.sum(0).cpu(), sa.sum(0).cpu()) < 1e-9
    dya, dyb = torch.empty_like(xg), torch.empty_like(xg)
    gam = DP(L, gamma, dev)
    L.call("awr_bn_bwd_apply", L.ptr(goutg), L.ptr(out2), L.ptr(xg), L.ptr(mean), L.ptr(invstd), gam, None, None, L.ptr(sa), L.ptr(coef), npix, C,
           L.ptr(dya), None, None, L.ptr(dgam), L.ptr(dbet), 0, 0, L.stream())
    L.call("awr_bn_bwd_apply", L.ptr(goutg), None, L.ptr(xg), L.ptr(mean), L.ptr(invstd), gam, L.ptr(scale), L.ptr(shift), L.ptr(sb), L.ptr(coef), npix, C,
           L.ptr(dyb), None, None, L.ptr(dgam), L.ptr(dbet), 0, 0, L.stream())
    assert torch.equal(dya, dyb)


def test_bn_fold_eval(L, dev):
    C = 64
    g, b, m, v = rnd(C, seed=1) + 1.5, rnd(C, seed=2), rnd(C, seed=3), rnd(C, seed=4) + 1.5
    sc, sh = torch.empty(C, device=dev), torch.empty(C, device=dev)
    L.call("awr_bn_fold_eval", C, DP(L, g, dev), DP(L, b, dev), DP(L, m, dev), DP(L, v, dev), 1e-5, L.ptr(sc), L.ptr(sh), L.stream())
    x = rnd(2, C, 4, 4, seed=5)
    ref = TF.batch_norm(x.double(), m.double(), v.double(), g.double(), b.double(), False, 0.1, 1e-5)
    got = x.to(dev) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    assert rel_err(got.cpu(), ref) < 2e-6


@pytest.mark.parametrize("k,s,p,B,H,C", [(3, 2, 1, 2, 16, 64), (2, 2, 0, 3, 8, 128), (3, 2, 1, 1, 128, 64)])
def test_maxpool(L, dev, k, s, p, B, H, C):
    from awr_amd import ops
    x = rnd(B, C, H, H, seed=1)
    x[:, :, ::3, ::3] = 0.0
    x = TF.relu(x)            # many exact ties at 0, like the post-ReLU stem output
    xd = x.double().requires_grad_(True)
    ref = TF.max_pool2d(xd, k, s, p)
    gout = rnd(*ref.shape, seed=2)
    (gx_ref,) = torch.autograd.grad(ref, xd, gout.double())
    Ho = ref.shape[-1]
    xg = ops.nhwc(x).to(dev)
    out = torch.empty(B, Ho, Ho, C, device=dev)
    arg = torch.empty(B, Ho, Ho, C, device=dev, dtype=torch.uint8)
    L.call("awr_maxpool_fwd", L.ptr(xg), None, None, 0, B, H, H, C, k, s, p, L.ptr(out), L.ptr(arg), L.stream())
    assert rel_err(ops.nchw(out).cpu(), ref.detach()) == 0.0
    dx = torch.empty_like(xg)
    L.call("awr_maxpool_bwd", DP(L, ops.nhwc(gout), dev), L.ptr(arg), B, H, H, C, k, s, p, L.ptr(dx), 0, L.stream())
    got = ops.nchw(dx).cpu().double()
    # ties: ATen and this kernel both route to the first maximum in scan order
    assert rel_err(got, gx_ref) < 1e-6
    L.call("awr_maxpool_bwd", DP(L, ops.nhwc(gout), dev), L.ptr(arg), B, H, H, C, k, s, p, L.ptr(dx), 1, L.stream())
    assert rel_err(ops.nchw(dx).cpu(), 2 * gx_ref) < 1e-6


def test_fused_affine_loaders(ops, L, dev):
    """Un-materialised BatchNorm+ReLU inputs: wgrad (conv: gathered operand, deconv: dense operand) and max-pool
    apply relu(x*s+t) while loading and must equal the same op on the materialised tensor."""
    B, H, cin, cout = 2, 16, 64, 96
    x = rnd(B, cin, H, H, seed=1)
    s_, t_ = rnd(cin, seed=2) + 0.2, rnd(cin, seed=3) * 0.5          # some negative scales too
    a = TF.relu(x * s_.view(1, -1, 1, 1) + t_.view(1, -1, 1, 1))
    xg, ag, sg, tg = ops.nhwc(x).to(dev), ops.nhwc(a).to(dev), s_.to(dev), t_.to(dev)
    for kind, k, st, p in (("conv", 3, 1, 1), ("conv", 3, 2, 1), ("deconv", 4, 2, 1)):
        spec = ops.ConvSpec(kind, cin, cout, k, st, p)
        ho = spec.out_hw(H, H)[0]
        gy = ops.nhwc(rnd(B, cout, ho, ho, seed=4)).to(dev)
        ref = ops.conv_wgrad(spec, ag, gy)
        got = ops.conv_wgrad(spec, xg, gy, x_affine=(sg, tg, True))
        assert rel_err(got.cpu(), ref.cpu()) < 2e-5, kind
    out_a, out_x = torch.empty(B, 8, 8, cin, device=dev), torch.empty(B, 8, 8, cin, device=dev)
    arg_a, arg_x = torch.empty(B, 8, 8, cin, device=dev, dtype=torch.uint8), torch.empty(B, 8, 8, cin, device=dev, dtype=torch.uint8)
    L.call("awr_maxpool_fwd", L.ptr(ag), None, None, 0, B, H, H, cin, 3, 2, 1, L.ptr(out_a), L.ptr(arg_a), L.stream())
    L.call("awr_maxpool_fwd", L.ptr(xg), L.ptr(sg), L.ptr(tg), 1, B, H, H, cin, 3, 2, 1, L.ptr(out_x), L.ptr(arg_x), L.stream())
    assert rel_err(out_x.cpu(), out_a.cpu()) < 1e-6 and float((arg_a != arg_x).float().mean()) < 1e-3


def test_upsample_add_and_misc(L, dev):
    from awr_amd import ops
    B, Hl, C = 2, 8, 64
    up1, low = rnd(B, C, 2 * Hl, 2 * Hl, seed=1), rnd(B, C, Hl, Hl, seed=2)
    ref = up1 + TF.interpolate(low, scale_factor=2, mode="nearest")
    out = torch.empty(B, 2 * Hl, 2 * Hl, C, device=dev)
    L.call("awr_upsample2_add", DP(L, ops.nhwc(up1), dev), DP(L, ops.nhwc(low), dev), B, Hl, Hl, C, L.ptr(out), L.stream())
    assert rel_err(ops.nchw(out).cpu(), ref) < 1e-7
    gout = rnd(B, C, 2 * Hl, 2 * Hl, seed=3)
    dlow = torch.empty(B, Hl, Hl, C, device=dev)
    L.call("awr_upsample2_bwd", DP(L, ops.nhwc(gout), dev), B, Hl, Hl, C, L.ptr(dlow), 0, L.stream())
    ref_d = gout.view(B, C, Hl, 2, Hl, 2).sum((3, 5))
    assert rel_err(ops.nchw(dlow).cpu(), ref_d) < 1e-6
    # relu_bwd / add / bias_grad
    a, d = rnd(B, Hl, Hl, C, seed=4), rnd(B, Hl, Hl, C, seed=5)
    g = torch.empty(B, Hl, Hl, C, device=dev)
    L.call("awr_relu_bwd", DP(L, d, dev), DP(L, a, dev), L.ptr(g), a.numel(), L.stream())
    assert rel_err(g.cpu(), d * (a > 0)) == 0.0
    L.call("awr_add", DP(L, d, dev), DP(L, a, dev), L.ptr(g), a.numel(), L.stream())
    assert rel_err(g.cpu(), d + a) == 0.0
    db = torch.empty(C, device=dev)
    L.call("awr_bias_grad", DP(L, d, dev), B * Hl * Hl, C, L.ptr(db), 0, L.stream())
    assert rel_err(db.cpu(), d.double().sum((0, 1, 2))) < 1e-5


@pytest.mark.parametrize("B,P,C,Cp", [(2, 4096, 56, 64), (1, 16384, 84, 96), (3, 100, 56, 64)])
def test_layout_bridges(L, dev, B, P, C, Cp):
    x = rnd(B, P, Cp, seed=1)
    out = torch.empty(B, C, P, device=dev)
    L.call("awr_nhwc_to_nchw", DP(L, x, dev), B, P, Cp, C, L.ptr(out), L.stream())
    assert torch.equal(out.cpu(), x[:, :, :C].permute(0, 2, 1).contiguous())
    back = torch.full((B, P, Cp), 7.0, device=dev)
    L.call("awr_nchw_to_nhwc", L.ptr(out), B, P, Cp, C, L.ptr(back), L.stream())
    exp = x.clone()
    exp[:, :, C:] = 0.0
    assert torch.equal(back.cpu(), exp)


def test_pack_unpack_roundtrip(ops, L, dev):
    w = rnd(96, 64, 3, 3, seed=1)
    spec = ops.ConvSpec("conv", 64, 96, 3, 1, 1)
    wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
    assert wp.shape == (128, 9, 64)
    assert torch.equal(wp[:96].cpu(), w.view(96, 64, 9).permute(0, 2, 1).contiguous())
    assert float(wp[96:].abs().max()) == 0.0
    g = torch.empty(96, 64, 3, 3, device=dev)
    L.call("awr_unpack_wgrad", L.ptr(wp), 96, 64, 9, 64, L.ptr(g), 0, L.stream())
    assert torch.equal(g.cpu(), w)
    wpt = ops.pack_weight(w.to(dev), spec.dgrad_pack())
    assert torch.equal(wpt[:64, :, :96].cpu(), w.view(96, 64, 9).permute(1, 2, 0).contiguous())


def test_batch_chunking_above_4gb(ops, dev):
    """Tensors >= 4 GiB (Hourglass 128-channel maps at 256x256, batch 128: BASELINE config 5) exceed the kernels' 32-bit
    buffer offsets; the entry points split the batch.  Property: the chunked call equals per-half calls bit for bit,
    for the forward GEMM (4 GiB input) and, to rounding, for the weight gradient (4 GiB gathered operand)."""
    B, H, cin, cout = 128, 256, 128, 64
    spec = ops.ConvSpec("conv", cin, cout, 1, 1, 0)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand(B, H, H, cin, device=dev, generator=g) - 0.5
    assert x.numel() * 4 >= (1 << 32)
    w = (torch.rand(cout, cin, 1, 1, device=dev, generator=g) - 0.5) * 0.2
    wp = ops.pack_weight(w, spec.fwd_pack())
    y = ops.conv_forward(spec, x, wp)
    for lo, hi in ((0, 64), (64, 128)):
        assert torch.equal(y[lo:hi], ops.conv_forward(spec, x[lo:hi].contiguous(), wp))
    ref = torch.einsum("bhwc,oc->bhwo", x[100, 17:19].unsqueeze(0).double(), w.view(cout, cin).double())
    assert rel_err(y[100, 17:19].unsqueeze(0).cpu(), ref.cpu()) < 2e-6
    gw = ops.conv_wgrad(spec, x, y)
    gw2 = ops.conv_wgrad(spec, x[:64].contiguous(), y[:64].contiguous()) + ops.conv_wgrad(spec, x[64:].contiguous(), y[64:].contiguous())
    assert rel_err(gw.cpu(), gw2.cpu()) < 1e-4


def test_split_mode_accuracy_matches_fp32_mfma(ops, L, dev):
    """The 6-product split mode (awr_set_gemm_products(6)) against the FP32-MFMA mode, both measured against an fp64
    reference on a long contraction (K = 9*256 forward / dgrad, K = 2*24*24 pixels for the weight gradient) with wide
    dynamic range.  Dropping the m*l, l*m, l*l partial products costs <= 2^-23 per product -- below the rounding of the
    fp32 accumulation both modes share -- so the two errors must be of the same size."""
    spec = ops.ConvSpec("conv", 256, 256, 3, 1, 1)
    B, H = 2, 24
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 256, H, H, generator=g) * torch.exp(2.0 * torch.randn(B, 256, H, H, generator=g))     # heavy-tailed
    w = torch.randn(256, 256, 3, 3, generator=g) * (256 * 9) ** -0.5
    gy = torch.randn(B, 256, H, H, generator=g) * torch.exp(1.5 * torch.randn(B, 256, H, H, generator=g))
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y_ref = TF.conv2d(xd, wd, None, 1, 1)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [xd, wd], gy.double())
    errs = {}
    try:
        for n in (1, 6):
            L.call("awr_set_gemm_products", n)
            wg = w.to(dev)
            y = ops.conv_forward(spec, ops.nhwc(x).to(dev), ops.pack_weight(wg, spec.fwd_pack()))
            gx = ops.conv_dgrad(spec, ops.nhwc(gy).to(dev), ops.pack_weight(wg, spec.dgrad_pack()), H, H)
            gw = ops.conv_wgrad(spec, ops.nhwc(x).to(dev), ops.nhwc(gy).to(dev))
            errs[n] = (rel_err(ops.nchw(y).cpu(), y_ref.detach()), rel_err(ops.nchw(gx).cpu(), gx_ref), rel_err(gw.cpu(), gw_ref))
    finally:
        L.call("awr_set_gemm_products", 1)
    print("rel. error vs fp64 (fwd, dgrad, wgrad): f32 MFMA %s | 6-product split %s" % (errs[1], errs[6]))
    for e1, e6 in zip(errs[1], errs[6]):
        assert e6 <= 2.0 * e1 + 1e-7, (errs[1], errs[6])
        assert e6 < 5e-6


@pytest.mark.parametrize("B,H,W", [(2, 32, 48), (3, 128, 128), (1, 16, 16)])
def test_fused_stem_matches_conv_bn_relu_maxpool(L, dev, B, H, W):
    """csrc/awr_stem.hip: conv 5x5 (1 -> 64, pad 2, no bias) -> BatchNorm (batch statistics) -> ReLU -> MaxPool(3,2,1) without
    ever writing the full-resolution map, and its backward (BN parameter gradients, conv weight gradient) by recomputation --
    against float64 torch autograd of the four separate operators (resnet_deconv.py:31-36, :118-121)."""
    from awr_amd import ops
    g = torch.Generator().manual_seed(5 + H)
    img = (torch.rand(B, 1, H, W, generator=g) * 2 - 1)
    img[:, :, : H // 3] = 1.0                                    # constant background like the real crops
    w = torch.randn(64, 1, 5, 5, generator=g) * 0.2
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    # ---- float64 reference ----
    wd, gd, bd = w.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = TF.conv2d(img.double(), wd, None, 1, 2)
    a = TF.relu(TF.batch_norm(y, None, None, gd, bd, True, 0.1, 1e-5))
    p_ref = TF.max_pool2d(a, 3, 2, 1)
    gout = rnd(*p_ref.shape, seed=3)
    gw_ref, gg_ref, gb_ref = torch.autograd.grad(p_ref, (wd, gd, bd), gout.double())
    mean_ref, var_ref = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    # ---- HIP ----
    s = L.stream()
    imgd, wdv = img.to(dev), w.to(dev).contiguous()
    stats = torch.zeros(16, 2, 64, device=dev, dtype=torch.float64)
    L.call("awr_stem_stats", L.ptr(imgd), L.ptr(wdv), None, B, H, W, L.ptr(stats), 0, s)
    n = B * H * W
    st = stats.sum(0).cpu()
    assert rel_err(st[0] / n, mean_ref) < 2e-6 and rel_err(st[1] / n - (st[0] / n) ** 2, var_ref) < 2e-5
    coef4 = torch.zeros(4, 64, device=dev)
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    gam, bet = gamma.to(dev), beta.to(dev)
    L.call("awr_bn_finalize", L.ptr(stats), 64, n, L.ptr(gam), L.ptr(bet), L.ptr(rm), L.ptr(rv), 0.1, 1e-5, L.ptr(coef4[0]), L.ptr(coef4[1]),
           L.ptr(coef4[2]), L.ptr(coef4[3]), 0, s)
    assert float(stats.abs().max()) == 0.0                       # re-armed
    pooled = torch.empty(B, H // 2, W // 2, 64, device=dev)
    arg = torch.empty(B, H // 2, W // 2, 64, device=dev, dtype=torch.uint8)
    L.call("awr_stem_pool", L.ptr(imgd), L.ptr(wdv), L.ptr(coef4[0]), L.ptr(coef4[1]), B, H, W, L.ptr(pooled), L.ptr(arg), s)
    assert rel_err(ops.nchw(pooled).cpu(), p_ref.detach()) < 5e-6
    assert int(arg.max()) <= 8
    # eval-mode use: same kernel, folded coefficients, no argmax
    pooled2 = torch.empty_like(pooled)
    L.call("awr_stem_pool", L.ptr(imgd), L.ptr(wdv), L.ptr(coef4[0]), L.ptr(coef4[1]), B, H, W, L.ptr(pooled2), None, s)
    assert torch.equal(pooled, pooled2)
    # ---- backward ----
    dpool = ops.nhwc(gout).to(dev)
    sums = torch.zeros(16, 2, 64, device=dev, dtype=torch.float64)
    L.call("awr_stem_bwd_reduce", L.ptr(imgd), L.ptr(wdv), None, L.ptr(coef4), L.ptr(dpool), L.ptr(arg), B, H, W, L.ptr(sums), 0, s)
    coef = torch.zeros(3, 64, device=dev)
    dgam, dbet = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    L.call("awr_bn_bwd_finalize", L.ptr(sums), 64, n, L.ptr(gam), L.ptr(coef4[3]), L.ptr(coef), L.ptr(dgam), L.ptr(dbet), 0, 0, s)
    assert rel_err(dgam.cpu(), gg_ref) < 2e-5 and rel_err(dbet.cpu(), gb_ref) < 2e-5
    slots = torch.zeros(16 * 64 * 26, device=dev)
    gw = torch.empty(64, 1, 5, 5, device=dev)
    for _ in range(2):                                           # second call: the slot accumulator was re-armed by the first
        L.call("awr_stem_bwd_wgrad", L.ptr(imgd), L.ptr(wdv), None, L.ptr(coef4), L.ptr(coef), L.ptr(dpool), L.ptr(arg), B, H, W, L.ptr(slots), L.ptr(gw), None, 0, s)
        assert rel_err(gw.cpu(), gw_ref) < 5e-5
    assert float(slots.abs().max()) == 0.0


@pytest.mark.parametrize("kind,cin,cout,k,stride,B,H", [("conv", 512, 512, 3, 1, 2, 8), ("conv", 256, 512, 3, 2, 1, 16), ("deconv", 512, 256, 4, 2, 1, 8),
                                                       ("conv", 96, 160, 1, 1, 3, 8)])
def test_conv_split_k(ops, dev, kind, cin, cout, k, stride, B, H):
    """awr_conv_args.partial / split_k: blockIdx.z walks a range of the K slices, the reduce kernel sums the copies in order and applies
    bias, folded-BN affine, residual and ReLU -- against float64 and against the single-pass kernel (low-batch inference path)."""
    pad = 1 if k > 1 else 0
    spec = ops.ConvSpec(kind, cin, cout, k, stride, pad)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w = rnd(*wshape, seed=1, scale=(cin * k * k) ** -0.5)
    x = rnd(B, cin, H, H, seed=2)
    bias, osc, osh = rnd(cout, seed=3), rnd(cout, seed=4) + 1.5, rnd(cout, seed=5)
    isc, ish = rnd(cin, seed=6) + 0.5, rnd(cin, seed=7) * 0.3
    a_in = TF.relu(x.double() * isc.double().view(1, -1, 1, 1) + ish.double().view(1, -1, 1, 1))
    y0 = _torch_fwd(kind, a_in, w.double(), bias.double(), stride, pad)
    res = rnd(*y0.shape, seed=8)
    y_ref = TF.relu(y0 * osc.double().view(1, -1, 1, 1) + osh.double().view(1, -1, 1, 1) + res.double())
    wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
    kw = dict(bias=bias.to(dev), out_scale=osc.to(dev), out_shift=osh.to(dev), res=ops.nhwc(res).to(dev), relu_out=True,
              in_scale=isc.to(dev), in_shift=ish.to(dev), relu_in=True)
    xg = ops.nhwc(x).to(dev)
    y1 = ops.conv_forward(spec, xg, wp, **kw)
    assert rel_err(ops.nchw(y1).cpu(), y_ref) < 5e-6
    for sk in (0, 2, 4, 8):
        part = torch.full((8,) + tuple(y1.shape), float("nan"), device=dev)      # every copy the launch reads must have been written
        y2 = ops.conv_forward(spec, xg, wp, partial=part, split_k=sk, **kw)
        assert rel_err(ops.nchw(y2).cpu(), y_ref) < 5e-6, sk
        assert float((y2 - y1).abs().max()) <= 2e-5 * float(y1.abs().max()), sk
    y3 = ops.conv_forward(spec, xg, wp, partial=part, split_k=4, **kw)
    assert torch.equal(ops.conv_forward(spec, xg, wp, partial=part, split_k=4, **kw), y3)      # fixed summation order


@pytest.mark.parametrize("B,H,W", [(2, 32, 48), (2, 128, 128), (1, 16, 16)])
def test_fused_stem_dense_matches_conv_bias_bn_relu(L, dev, B, H, W):
    """Hourglass stem (hourglass.py:112): conv 5x5 (1 -> 64, pad 2, WITH bias) -> BatchNorm (batch statistics) -> ReLU at full
    resolution, and its backward from the dense gradient of that activation (BN parameter gradients, conv weight and bias
    gradients) by recomputing the conv -- against float64 torch autograd of the three operators."""
    from awr_amd import ops
    g = torch.Generator().manual_seed(11 + H)
    img = (torch.rand(B, 1, H, W, generator=g) * 2 - 1)
    img[:, :, : H // 3] = 1.0
    w = torch.randn(64, 1, 5, 5, generator=g) * 0.2
    bias = torch.randn(64, generator=g) * 0.5
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    wd, bsd = w.double().requires_grad_(True), bias.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = TF.conv2d(img.double(), wd, bsd, 1, 2)
    a_ref = TF.relu(TF.batch_norm(y, None, None, gd, bd, True, 0.1, 1e-5))
    gout = rnd(*a_ref.shape, seed=4)
    gw_ref, gg_ref, gb_ref = torch.autograd.grad(a_ref, (wd, gd, bd), gout.double())
    mean_ref, var_ref = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    s = L.stream()
    imgd, wdv, bsv = img.to(dev), w.to(dev).contiguous(), bias.to(dev)
    stats = torch.zeros(16, 2, 64, device=dev, dtype=torch.float64)
    L.call("awr_stem_stats", L.ptr(imgd), L.ptr(wdv), L.ptr(bsv), B, H, W, L.ptr(stats), 0, s)
    n = B * H * W
    st = stats.sum(0).cpu()
    assert rel_err(st[0] / n, mean_ref) < 2e-6 and rel_err(st[1] / n - (st[0] / n) ** 2, var_ref) < 2e-5
    coef4 = torch.zeros(4, 64, device=dev)
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    gam, bet = gamma.to(dev), beta.to(dev)
    L.call("awr_bn_finalize", L.ptr(stats), 64, n, L.ptr(gam), L.ptr(bet), L.ptr(rm), L.ptr(rv), 0.1, 1e-5, L.ptr(coef4[0]), L.ptr(coef4[1]),
           L.ptr(coef4[2]), L.ptr(coef4[3]), 0, s)
    assert rel_err(rm.cpu(), 0.1 * mean_ref) < 2e-6
    act = torch.empty(B, H, W, 64, device=dev)
    L.call("awr_stem_conv", L.ptr(imgd), L.ptr(wdv), L.ptr(bsv), L.ptr(coef4[0]), L.ptr(coef4[1]), 1, B, H, W, L.ptr(act), s)
    assert rel_err(ops.nchw(act).cpu(), a_ref.detach()) < 5e-6
    pre = torch.empty_like(act)                               # relu = 0: the plain normalised map
    L.call("awr_stem_conv", L.ptr(imgd), L.ptr(wdv), L.ptr(bsv), L.ptr(coef4[0]), L.ptr(coef4[1]), 0, B, H, W, L.ptr(pre), s)
    assert rel_err(ops.nchw(pre).cpu(), TF.batch_norm(y, None, None, gd, bd, True, 0.1, 1e-5).detach()) < 5e-6
    # ---- backward from the dense gradient ----
    dg = ops.nhwc(gout).to(dev)
    sums = torch.zeros(16, 2, 64, device=dev, dtype=torch.float64)
    L.call("awr_stem_bwd_reduce", L.ptr(imgd), L.ptr(wdv), L.ptr(bsv), L.ptr(coef4), L.ptr(dg), None, B, H, W, L.ptr(sums), 0, s)
    coef = torch.zeros(3, 64, device=dev)
    dgam, dbet = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    L.call("awr_bn_bwd_finalize", L.ptr(sums), 64, n, L.ptr(gam), L.ptr(coef4[3]), L.ptr(coef), L.ptr(dgam), L.ptr(dbet), 0, 0, s)
    assert rel_err(dgam.cpu(), gg_ref) < 2e-5 and rel_err(dbet.cpu(), gb_ref) < 2e-5
    slots = torch.zeros(16 * 64 * 26, device=dev)
    gw, gbias = torch.empty(64, 1, 5, 5, device=dev), torch.full((64,), 7.0, device=dev)
    for _ in range(2):
        L.call("awr_stem_bwd_wgrad", L.ptr(imgd), L.ptr(wdv), L.ptr(bsv), L.ptr(coef4), L.ptr(coef), L.ptr(dg), None, B, H, W, L.ptr(slots), L.ptr(gw),
               L.ptr(gbias), 0, s)
        assert rel_err(gw.cpu(), gw_ref) < 5e-5
        # a bias in front of a BatchNorm has a zero gradient in exact arithmetic: what is left is rounding noise of a sum of n terms
        assert float(gbias.abs().max()) < 1e-6 * n * float(gout.abs().mean())
    assert float(slots.abs().max()) == 0.0
    # the bias column proper: with the identity as "BatchNorm backward" (k1 = k2 = 0, gamma*invstd = 1) it is sum of relu' * g
    ident = torch.zeros(3, 64, device=dev)
    ident[2] = 1.0
    L.call("awr_stem_bwd_wgrad", L.ptr(imgd), L.ptr(wdv), L.ptr(bsv), L.ptr(coef4), L.ptr(ident), L.ptr(dg), None, B, H, W, L.ptr(slots), L.ptr(gw),
           L.ptr(gbias), 0, s)
    assert rel_err(gbias.cpu(), (gout.double() * (a_ref.detach() > 0)).sum((0, 2, 3))) < 2e-5


@pytest.mark.parametrize("kind,cin,cout,k,stride,B,H", [
    ("conv", 64, 64, 3, 1, 2, 32), ("conv", 96, 160, 3, 1, 3, 16),        # ragged channel tiles, odd batch
    ("conv", 64, 128, 3, 2, 2, 32), ("conv", 128, 256, 3, 2, 3, 16),      # strided: D map 16x16 / 8x8
    ("deconv", 128, 64, 4, 2, 2, 8), ("deconv", 512, 256, 4, 2, 1, 8), ("deconv", 96, 96, 4, 2, 3, 16),
    ("conv", 256, 256, 3, 1, 1, 8),                                        # 8x8 map: two patches per image
])
def test_wgrad_one_wave_per_tap(ops, L, dev, kind, cin, cout, k, stride, B, H):
    """awr_conv_wgrad algo 2 (a workgroup owns a 64x64 channel tile for all taps, wave t contracts tap t from one staged D patch
    + halo'd G patch) against float64 autograd and against algo 1, with the fused BatchNorm+ReLU loader and the bias-gradient
    by-product, and with split-K chunk counts that do not divide the patch count."""
    spec = ops.ConvSpec(kind, cin, cout, k, stride, 1)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    x = rnd(B, cin, H, H, seed=2)
    s_, t_ = rnd(cin, seed=5) + 0.3, rnd(cin, seed=6) * 0.5
    a = TF.relu(x * s_.view(1, -1, 1, 1) + t_.view(1, -1, 1, 1))
    wd = torch.zeros(*wshape, dtype=torch.float64, requires_grad=True)
    y_ref = _torch_fwd(kind, a.double(), wd, None, stride, 1)
    gy = rnd(*y_ref.shape, seed=4)
    (gw_ref,) = torch.autograd.grad(y_ref, [wd], gy.double())
    xg, gyg = ops.nhwc(x).to(dev), ops.nhwc(gy).to(dev)
    aff = (s_.to(dev), t_.to(dev), True)
    bg = torch.empty(cout, device=dev) if kind == "conv" else None
    gw = ops.conv_wgrad(spec, xg, gyg, x_affine=aff, bias_grad=bg, algo=2)
    assert rel_err(gw.cpu(), gw_ref) < 1e-5
    if bg is not None:
        assert rel_err(bg.cpu(), gy.double().sum((0, 2, 3))) < 1e-5
    old = ops.conv_wgrad(spec, xg, gyg, x_affine=aff, algo=1)
    assert rel_err(gw.cpu(), old.cpu()) < 2e-6
    # split-K depths that leave a ragged last chunk / a single chunk
    import ctypes as C
    prob = spec.wgrad_problem(H, H)
    D, G = (gyg, xg) if prob["D"] == "dy" else (xg, gyg)
    for blocks in (1, 7, 100000):
        R = torch.zeros(prob["Cd"], len(prob["taps"]), prob["Cg"], device=dev)
        wa = ops.make_wgrad_args(prob, B, D, G, R, prob["Cg"], algo=2, **{"g_affine" if prob["D"] == "dy" else "d_affine": aff})
        wa.target_blocks = blocks
        L.call("awr_conv_wgrad", C.byref(wa), L.stream())
        out = torch.empty(*wshape, device=dev)
        L.call("awr_unpack_wgrad", L.ptr(R), prob["d0"], prob["d1"], spec.T, prob["Cg"], L.ptr(out), 0, L.stream())
        assert rel_err(out.cpu(), gw_ref) < 1e-5, blocks


@pytest.mark.parametrize("cin,cout,B,H,affine", [(64, 64, 2, 32, True), (64, 96, 3, 16, False), (128, 160, 2, 8, True), (96, 64, 1, 64, True), (64, 64, 5, 8, False)])
def test_wgrad_one_workgroup_per_kernel_row(ops, L, dev, cin, cout, B, H, affine):
    """awr_conv_wgrad algo 3 (3x3 stride 1: a workgroup contracts 16 staged D pixels against the three taps of one kernel row from one halo'd
    G row segment, three accumulators per wave, operands by LDS-DMA) against float64 autograd and against algo 1: maps of 64 / 32 / 16 pixels
    (one-row stages) and 8 pixels (two-row stages), ragged channel tiles, the fused BatchNorm + ReLU loader on the gathered operand (halo and
    padding pixels must stay zero through it), the bias gradient from the A fragments, and split-K depths with a ragged last chunk."""
    import ctypes as C
    spec = ops.ConvSpec("conv", cin, cout, 3, 1, 1)
    x = rnd(B, cin, H, H, seed=2)
    s_, t_ = rnd(cin, seed=5) + 0.3, rnd(cin, seed=6) * 0.5
    a = TF.relu(x * s_.view(1, -1, 1, 1) + t_.view(1, -1, 1, 1)) if affine else x
    wd = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    y_ref = TF.conv2d(a.double(), wd, None, 1, 1)
    gy = rnd(*y_ref.shape, seed=4)
    (gw_ref,) = torch.autograd.grad(y_ref, [wd], gy.double())
    xg, gyg = ops.nhwc(x).to(dev), ops.nhwc(gy).to(dev)
    aff = (s_.to(dev), t_.to(dev), True) if affine else None
    bg = torch.empty(cout, device=dev)
    gw = ops.conv_wgrad(spec, xg, gyg, x_affine=aff, bias_grad=bg, algo=3)
    assert rel_err(gw.cpu(), gw_ref) < 1e-5
    assert rel_err(bg.cpu(), gy.double().sum((0, 2, 3))) < 1e-5
    old = ops.conv_wgrad(spec, xg, gyg, x_affine=aff, algo=1)
    assert rel_err(gw.cpu(), old.cpu()) < 2e-6
    prob = spec.wgrad_problem(H, H)
    for blocks in (1, 7, 100000):
        R = torch.zeros(prob["Cd"], 9, prob["Cg"], device=dev)
        wa = ops.make_wgrad_args(prob, B, gyg, xg, R, prob["Cg"], algo=3, **({"g_affine": aff} if aff else {}))
        wa.target_blocks = blocks
        L.call("awr_conv_wgrad", C.byref(wa), L.stream())
        out = torch.empty(cout, cin, 3, 3, device=dev)
        L.call("awr_unpack_wgrad", L.ptr(R), prob["d0"], prob["d1"], 9, prob["Cg"], L.ptr(out), 0, L.stream())
        assert rel_err(out.cpu(), gw_ref) < 1e-5, blocks
    # geometries the kernel does not serve are refused, not mis-computed
    bad = ops.ConvSpec("conv", cin, cout, 3, 2, 1)
    pb = bad.wgrad_problem(H, H)
    Rb = torch.zeros(pb["Cd"], 9, pb["Cg"], device=dev)
    wb = ops.make_wgrad_args(pb, B, ops.nhwc(rnd(B, cout, H // 2, H // 2, seed=7)).to(dev), xg, Rb, pb["Cg"], algo=3)
    with pytest.raises(Exception):
        L.call("awr_conv_wgrad", C.byref(wb), L.stream())


@pytest.mark.parametrize("B,H,cin,k,n1,cx,tm", [(2, 16, 128, 3, 128, 0, 1), (3, 10, 128, 3, 128, 0, 1), (1, 8, 64, 1, 128, 0, 1),      # ragged M (300 pixels), a 1x1 first conv
                                                (2, 16, 128, 3, 128, 128, 1), (3, 10, 64, 3, 64, 64, 2), (3, 10, 64, 3, 64, 64, 1), (2, 16, 64, 3, 64, 0, 2)])
def test_fused_conv_pair(ops, L, dev, B, H, cin, k, n1, cx, tm):
    """awr_conv_args.w2: conv (cin -> n1) -> bias -> folded BatchNorm -> ReLU -> conv 1x1 (n1 [+ cx channels of a second tensor] -> 2 n1) ->
    bias2 [-> + residual] in ONE launch (the hourglass residual's conv2 / bn3 / conv3 / skip at inference, hourglass.py:44-59: identity skip =
    residual, skip conv = extra K of the second GEMM), against float64 and against the two-launch form; 64x128, 128x64 and 64x64 tiles."""
    import ctypes as C
    n2 = 2 * n1
    s1 = ops.ConvSpec("conv", cin, n1, k, 1, k // 2)
    s2 = ops.ConvSpec("conv", n1 + cx, n2, 1, 1, 0)
    x = rnd(B, cin, H, H, seed=1)
    w1, b1 = rnd(n1, cin, k, k, seed=2, scale=0.05), rnd(n1, seed=3)
    w2, b2 = rnd(n2, n1 + cx, 1, 1, seed=4, scale=0.1), rnd(n2, seed=5)
    isc, ish = rnd(cin, seed=6) + 1.5, rnd(cin, seed=7)
    sc, sh = rnd(n1, seed=8) + 1.5, rnd(n1, seed=9)
    res = rnd(B, n2, H, H, seed=10) if not cx else None
    x2 = rnd(B, cx, H, H, seed=11) if cx else None
    a0 = TF.relu(x.double() * isc.double().view(1, -1, 1, 1) + ish.double().view(1, -1, 1, 1))
    mid = TF.relu((TF.conv2d(a0, w1.double(), b1.double(), 1, k // 2)) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    ref = TF.conv2d(torch.cat([mid, x2.double()], 1) if cx else mid, w2.double(), b2.double()) + (res.double() if res is not None else 0.0)
    wp1, wp2 = ops.pack_weight(w1.to(dev), s1.fwd_pack()), ops.pack_weight(w2.to(dev), s2.fwd_pack())
    xg = ops.nhwc(x).to(dev)
    rg = ops.nhwc(res).to(dev) if res is not None else None
    x2g = ops.nhwc(x2).to(dev) if cx else None
    d = lambda t: t.to(dev)
    iscg, ishg, scg, shg, b1g, b2g = d(isc), d(ish), d(sc), d(sh), d(b1), d(b2)
    # two launches
    m2 = ops.conv_forward(s1, xg, wp1, in_scale=iscg, in_shift=ishg, relu_in=True, bias=b1g, out_scale=scg, out_shift=shg, relu_out=True)
    y2 = ops.conv_forward(s2, torch.cat([m2, x2g], 3).contiguous() if cx else m2, wp2, bias=b2g, res=rg)
    # one launch
    prob = s1.fwd_problem(H, H)
    y1 = torch.full((B, H, H, n2), float("nan"), device=dev)
    a = ops.make_conv_args(prob, B, xg, wp1, y1, in_scale=iscg, in_shift=ishg, relu_in=True, bias=b1g, out_scale=scg, out_shift=shg, relu_out=True, res=rg, T=s1.T)
    a.w2, a.bias2, a.N1, a.N, a.tile_m, a.tile_n = L.ptr(wp2), L.ptr(b2g), n1, n2, tm, (2 if n1 == 128 else 1)
    if cx:
        a.in2, a.N1x = L.ptr(x2g), cx
    L.call("awr_conv_gemm", C.byref(a), L.stream())
    torch.cuda.synchronize()
    assert rel_err(ops.nchw(y1).cpu(), ref) < 3e-6 and rel_err(ops.nchw(y2).cpu(), ref) < 3e-6
    assert rel_err(y1.cpu(), y2.cpu()) < 2e-6
    # refusals: other channel counts have no fused form
    a.N1 = 32
    with pytest.raises(Exception):
        L.call("awr_conv_gemm", C.byref(a), L.stream())
    a.N1, a.N = n1, n1
    with pytest.raises(Exception):
        L.call("awr_conv_gemm", C.byref(a), L.stream())


@pytest.mark.parametrize("tm,tn", [(1, 1), (1, 2), (2, 1), (2, 2)])
@pytest.mark.parametrize("cin,cout,B,H", [(128, 160, 3, 10), (256, 128, 2, 16), (64, 96, 2, 12)])      # ragged M and N; 4 / 8 / 2 K-slices
def test_short_k_epilogue_operand_prefetch(ops, L, dev, tm, tn, cin, cout, B, H):
    """1x1 convs with at most eight K-slices whose epilogue reads ONE operand tensor run the instantiation that requests that tensor's
    rows ahead of their use (tile (0, 0) before the K loop): residual + statistics forward (hourglass.py:44-59 conv3 + identity skip), and
    the data gradient with the fused BatchNorm-backward reduction (mask re-derived from y), every tile, against float64 and against the
    plain instantiation (AWR_NO_EPRE is read once per process: the comparison is with the 3x3 / two-operand paths' arithmetic, i.e. float64)."""
    import ctypes as C
    spec = ops.ConvSpec("conv", cin, cout, 1, 1, 0)
    x, w, bias = rnd(B, cin, H, H, seed=1), rnd(cout, cin, 1, 1, seed=2, scale=0.1), rnd(cout, seed=3)
    res = rnd(B, cout, H, H, seed=4)
    pre = TF.conv2d(x.double(), w.double(), bias.double()) + res.double()
    wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
    stats = torch.zeros(16, 2, cout, device=dev, dtype=torch.float64)
    prob = spec.fwd_problem(H, H)
    out = torch.full((B, H, H, prob["N"]), float("nan"), device=dev)
    a = ops.make_conv_args(prob, B, ops.nhwc(x).to(dev), wp, out, bias=bias.to(dev), res=ops.nhwc(res).to(dev), stats=stats, T=spec.T)
    a.tile_m, a.tile_n = tm, tn
    L.call("awr_conv_gemm", C.byref(a), L.stream())
    assert rel_err(ops.nchw(out)[:, :cout].cpu(), pre) < 2e-6
    assert rel_err(stats.sum(0)[0][:cout].cpu(), pre.sum((0, 2, 3))) < 1e-5 and rel_err(stats.sum(0)[1][:cout].cpu(), (pre * pre).sum((0, 2, 3))) < 1e-5
    # data gradient (cout -> cin channels) masked by relu(bn(y)) and reduced for the BatchNorm backward in the same launch
    gy = rnd(B, cout, H, H, seed=5)
    y = rnd(B, cin, H, H, seed=6)
    coef4 = torch.stack([rnd(cin, seed=7) + 1.2, rnd(cin, seed=8) * 0.3, rnd(cin, seed=9) * 0.2, rnd(cin, seed=10) + 1.5])      # scale, shift, mean, invstd
    v = TF.conv_transpose2d(gy.double(), w.double())
    mask = (y.double() * coef4[0].double().view(1, -1, 1, 1) + coef4[1].double().view(1, -1, 1, 1)) > 0
    g_ref = v * mask
    xhat = (y.double() - coef4[2].double().view(1, -1, 1, 1)) * coef4[3].double().view(1, -1, 1, 1)
    wd = ops.pack_weight(w.to(dev), spec.dgrad_pack())
    dprob = spec.dgrad_problem(H, H)
    g = torch.full((B, H, H, dprob["N"]), float("nan"), device=dev)
    sums = torch.zeros(16, 2, dprob["N"], device=dev, dtype=torch.float64)
    yg = torch.zeros(B, H, H, dprob["N"], device=dev)
    yg[..., :cin] = ops.nhwc(y).to(dev)
    c4 = torch.zeros(4, dprob["N"], device=dev)
    c4[:, :cin] = coef4.to(dev)
    d = ops.make_conv_args(dprob, B, ops.nhwc(gy).to(dev) if cout == dprob["Cin"] else torch.nn.functional.pad(ops.nhwc(gy), (0, dprob["Cin"] - cout)).to(dev),
                           wd, g, stats=sums, T=spec.T)
    d.bnr_y, d.bnr_coef, d.tile_m, d.tile_n = L.ptr(yg), L.ptr(c4), tm, tn
    L.call("awr_conv_gemm", C.byref(d), L.stream())
    torch.cuda.synchronize()
    assert rel_err(ops.nchw(g)[:, :cin].cpu(), g_ref) < 3e-6
    assert rel_err(sums.sum(0)[0][:cin].cpu(), g_ref.sum((0, 2, 3))) < 1e-5 and rel_err(sums.sum(0)[1][:cin].cpu(), (g_ref * xhat).sum((0, 2, 3))) < 2e-5


@pytest.mark.parametrize("staging_at_launch", [2, 0])
def test_deterministic_wgrad_copies_survive_a_mode_switch(ops, L, dev, staging_at_launch):
    """a deterministic plan sizes its K-chunk copies of a weight gradient when it is built; the kernel a launch picks follows the
    process-wide staging mode AT LAUNCH TIME and may write fewer copies -- the batched scatter sums ALL of them.  awr_conv_wgrad zero-fills the
    copies it leaves unwritten: poisoned scratch, copies sized in one mode, launched in the other, and the sum of all copies is the gradient."""
    import ctypes as C
    spec = ops.ConvSpec("conv", 64, 64, 3, 1, 1)
    B, H = 4, 16
    x, gy = rnd(B, 64, H, H, seed=1), rnd(B, 64, H, H, seed=2)
    prob = spec.wgrad_problem(H, H)
    D, G = (ops.nhwc(gy).to(dev), ops.nhwc(x).to(dev)) if prob["D"] == "dy" else (ops.nhwc(x).to(dev), ops.nhwc(gy).to(dev))
    ld = prob["Cg"]
    rsize = prob["Cd"] * len(prob["taps"]) * ld
    ref = TF.conv2d(x.double().transpose(0, 1), gy.double().transpose(0, 1), padding=1).transpose(0, 1)      # (cout, cin, 3, 3)
    sizes = {}
    try:
        for mode in (2, 0):          # how many copies each mode's kernel writes
            L.call("awr_set_gemm_staging", mode)
            a = ops.make_wgrad_args(prob, B, D, G, D, ld)
            a.split_stride, a.max_split = rsize, 64
            n = C.c_int()
            L.call("awr_conv_wgrad_splits", C.byref(a), C.byref(n))
            sizes[mode] = n.value
        built = max(sizes.values()) + 2                      # what a plan built in the "larger" mode (and then some) would have allocated
        R = torch.full((built, rsize), float("nan"), device=dev)
        L.call("awr_set_gemm_staging", staging_at_launch)
        a = ops.make_wgrad_args(prob, B, D, G, R, ld)
        a.split_stride, a.max_split = rsize, built
        L.call("awr_conv_wgrad", C.byref(a), L.stream())
        torch.cuda.synchronize()
        assert not torch.isnan(R).any()
        assert bool((R[sizes[staging_at_launch]:] == 0).all())              # the unwritten copies were cleared, not left stale
        grad = torch.empty(prob["d0"], prob["d1"], 3, 3, device=dev)
        Rs = R.sum(0).contiguous()
        L.call("awr_unpack_wgrad", L.ptr(Rs), prob["d0"], prob["d1"], spec.T, ld, L.ptr(grad), 0, L.stream())
        assert rel_err(grad.cpu(), ref) < 2e-6
    finally:
        L.call("awr_set_gemm_staging", 2)


@pytest.mark.parametrize("k,s,p,B,H,C,lazy", [(2, 2, 0, 3, 16, 128, False), (2, 2, 0, 2, 8, 256, True), (3, 2, 1, 2, 14, 64, True), (2, 2, 0, 5, 12, 2048, False)])
def test_maxpool_and_upsample_add_with_fused_statistics(L, dev, k, s, p, B, H, C, lazy):
    """Round 5: awr_maxpool_fwd_stats / awr_upsample2_add_stats write exactly what awr_maxpool_fwd / awr_upsample2_add write (values and argmax bit for
    bit) and leave, in the slot layout of awr_channel_stats, the per-channel sum and sum of squares of that tensor -- the statistics pass the next
    BatchNorm would otherwise run over it (hourglass.py:62-88).  Incl. the un-materialised BatchNorm + ReLU on the pool's input, a padded 3x3 / 2
    window, ragged row slabs and the 2048-channel chunking; nearly constant channels keep their variance (shifted sums)."""
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(B, H, H, C, generator=g) * 0.5 + 3.0).to(dev)                 # |mean| >> std on purpose
    sc, sh = ((torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)) if lazy else (None, None)
    Ho = (H + 2 * p - k) // s + 1
    out0, out1 = torch.full((B, Ho, Ho, C), float("nan"), device=dev), torch.full((B, Ho, Ho, C), float("nan"), device=dev)
    arg0, arg1 = torch.zeros(B, Ho, Ho, C, dtype=torch.uint8, device=dev), torch.zeros(B, Ho, Ho, C, dtype=torch.uint8, device=dev)
    st_f, st_r = torch.zeros(16, 2, C, dtype=torch.float64, device=dev), torch.zeros(16, 2, C, dtype=torch.float64, device=dev)
    L.call("awr_maxpool_fwd", L.ptr(x), L.ptr(sc), L.ptr(sh), int(lazy), B, H, H, C, k, s, p, L.ptr(out0), L.ptr(arg0), L.stream())
    L.call("awr_maxpool_fwd_stats", L.ptr(x), L.ptr(sc), L.ptr(sh), int(lazy), B, H, H, C, k, s, p, L.ptr(out1), L.ptr(arg1), L.ptr(st_f), 0, L.stream())
    L.call("awr_channel_stats", L.ptr(out0), B * Ho * Ho, C, L.ptr(st_r), 0, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(out0, out1) and torch.equal(arg0, arg1)
    ref = out0.double().reshape(-1, C)
    for name, got in (("fused", st_f.sum(0)), ("separate", st_r.sum(0))):
        assert float(((got[0] - ref.sum(0)).abs() / ref.sum(0).abs().clamp_min(1e-9)).max()) < 1e-6, name
        assert float(((got[1] - (ref * ref).sum(0)).abs() / (ref * ref).sum(0)).max()) < 1e-6, name
    # variance of a nearly constant channel: computed from the fused sums to 1e-4 relative
    n = ref.shape[0]
    var = st_f.sum(0)[1] / n - (st_f.sum(0)[0] / n) ** 2
    assert float(((var - ref.var(0, unbiased=False)).abs() / ref.var(0, unbiased=False)).max()) < 1e-4
    if k == 2 and not lazy and C <= 1024:      # up-sampling add at the same sizes: out = up1 + up(low)
        up1, low = torch.randn(B, 2 * Ho, 2 * Ho, C, generator=g).to(dev), out0
        o0, o1 = torch.empty_like(up1), torch.empty_like(up1)
        s_f, s_r = torch.zeros(16, 2, C, dtype=torch.float64, device=dev), torch.zeros(16, 2, C, dtype=torch.float64, device=dev)
        L.call("awr_upsample2_add", L.ptr(up1), L.ptr(low), B, Ho, Ho, C, L.ptr(o0), L.stream())
        L.call("awr_upsample2_add_stats", L.ptr(up1), L.ptr(low), B, Ho, Ho, C, L.ptr(o1), L.ptr(s_f), 0, L.stream())
        L.call("awr_channel_stats", L.ptr(o0), B * 4 * Ho * Ho, C, L.ptr(s_r), 0, L.stream())
        torch.cuda.synchronize()
        assert torch.equal(o0, o1)
        r2 = o0.double().reshape(-1, C)
        assert float(((s_f.sum(0)[0] - r2.sum(0)).abs()).max()) < 1e-6 * float(r2.abs().sum(0).max())
        assert float(((s_f.sum(0)[1] - (r2 * r2).sum(0)).abs() / (r2 * r2).sum(0)).max()) < 1e-6


@study_only
@pytest.mark.parametrize("kind,cin,cout,k,stride,pad,B,H", [("conv", 64, 96, 3, 1, 1, 4, 18), ("deconv", 64, 96, 4, 2, 1, 6, 10), ("conv", 96, 64, 1, 1, 0, 2, 12)])
def test_conv_gemm_in_batch_parts_equals_the_whole_launch(ops, L, dev, kind, cin, cout, k, stride, pad, B, H):
    """awr_conv_gemm_part: images are independent rows of the GEMM, so the launch issued as 2 (or B) equal batch parts writes the SAME BITS as the
    whole launch -- input, output, residual and the fused-reduction operands all advance with the part; the statistics of the parts add up."""
    import ctypes as C
    spec = ops.ConvSpec(kind, cin, cout, k, stride, pad)
    x = rnd(B, cin, H, H, seed=1)
    w = rnd(*((cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)), seed=2, scale=0.05)
    prob = spec.fwd_problem(H, H)
    xin = ops.nhwc(x).to(dev)
    if prob["Cin"] != cin:
        xin = torch.nn.functional.pad(xin, (0, prob["Cin"] - cin))
    xin = xin.contiguous()
    wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
    res = rnd(B, prob["Hout"], prob["Wout"], prob["N"], seed=3).to(dev)
    bias = rnd(prob["N"], seed=4).to(dev)
    outs = []
    for nparts in (1, 2, B):
        out = torch.full((B, prob["Hout"], prob["Wout"], prob["N"]), float("nan"), device=dev)
        st = torch.zeros(16, 2, prob["N"], device=dev, dtype=torch.float64)
        a = ops.make_conv_args(prob, B, xin, wp, out, bias=bias, res=res, relu_out=True, stats=st, T=spec.T)
        for part in range(nparts):
            L.call("awr_conv_gemm_part", C.byref(a), nparts, part, L.stream())
        torch.cuda.synchronize()
        outs.append((out.clone(), st.sum(0).float()))
    for o, s_ in outs[1:]:
        assert torch.equal(outs[0][0], o)
        assert rel_err(s_.cpu(), outs[0][1].cpu()) < 1e-6
    with pytest.raises(L.AwrError):
        L.call("awr_conv_gemm_part", C.byref(a), 3 if B % 3 else 5, 0, L.stream())      # the batch is not divisible by the part count


@pytest.mark.parametrize("kind,cin,cout,k,stride,pad,B,H", [("conv", 64, 96, 3, 1, 1, 4, 18), ("deconv", 64, 96, 4, 2, 1, 6, 10), ("conv", 96, 64, 1, 1, 0, 2, 12),
                                                             ("conv", 128, 256, 1, 1, 0, 3, 16)])
def test_streaming_output_stores_write_the_same_bits(ops, L, dev, kind, cin, cout, k, stride, pad, B, H):
    """awr_conv_args.out_nt: 1 = cached, 2 = streaming (`buffer_store ... nt`, the epilogue's residual loads `nt` too), 0 = the library's size / K-extent rule.
    A cache policy: every form writes the SAME BITS and the same statistics, with the residual + ReLU + statistics epilogue and on the short-K
    operand-prefetch form (the two 1x1 shapes); an unknown policy is refused."""
    import ctypes as C
    spec = ops.ConvSpec(kind, cin, cout, k, stride, pad)
    x = rnd(B, cin, H, H, seed=1)
    w = rnd(*((cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)), seed=2, scale=0.05)
    prob = spec.fwd_problem(H, H)
    xin = ops.nhwc(x).to(dev).contiguous()
    wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
    res = rnd(B, prob["Hout"], prob["Wout"], prob["N"], seed=3).to(dev)
    bias = rnd(prob["N"], seed=4).to(dev)
    outs = []
    for policy in (1, 2, 0):
        out = torch.full((B, prob["Hout"], prob["Wout"], prob["N"]), float("nan"), device=dev)
        st = torch.zeros(16, 2, prob["N"], device=dev, dtype=torch.float64)
        a = ops.make_conv_args(prob, B, xin, wp, out, bias=bias, res=res, relu_out=True, stats=st, T=spec.T)
        a.out_nt = policy
        L.call("awr_conv_gemm", C.byref(a), L.stream())
        torch.cuda.synchronize()
        outs.append((out.clone(), st.sum(0).float()))
    assert not torch.isnan(outs[0][0]).any()
    for o, s_ in outs[1:]:
        assert torch.equal(outs[0][0], o)
        assert rel_err(s_.cpu(), outs[0][1].cpu()) < 1e-6
    a.out_nt = 3
    with pytest.raises(L.AwrError):
        L.call("awr_conv_gemm", C.byref(a), L.stream())


@pytest.mark.parametrize("tile", [(1, 1), (2, 1), (1, 2)])
@pytest.mark.parametrize("affine", [False, True])
def test_blocked_accumulation_in_the_deep_pipeline_is_bit_identical(ops, L, dev, tile, affine):
    """Round 6: launches that cannot fill the chip (<= 384 workgroups: layer4 at batch 64, the two-image fixtures) take the deep pipeline (four stage
    buffers) in the BLOCKED accumulation mode too -- same k order, same fold points every 128 k, so the same bits as the two-buffer blocked kernel;
    and blocked differs from ordered (otherwise the flag did nothing).  3x3, K = 1152, statistics epilogue, with / without the fused input affine."""
    import ctypes as C
    B, H, cin, cout = 2, 8, 128, 128
    spec = ops.ConvSpec("conv", cin, cout, 3, 1, 1)
    x, w = rnd(B, cin, H, H, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.05)
    prob = spec.fwd_problem(H, H)
    xin = ops.nhwc(x).to(dev).contiguous()
    wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
    sc, sh = (rnd(cin, seed=5).abs().to(dev) + 0.5, rnd(cin, seed=6).to(dev)) if affine else (None, None)
    got = {}
    L.call("awr_debug_force_tile", *tile)
    try:
        for accum, deep in ((1, 1), (1, 0), (0, 1)):
            L.call("awr_debug_set_knob", b"deep", deep)
            out = torch.full((B, H, H, prob["N"]), float("nan"), device=dev)
            st = torch.zeros(16, 2, prob["N"], device=dev, dtype=torch.float64)
            a = ops.make_conv_args(prob, B, xin, wp, out, in_scale=sc, in_shift=sh, relu_in=affine, stats=st, T=spec.T)
            a.accum = accum
            L.call("awr_conv_gemm", C.byref(a), L.stream())
            torch.cuda.synchronize()
            got[(accum, deep)] = (out.clone(), st.sum(0).clone())
    finally:
        L.call("awr_debug_set_knob", b"deep", 1)
        L.call("awr_debug_force_tile", 0, 0)
    assert not torch.isnan(got[(1, 1)][0]).any()
    assert torch.equal(got[(1, 1)][0], got[(1, 0)][0]) and torch.equal(got[(1, 1)][1], got[(1, 0)][1])
    assert not torch.equal(got[(1, 1)][0], got[(0, 1)][0])
    assert rel_err(got[(1, 1)][0].cpu(), got[(0, 1)][0].cpu()) < 1e-5


@pytest.mark.parametrize("tile", [(1, 1), (2, 1), (1, 2), (2, 2)])
def test_statistics_from_the_accumulators_match_the_row_layout_form(ops, L, dev, tile):
    """EM 5 (round 5): a statistics launch whose stored value is accumulator + bias on tiles wholly inside M sums x - c and (x - c)^2 in the ACCUMULATOR
    layout (a lane owns one channel, its registers 16 rows) instead of behind the LDS bounce.  Same output bits; sum and sum of squares agree with the
    row-layout form (awr_debug_set_knob fast_stats = 0) and with float64 sums of the stored tensor -- on a ragged N (96 of a 128-column tile), with a bias that puts the
    mean 100 standard deviations from zero (the shift is what keeps the variance)."""
    import ctypes as C
    B, H, cin, cout = 2, 16, 64, 96            # M = 512: a multiple of both tile heights
    spec = ops.ConvSpec("conv", cin, cout, 3, 1, 1)
    x, w = rnd(B, cin, H, H, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.05)
    bias = (rnd(cout, seed=3) + 100.0).to(dev)
    prob = spec.fwd_problem(H, H)
    xin = ops.nhwc(x).to(dev).contiguous()
    wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
    got = {}
    L.call("awr_debug_force_tile", *tile)
    try:
        for fast in ("0", "1"):
            L.call("awr_debug_set_knob", b"fast_stats", int(fast))
            out = torch.full((B, H, H, prob["N"]), float("nan"), device=dev)
            st = torch.zeros(16, 2, prob["N"], device=dev, dtype=torch.float64)
            a = ops.make_conv_args(prob, B, xin, wp, out, bias=bias, stats=st, T=spec.T)
            L.call("awr_conv_gemm", C.byref(a), L.stream())
            torch.cuda.synchronize()
            got[fast] = (out.clone(), st.sum(0).clone())
    finally:
        L.call("awr_debug_set_knob", b"fast_stats", 1)
        L.call("awr_debug_force_tile", 0, 0)
    assert torch.equal(got["0"][0], got["1"][0])
    ref = got["1"][0].double().reshape(-1, prob["N"])
    n = ref.shape[0]
    for k in ("0", "1"):
        s1, s2 = got[k][1][0], got[k][1][1]
        assert float((s1 - ref.sum(0)).abs().max()) < 1e-7 * float(ref.abs().sum(0).max())
        var, var_ref = s2 / n - (s1 / n) ** 2, ref.var(0, unbiased=False)
        assert float(((var - var_ref).abs() / var_ref).max()) < 1e-4      # mean ~ 100, std ~ 1: the shifted sums keep five digits of the variance
