"""GPU: the Winograd F(2x2, 3x3) study kernel (csrc/awr_wino.hip, operator level) against float64 torch-CPU convolutions: forward with bias /
ReLU, ragged tiles (patch counts that are not a multiple of 64), several maps per tile (4x4 maps), and the data-gradient form (mirrored
weights) against autograd.  Winograd is not bit-compatible with the direct kernel: the bar is a relative error against float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import awr_amd  # noqa: F401
    from awr_amd import _lib as L, ops
    return L, ops, torch.device("cuda:0")


def wino(L, x_nhwc, w, bias, relu, kb, mirror=False):
    dev = x_nhwc.device
    B, H, W, C = x_nhwc.shape
    N = w.shape[1] if mirror else w.shape[0]
    Npad = (N + 31) // 32 * 32
    U = torch.zeros(16, C, Npad, device=dev)
    L.call("awr_wino_weights", L.ptr(w.to(dev).contiguous()), N, C, Npad, C, int(mirror), L.ptr(U), L.stream())
    out = torch.full((B, H, W, Npad), float("nan"), device=dev)
    bp = None
    if bias is not None:
        bp = torch.zeros(Npad, device=dev)
        bp[:N] = bias.to(dev)
    L.call("awr_wino_conv3x3", L.ptr(x_nhwc), L.ptr(U), L.ptr(bp), L.ptr(out), B, H, W, C, Npad, int(relu), kb, L.stream())
    torch.cuda.synchronize()
    return out[..., :N].cpu()


@pytest.mark.parametrize("kb", [8, 4, 108, 104])
@pytest.mark.parametrize("B,H,W,cin,cout,bias,relu", [(2, 16, 16, 64, 64, True, True), (3, 8, 12, 32, 96, False, False), (5, 4, 4, 128, 40, True, False),
                                                       (1, 2, 2, 8, 32, False, True), (2, 32, 32, 16, 128, True, False)])
def test_winograd_forward_matches_float64(env, kb, B, H, W, cin, cout, bias, relu):
    L, ops, dev = env
    g = torch.Generator().manual_seed(B * 100 + H + cin)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    if relu:
        ref = ref.clamp(min=0)
    got = wino(L, x.permute(0, 2, 3, 1).contiguous().to(dev), w, b, relu, kb).permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    err = float((got.double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, err


def test_winograd_data_gradient_form(env):
    """mirror = 1: d(x) = conv(d(y), w mirrored, channel roles swapped) -- against autograd in float64"""
    L, ops, dev = env
    g = torch.Generator().manual_seed(7)
    B, H, cin, cout = 2, 16, 48 + 16, 32
    x = torch.randn(B, cin, H, H, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    dy = torch.randn(B, cout, H, H, generator=g)
    (gx,) = torch.autograd.grad((torch.nn.functional.conv2d(x, w.double(), padding=1) * dy.double()).sum(), x)
    got = wino(L, dy.permute(0, 2, 3, 1).contiguous().to(dev), w, None, False, 8, mirror=True).permute(0, 3, 1, 2)
    err = float((got.double() - gx).abs().max()) / float(gx.abs().max())
    assert err < 2e-5, err


def wino2(L, x_nhwc, w, bias, relu, in_affine=None, relu_in=False, stats=False):
    dev = x_nhwc.device
    B, H, W, C = x_nhwc.shape
    N = w.shape[0]
    Npad = (N + 31) // 32 * 32
    U = torch.zeros(16, C, Npad, device=dev)
    L.call("awr_wino_weights", L.ptr(w.to(dev).contiguous()), N, C, Npad, C, 0, L.ptr(U), L.stream())
    out = torch.full((B, H, W, Npad), float("nan"), device=dev)
    bp = None
    if bias is not None:
        bp = torch.zeros(Npad, device=dev)
        bp[:N] = bias.to(dev)
    st = torch.zeros(16, 2, Npad, device=dev, dtype=torch.float64) if stats else None
    sc, sh = (in_affine[0].to(dev), in_affine[1].to(dev)) if in_affine is not None else (None, None)
    L.call("awr_wino2_conv3x3", L.ptr(x_nhwc), L.ptr(U), L.ptr(bp), L.ptr(sc), L.ptr(sh), int(relu_in), L.ptr(out), L.ptr(st), 16 if stats else 0,
           B, H, W, C, Npad, int(relu), L.stream())
    torch.cuda.synchronize()
    return out[..., :N].cpu(), (st.sum(0)[:, :N].cpu() if stats else None)


@pytest.mark.parametrize("B,H,W,cin,cout,bias,relu,affine", [(2, 16, 16, 64, 64, True, True, False), (3, 8, 16, 32, 96, False, False, True), (5, 4, 4, 128, 40, True, False, True),
                                                              (18, 4, 4, 8, 32, False, True, False), (2, 64, 64, 16, 32, True, False, True), (1, 128, 128, 8, 32, False, False, False),
                                                              (3, 32, 8, 24, 64, True, False, True)])
@pytest.mark.parametrize("wide", [False, True])
def test_winograd_v2_forward_prologue_and_statistics(env, B, H, W, cin, cout, bias, relu, affine, wide):
    """the raw-tile form: 2-D patch tiles (several small images per tile, ragged last tile), fused input affine + ReLU with zero padding kept
    zero, per-channel statistics of the stored output"""
    L, ops, dev = env
    g = torch.Generator().manual_seed(B * 100 + H + cin)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    sc, sh = (torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3) if affine else (None, None)
    xin = x.double()
    if affine:
        xin = (xin * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).clamp(min=0)
    ref = torch.nn.functional.conv2d(xin, w.double(), None if b is None else b.double(), padding=1)
    if relu:
        ref = ref.clamp(min=0)
    L.call("awr_set_conv_winograd", 4 if wide else 12)       # + 4: the 64-channel tile form whenever N % 64 == 0 (whatever the launch size); + 8: never
    try:
        got, st = wino2(L, x.permute(0, 2, 3, 1).contiguous().to(dev), w, b, relu, (sc, sh) if affine else None, relu_in=affine, stats=True)
    finally:
        L.call("awr_set_conv_winograd", 0)
    got = got.permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    err = float((got.double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, err
    s1, s2 = ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))
    assert float((st[0] - s1).abs().max()) <= 2e-5 * float(ref.abs().max()) * ref[:, 0].numel()
    assert float((st[1] - s2).abs().max()) <= 2e-5 * float(s2.abs().max())


@pytest.mark.parametrize("use_res,use_bnr,use_act", [(False, False, False), (True, False, False), (False, True, False), (True, True, False), (True, True, True)])
def test_winograd_data_gradient_epilogues(env, use_res, use_bnr, use_act):
    """awr_wino_conv with the data-gradient epilogue forms plans use (include/awr_hip.h: awr_wino_args): mirrored weights, accumulate onto the gradient
    already in `out`, mask with the re-derived ReLU (or the stored activation's sign) and reduce sum g / sum g * xhat for the BatchNorm backward --
    against float64 autograd + the same formulas in torch."""
    import ctypes as C
    L, ops, dev = env
    g = torch.Generator().manual_seed(11 + use_res + 2 * use_bnr + 4 * use_act)
    B, H, W, cin, cout = 3, 16, 32, 64, 96
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    dy = torch.randn(B, cout, H, W, generator=g)
    x = torch.zeros(B, cin, H, W, dtype=torch.float64, requires_grad=True)
    (dx,) = torch.autograd.grad((torch.nn.functional.conv2d(x, w.double(), padding=1) * dy.double()).sum(), x)
    r = torch.randn(B, cin, H, W, generator=g)
    yb = torch.randn(B, cin, H, W, generator=g)
    act = torch.randn(B, cin, H, W, generator=g)
    sc, sh, mu, istd = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3, torch.randn(cin, generator=g) * 0.2, torch.rand(cin, generator=g) + 0.5
    v = dx + (r.double() if use_res else 0)
    if use_bnr:
        mask = (act > 0) if use_act else (yb.double() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) > 0)
        v = v * mask
        s1 = v.sum((0, 2, 3))
        s2 = (v * ((yb.double() - mu.view(1, -1, 1, 1)) * istd.view(1, -1, 1, 1))).sum((0, 2, 3))
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().float().to(dev)      # noqa: E731
    U = torch.zeros(16, cout, cin, device=dev)
    L.call("awr_wino_weights", L.ptr(w.to(dev).contiguous()), cin, cout, cin, cout, 1, L.ptr(U), L.stream())
    out = nhwc(r) if use_res else torch.full((B, H, W, cin), float("nan"), device=dev)
    din, yy, aa = nhwc(dy), nhwc(yb), nhwc(act)
    coef = torch.cat([sc, sh, mu, istd]).to(dev).contiguous()
    st = torch.zeros(16, 2, cin, device=dev, dtype=torch.float64)
    a = L.WinoArgs()
    a.in_, a.U, a.out = L.ptr(din), L.ptr(U), L.ptr(out)
    a.B, a.H, a.W, a.C, a.N = B, H, W, cout, cin
    if use_res:
        a.res = L.ptr(out)
    if use_bnr:
        a.bnr_y, a.bnr_coef, a.stats, a.nslots = L.ptr(yy), L.ptr(coef), L.ptr(st), 16
        if use_act:
            a.bnr_act = L.ptr(aa)
    L.call("awr_wino_conv", C.byref(a), L.stream())
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2).double()
    scale = float(v.abs().max())
    # a mask decided by a float32 sign may flip where the float64 argument is within rounding of zero: compare where both agree
    diff = (got - v).abs()
    assert float((diff > 2e-5 * scale).float().mean()) < 1e-4 and float(diff.max()) < 1.0 * scale
    if use_bnr:
        tot = st.sum(0).cpu()
        assert float((tot[0] - s1).abs().max()) <= 1e-4 * float(v.abs().sum((0, 2, 3)).max())
        assert float((tot[1] - s2).abs().max()) <= 1e-4 * float(v.abs().sum((0, 2, 3)).max()) * 3
    # and the dispatcher: an argument block this kernel does not implement goes through the direct kernel (same numbers within fp32)
    from awr_amd._lib import ConvArgs
    d = ConvArgs()
    assert L.lib.awr_wino_dgrad_supported(C.byref(d)) == 1
    d.relu_out = 1
    assert L.lib.awr_wino_dgrad_supported(C.byref(d)) == 0




@pytest.mark.parametrize("B,H,W,cin,cout,affine,relu_in,bias", [(2, 16, 16, 64, 64, False, False, True), (3, 8, 8, 128, 64, True, True, True),
                                                                (1, 32, 16, 64, 128, True, False, False), (5, 8, 16, 64, 64, False, True, True),
                                                                (4, 32, 32, 128, 128, True, True, True)])
def test_winograd_weight_gradient_matches_float64(env, B, H, W, cin, cout, affine, relu_in, bias):
    """awr_wino_wgrad: dg = G^T [sum_patches (B^T d B)(.)(A dY A^T)] G against autograd in float64 -- with the fused input affine / ReLU (padding
    stays zero), the bias gradient, split-K ranges that do not divide the patch blocks evenly (B = 3, 5), packed rows longer than C (ld)"""
    L, ops, dev = env
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + cin)
    x = torch.randn(B, cin, H, W, generator=g)
    dy = torch.randn(B, cout, H, W, generator=g)
    sc = (torch.rand(cin, generator=g) + 0.5) if affine else None
    sh = (torch.randn(cin, generator=g) * 0.3) if affine else None
    a = x.double()
    if affine:
        a = a * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if relu_in:
        a = a.clamp(min=0)
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (gw,) = torch.autograd.grad((torch.nn.functional.conv2d(a, w, padding=1) * dy.double()).sum(), w)
    gb = dy.double().sum(dim=(0, 2, 3))
    assert L.lib.awr_wino_wgrad_eligible(B, H, W, cin, cout) in (0, 1)
    n = int(L.lib.awr_wino_wgrad_scratch(B, H, W, cin, cout))
    scratch = torch.full((n,), float("nan"), device=dev)
    ld = cin + 64
    R = torch.full((cout, 9, ld), float("nan"), device=dev)
    bg = torch.full((cout,), float("nan"), device=dev) if bias else None
    xd, dyd = x.permute(0, 2, 3, 1).contiguous().to(dev), dy.permute(0, 2, 3, 1).contiguous().to(dev)
    scd, shd = (sc.to(dev), sh.to(dev)) if affine else (None, None)
    L.call("awr_wino_wgrad", L.ptr(xd), L.ptr(dyd), L.ptr(scd), L.ptr(shd), int(relu_in), B, H, W, cin, cout, L.ptr(scratch), L.ptr(R), ld, L.ptr(bg), L.stream())
    torch.cuda.synchronize()
    got = R[:, :, :cin].cpu().permute(0, 2, 1).reshape(cout, cin, 3, 3)
    assert torch.isfinite(got).all()
    assert torch.isnan(R[:, :, cin:]).all()                 # nothing beyond the C columns of a packed row is touched
    err = float((got.double() - gw).abs().max()) / float(gw.abs().max())
    assert err < 2e-5, err
    if bias:
        eb = float((bg.cpu().double() - gb).abs().max()) / float(gb.abs().max())
        assert eb < 2e-5, eb
    # twice the same launch: bit-identical (ordered sums over the split copies, no atomics)
    R2 = torch.empty_like(R)
    L.call("awr_wino_wgrad", L.ptr(xd), L.ptr(dyd), L.ptr(scd), L.ptr(shd), int(relu_in), B, H, W, cin, cout, L.ptr(scratch), L.ptr(R2), ld, L.ptr(bg), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(R2[:, :, :cin], R[:, :, :cin])
