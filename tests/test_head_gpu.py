"""GPU parity: HIP head / GT-map / loss / optimiser kernels (through the C ABI) vs the oracle."""
import math
import os

import numpy as np
import pytest
import torch

import awr_oracle as O

pytestmark = pytest.mark.gpu

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
def amd():
    import awr_amd
    return awr_amd


def _hashed(shape, stream, scale):
    return torch.from_numpy((O._hash_uniform(int(np.prod(shape)), stream, 99) * np.float32(2 * scale)).reshape(shape).copy())


CASES = [(2, 14, 128, 0.4), (3, 14, 128, 1.0), (2, 21, 256, 0.4), (1, 14, 64, 0.4), (2, 16, 128, 0.7)]


@pytest.mark.parametrize("B,J,H,ks", CASES)
def test_joint2offset_bit_exact(amd, dev, B, J, H, ks):
    """The GT map has hard thresholds (heat map >= 0, depth < 0.99): the kernel restates util/feature_tool.py:29-39 operation for
    operation (no FMA contraction, correctly rounded sqrt / divide) and must reproduce the IEEE-754 restatement of those lines
    (oracle.joint2offset_ieee: numpy float32, every operation correctly rounded) BIT FOR BIT.  The torch-CPU oracle itself is NOT that:
    torch's CPU sqrt (MKL VML) is off by one ulp for 0.6 % of arguments on an Intel host and more elsewhere (tools/debug_j2o.py), so against
    it the map may differ by one unit in the last place of the affected elements -- never in its mask."""
    img, jt = O.synth_batch(B, H, J, seed=21)
    F = H // 2
    out = amd.FeatureModule().joint2offset(jt.to(dev), img.to(dev), ks, F).cpu()
    ieee = torch.from_numpy(O.joint2offset_ieee(jt, img, ks, F))
    nbad = int((out != ieee).sum())
    assert torch.equal(out, ieee), "max diff %g, %d elements differ from the IEEE restatement" % (float((out - ieee).abs().max()), nbad)
    ref = O.joint2offset(jt, img, ks, F)
    assert torch.equal(out != 0, ref != 0)                                   # same mask as the torch oracle
    assert float((out - ref).abs().max()) <= 2.4e-7                          # <= 1 ulp of values in [1, 2): torch's sqrt


@pytest.mark.parametrize("B,J,H,ks", CASES)
def test_head_forward_backward(amd, dev, B, J, H, ks):
    img, _ = O.synth_batch(B, H, J, seed=22)
    F = H // 2
    off = _hashed((B, 4 * J, F, F), 3, 0.6)
    g_jt = _hashed((B, J, 3), 4, 1.0)
    x = off.to(dev).requires_grad_(True)
    jt = amd.FeatureModule().offset2joint_softmax(x, img.to(dev), ks)
    ref = O.offset2joint_softmax(off, img, ks)
    d = float((jt.detach().cpu() - ref).abs().max())
    # tolerance: 1e-3 mm of a 150 mm half-cube = 6.7e-6 normalised (north_star); we hold 3e-6
    assert d <= 3e-6, d
    (jt * g_jt.to(dev)).sum().backward()
    gref = O.head_backward(off, img, ks, g_jt)
    gd = float((x.grad.cpu() - gref).abs().max())
    assert gd <= 3e-6 * float(gref.abs().max()) + 1e-9, (gd, float(gref.abs().max()))


def _to_nhwc(x, cp):
    """(B, C, F, F) -> the backbone's (B, F*F, Cp) rows, padding channels zero"""
    B, C, F, _ = x.shape
    out = torch.zeros(B, F * F, cp, dtype=x.dtype, device=x.device)
    out[:, :, :C] = x.permute(0, 2, 3, 1).reshape(B, F * F, C)
    return out.contiguous()


@pytest.mark.parametrize("cw", [0.0, 1.0])
@pytest.mark.parametrize("B,J,H,ks", CASES + [(64, 14, 128, 1.0)])
def test_nhwc_head_and_loss_step(amd, dev, B, J, H, ks, cw):
    """The NHWC forms the fused engines use (awr_head_forward_nhwc, awr_head_loss_step_nhwc: joints, both Huber losses and the total
    gradient w.r.t. the dense map in one call, on the (B, P, Cp) layout of the head GEMM) against the oracle's formulas + autograd."""
    import ctypes as C
    from awr_amd import _lib as L
    img, jt_gt = O.synth_batch(B, H, J, seed=31)
    F, cp = H // 2, (4 * J + 31) // 32 * 32
    off = _hashed((B, 4 * J, F, F), 5, 0.6)
    x = off.clone().requires_grad_(True)
    jt_ref = O.offset2joint_softmax(x, img, ks)
    lc, ld = O.huber(jt_ref, jt_gt), O.huber(x, O.joint2offset(jt_gt, img, ks, F))
    (cw * lc + 1.0 * ld).backward()
    pred = _to_nhwc(off, cp).to(dev)
    pred[:, :, 4 * J:] = 0
    n = int(L.lib.awr_head_nhwc_scratch(B, J, F))
    scratch, jt, stat = torch.zeros(n, device=dev), torch.zeros(B, J, 3, device=dev), torch.zeros(B, J, 2, device=dev)
    g_jt, acc = torch.zeros(B, J, 3, device=dev), torch.zeros(2, device=dev, dtype=torch.float64)
    grad = torch.full((B, F * F, cp), float("nan"), device=dev)
    losses = torch.zeros(3, device=dev)
    imd, jgd = img.to(dev), jt_gt.to(dev)
    s = L.stream()
    L.call("awr_head_loss_step_nhwc", L.ptr(pred), cp, L.ptr(imd), L.ptr(jgd), B, J, F, H, ks, 0.01, cw, 1.0, L.ptr(scratch), L.ptr(jt), L.ptr(stat),
           L.ptr(g_jt), L.ptr(acc), L.ptr(grad), s)
    L.call("awr_loss_finalize", L.ptr(acc), 2, L.ptr(losses), s)
    torch.cuda.synchronize()
    assert float((jt.cpu() - jt_ref.detach()).abs().max()) <= 3e-6
    assert abs(float(losses[0]) - cw * float(lc)) <= 2e-6 * max(1e-3, cw * float(lc)) + 1e-9
    assert abs(float(losses[1]) - float(ld)) <= 2e-6 * float(ld) + 1e-9
    g = grad.cpu()
    assert bool((g[:, :, 4 * J:] == 0).all()), "padding channels of the gradient must be zero"
    gref = _to_nhwc(x.grad, cp)
    # the dense part is clamp(z, +-0.01) / N: elements sitting on the Huber kink / a GT-map threshold may differ by one quantum
    d = (g - gref).abs()
    assert float(d.max()) <= 3e-6 * float(gref.abs().max()) + 1e-9, (float(d.max()), float(gref.abs().max()))
    # inference form: joints only
    jt2 = torch.zeros(B, J, 3, device=dev)
    L.call("awr_head_forward_nhwc", L.ptr(pred), cp, L.ptr(imd), B, J, F, H, ks, L.ptr(scratch), L.ptr(jt2), None, s)
    torch.cuda.synchronize()
    assert float((jt2.cpu() - jt_ref.detach()).abs().max()) <= 3e-6


def test_head_golden(amd, dev, golden_dir):
    for tag in ("j14_ks04", "j14_ks10", "j21_h256"):
        g = np.load(os.path.join(golden_dir, "head_%s.npz" % tag))
        img = torch.from_numpy(g["img"])
        J, ks = int(g["J"]), float(g["ks"])
        F = img.shape[-1] // 2
        off = _hashed((2, 4 * J, F, F), int(g["offset_stream"]), float(g["offset_scale"])).to(dev).requires_grad_(True)
        jt = amd.FeatureModule().offset2joint_softmax(off, img.to(dev), ks)
        np.testing.assert_allclose(jt.detach().cpu().numpy(), g["jt"], rtol=0, atol=3e-6)
        (jt * torch.from_numpy(g["g_jt"]).to(dev)).sum().backward()
        scale = float(np.abs(g["g_val"]).max())
        np.testing.assert_allclose(off.grad.cpu().reshape(-1).numpy()[g["g_idx"]], g["g_val"], rtol=0, atol=3e-6 * scale + 1e-9)


def test_roundtrip_property_full_size(amd, dev):
    """joint2offset -> offset2joint_softmax recovers joints that lie on the hand surface (the head's
    defining property), at BASELINE's full batch 256."""
    B, J = 256, 14
    img, _ = O.synth_batch(B, 128, J, seed=5)
    d = img[:, 0, ::2, ::2]
    ys = torch.randint(20, 44, (B, J)); xs = torch.randint(20, 44, (B, J))
    a = 2.0 * (torch.arange(64).float() + 0.5) / 64 - 1.0
    dep = d[torch.arange(B).view(B, 1), ys, xs]
    jt = torch.stack([a[xs], a[ys], dep], -1)
    fg = dep < 0.99
    fm = amd.FeatureModule()
    gt = fm.joint2offset(jt.to(dev), img.to(dev), 0.4, 64)
    back = fm.offset2joint_softmax(gt, img.to(dev), 0.4).cpu()
    ref = O.offset2joint_softmax(O.joint2offset(jt, img, 0.4, 64), img, 0.4)
    assert float((back - ref).abs().max()) <= 3e-6
    assert float((back - jt)[fg].abs().max()) < 0.05


def test_nhwc_roundtrip_and_loss_properties_full_size(amd, dev):
    """Size-independent properties of the NHWC forms at BASELINE's full batch (256 / GPU, config 4): (i) the GT map of joints lying on
    the hand surface, fed back as the prediction, decodes to those joints (round trip) and has ZERO dense loss and zero gradient;
    (ii) the dense-loss gradient of a perturbed map is clamp(z, +-0.01) / N element for element (so its absolute sum is bounded by
    0.01 per element and the loss is non-negative); (iii) scaling dense_weight scales loss and gradient linearly."""
    from awr_amd import _lib as L
    B, J, H, ks = 256, 14, 128, 0.4
    F, cp = H // 2, 64
    img, _ = O.synth_batch(B, H, J, seed=6)
    d = img[:, 0, ::2, ::2]
    g = torch.Generator().manual_seed(1)
    ys, xs = torch.randint(20, 44, (B, J), generator=g), torch.randint(20, 44, (B, J), generator=g)
    a = 2.0 * (torch.arange(F).float() + 0.5) / F - 1.0
    dep = d[torch.arange(B).view(B, 1), ys, xs]
    jt_gt = torch.stack([a[xs], a[ys], dep], -1).contiguous()
    imd, jgd = img.to(dev), jt_gt.to(dev)
    gt = amd.FeatureModule().joint2offset(jgd, imd, ks, F)                 # (B, 4J, F, F), bit-exact IEEE GT map
    pred = _to_nhwc(gt, cp)
    n = int(L.lib.awr_head_nhwc_scratch(B, J, F))
    scratch, jt, stat = torch.zeros(n, device=dev), torch.zeros(B, J, 3, device=dev), torch.zeros(B, J, 2, device=dev)
    g_jt, acc, losses = torch.zeros(B, J, 3, device=dev), torch.zeros(2, device=dev, dtype=torch.float64), torch.zeros(3, device=dev)
    grad = torch.full((B, F * F, cp), float("nan"), device=dev)
    s = L.stream()

    def step(p, dw):
        acc.zero_()
        L.call("awr_head_loss_step_nhwc", L.ptr(p), cp, L.ptr(imd), L.ptr(jgd), B, J, F, H, ks, 0.01, 0.0, dw, L.ptr(scratch), L.ptr(jt), L.ptr(stat),
               L.ptr(g_jt), L.ptr(acc), L.ptr(grad), s)
        L.call("awr_loss_finalize_reset", L.ptr(acc), 2, L.ptr(losses), s)
        torch.cuda.synchronize()
        return float(losses[1]), grad.clone(), jt.clone()
    l0, g0, j0 = step(pred, 1.0)
    assert l0 == 0.0 and float(g0.abs().max()) == 0.0 and float(acc.abs().max()) == 0.0           # (i) exact: the fused GT map IS the prediction
    fg = (dep < 0.99).to(dev)
    assert float((j0 - jgd)[fg].abs().max()) < 0.05
    noise = torch.from_numpy((O._hash_uniform(pred.numel(), 11, 3) * np.float32(0.1)).reshape(pred.shape).copy()).to(dev)
    noise[:, :, 4 * J:] = 0
    l1, g1, _ = step(pred + noise, 1.0)
    N = B * 4 * J * F * F
    z = noise[:, :, :4 * J]
    assert torch.allclose(g1[:, :, :4 * J] * N, z.clamp(-0.01, 0.01), rtol=1e-5, atol=3e-7) and l1 > 0                        # (ii)
    assert bool((g1[:, :, 4 * J:] == 0).all())
    l2, g2, _ = step(pred + noise, 2.0)
    assert abs(l2 - 2.0 * l1) <= 1e-6 * l2 and torch.allclose(g2, 2.0 * g1, rtol=1e-6, atol=0)                              # (iii)


def test_huber_and_dense_loss(amd, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "huber.npz"))
    x = torch.from_numpy(g["x"]).to(dev).requires_grad_(True)
    y = torch.from_numpy(g["y"]).to(dev)
    crit = amd.My_SmoothL1Loss().cuda()
    loss = crit(x, y)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-7 * max(1.0, float(g["loss"]))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["gx"], rtol=0, atol=1e-9)
    # fused GT-map + dense Huber vs oracle (loss and gradient), incl. accumulate
    from awr_amd import _lib as L
    for (B, J, H, ks) in [(2, 14, 128, 0.4), (2, 14, 128, 1.0), (1, 21, 256, 0.4)]:
        img, jt = O.synth_batch(B, H, J, seed=23)
        F = H // 2
        pred = _hashed((B, 4 * J, F, F), 6, 0.05).requires_grad_(True)
        lref = 0.7 * O.huber(pred, O.joint2offset(jt, img, ks, F))
        (gref,) = torch.autograd.grad(lref, pred)
        acc = torch.zeros(2, device=dev, dtype=torch.float64)
        gout = torch.empty((B, 4 * J, F, F), device=dev)
        p, j_, i_ = pred.detach().to(dev), jt.to(dev), img.to(dev)
        L.call("awr_dense_loss", L.ptr(p), L.ptr(j_), L.ptr(i_), B, J, F, H, ks, 0.01, 0.7, L.ptr(acc), L.ptr(gout), 0, L.stream())
        out = torch.empty(3, device=dev)
        L.call("awr_loss_finalize", L.ptr(acc), 2, L.ptr(out), L.stream())
        assert abs(float(out[0]) - float(lref)) <= 2e-6 * max(1e-3, float(lref)), (float(out[0]), float(lref))
        assert float(out[2]) == float(out[0])
        assert float((gout.cpu() - gref).abs().max()) <= 1e-12 + 1e-6 * float(gref.abs().max())
        acc2 = torch.zeros(2, device=dev, dtype=torch.float64)      # accumulate=1 adds onto the existing gradient
        L.call("awr_dense_loss", L.ptr(p), L.ptr(j_), L.ptr(i_), B, J, F, H, ks, 0.01, 0.7, L.ptr(acc2), L.ptr(gout), 1, L.stream())
        assert float((gout.cpu() - 2 * gref).abs().max()) <= 1e-12 + 1e-6 * float(gref.abs().max())


@pytest.mark.parametrize("n", [7, 4096, 1000003])
def test_adam_and_sgd(dev, n):
    from awr_amd import _lib as L
    g0 = torch.Generator().manual_seed(n)
    p = torch.randn(n, generator=g0); p_ref = p.clone()
    m = torch.zeros(n); v = torch.zeros(n)
    pd, md, vd = p.to(dev), m.to(dev), v.to(dev)
    for step in range(1, 4):
        g = torch.randn(n, generator=g0) * 0.01
        O.adam_update(p_ref, g, m, v, step, lr=1e-3, wd=0.0)
        L.call("awr_adam_step", L.ptr(pd), DP(L, g, dev), L.ptr(md), L.ptr(vd), n, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, 1.0, L.stream())
    assert float((pd.cpu() - p_ref).abs().max()) <= 5e-7      # 1 ulp at |p| ~ 4
    assert float((md.cpu() - m).abs().max()) <= 1e-8 and float((vd.cpu() - v).abs().max()) <= 1e-10
    # SGD momentum vs torch.optim.SGD
    q = torch.randn(n, generator=g0).requires_grad_(True)
    qd, bd = q.detach().clone().to(dev), torch.zeros(n, device=dev)
    opt = torch.optim.SGD([q], lr=0.01, momentum=0.9)
    for step in range(1, 4):
        g = torch.randn(n, generator=g0)
        q.grad = g.clone(); opt.step()
        L.call("awr_sgd_step", L.ptr(qd), DP(L, g, dev), L.ptr(bd), n, 0.01, 0.9, 0.0, step, 1.0, L.stream())
    assert float((qd.cpu() - q.detach()).abs().max()) <= 1e-6


def test_errors_are_loud(amd, dev):
    from awr_amd import _lib as L
    fm = amd.FeatureModule()
    with pytest.raises(L.AwrError):
        fm.joint2offset(torch.zeros(1, 14, 3), torch.zeros(1, 1, 128, 128), 0.4, 64)      # CPU tensors: no fallback
    with pytest.raises(L.AwrError):
        fm.joint2offset(torch.zeros(1, 14, 3, device=dev), torch.zeros(1, 1, 128, 128, device=dev), 0.4, 63)  # F % 4 != 0
