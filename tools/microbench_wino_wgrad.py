"""Winograd-domain weight gradient (csrc/awr_wino.hip: wino_wgrad_kernel + wino_wgrad_reduce_kernel) against the direct awr_conv_wgrad on the stride-1
3x3 shapes of the BASELINE networks: time of both launches together, algorithmic / executed TFLOP/s, max error against float64 (on a two-image
slice of the same statistics).   -> profiles/r06_winograd.txt"""
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import awr_amd  # noqa: E402,F401
from awr_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")


def time_us(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(name, B, H, cin, cout, affine=True):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, H, cin, generator=g).to(dev)
    dy = torch.randn(B, H, H, cout, generator=g).to(dev)
    sc, sh = (torch.rand(cin, generator=g) + 0.5).to(dev), (torch.randn(cin, generator=g) * 0.2).to(dev)
    spec = ops.ConvSpec("conv", cin, cout, 3, 1, 1)
    prob = spec.wgrad_problem(H, H)
    Rd = torch.zeros(cout, 9, cin, device=dev)
    aff = {"g_affine": (sc, sh, True)} if affine else {}
    res = {}
    for algo in (0, 2, 3):
        a = ops.make_wgrad_args(prob, B, dy, x, Rd, cin, algo=algo, **aff)
        if not L.lib.awr_conv_wgrad_algo_ok(C.byref(a), algo):
            continue
        res[algo] = time_us(lambda: L.call("awr_conv_wgrad", C.byref(a), L.stream()))
    best = min(res, key=res.get)
    Rd.zero_()
    a = ops.make_wgrad_args(prob, B, dy, x, Rd, cin, algo=best, **aff)
    L.call("awr_conv_wgrad", C.byref(a), L.stream())
    n = int(L.lib.awr_wino_wgrad_scratch(B, H, H, cin, cout))
    scratch = torch.empty(n, device=dev)
    Rw = torch.empty(cout, 9, cin, device=dev)
    scp, shp = (L.ptr(sc), L.ptr(sh)) if affine else (None, None)
    t_w = time_us(lambda: L.call("awr_wino_wgrad", L.ptr(x), L.ptr(dy), scp, shp, int(affine), B, H, H, cin, cout, L.ptr(scratch), L.ptr(Rw), cin, None, L.stream()))
    torch.cuda.synchronize()
    # float64 reference
    a64 = x.double()
    if affine:
        a64 = (a64 * sc.double() + sh.double()).clamp(min=0)
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device=dev, requires_grad=True)
    (gw,) = torch.autograd.grad((torch.nn.functional.conv2d(a64.permute(0, 3, 1, 2), w, padding=1) * dy.double().permute(0, 3, 1, 2)).sum(), w)
    ref = gw.permute(0, 2, 3, 1).reshape(cout, 9, cin)
    scale = float(ref.abs().max())
    e_d = float((Rd.double() - ref).abs().max()) / scale
    e_w = float((Rw.double() - ref).abs().max()) / scale
    fl = 2.0 * 9 * cin * cout * B * H * H
    print("%-30s B=%3d %3dx%-3d %3d->%-3d | direct (algo %d) %7.1f us %6.1f TF | winograd %7.1f us  %6.1f TF algorithmic  %6.1f TF executed | x%.2f | eligible %d | "
          "scratch %.0f MB | max err / max|g|: direct %.2e  wino %.2e (x%.1f) | direct algos: %s" % (
              name, B, H, H, cin, cout, best, res[best], fl / res[best] / 1e6, t_w, fl / t_w / 1e6, fl * 16 / 36 / t_w / 1e6, res[best] / t_w,
              L.lib.awr_wino_wgrad_eligible(B, H, H, cin, cout), n * 4 / 1e6, e_d, e_w, e_w / e_d, {k: round(v, 1) for k, v in res.items()}), flush=True)


if __name__ == "__main__":
    L.call("awr_set_conv_winograd", 4)
    if "--quick" in sys.argv:
        run("hg 3x3 128->128 @64 (HG-1)", 64, 64, 128, 128)
        run("hg 128->128 @32", 64, 32, 128, 128)
        sys.exit(0)
    print("Winograd-domain weight gradient vs awr_conv_wgrad (input behind the fused BatchNorm affine + ReLU), HIP events, 20 reps")
    run("hg 3x3 128->128 @64 (HG-1)", 64, 64, 128, 128)
    run("hg 3x3 128->128 @64 plain x", 64, 64, 128, 128, affine=False)
    run("hg 3x3 128->128 @64 B=128", 128, 64, 128, 128)
    run("hg 64->64 @64 (pre.1)", 64, 64, 64, 64)
    run("hg 128->128 @32", 64, 32, 128, 128)
    run("hg 128->128 @16", 64, 16, 128, 128)
    run("layer1 64->64 @32 (R18)", 64, 32, 64, 64, affine=False)
    run("layer2 128->128 @16 (R18)", 64, 16, 128, 128, affine=False)
    run("layer3 256->256 @8 (R18)", 64, 8, 256, 256, affine=False)
    run("hg2 64->64 @128 (cfg5)", 32, 128, 64, 64)
    run("layer1 B=256 (cfg4)", 256, 32, 64, 64, affine=False)
