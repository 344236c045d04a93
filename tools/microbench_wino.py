"""Winograd F(2x2, 3x3) study: the fused FP32-MFMA Winograd kernel (csrc/awr_wino.hip) against the direct LDS-DMA implicit
GEMM on the stride-1 3x3 shapes of the BASELINE networks -- algorithmic TFLOP/s (2 * 9 * Cin * Cout * pixels per launch over time), executed
MFMA TFLOP/s (Winograd executes 16 / 36 of the algorithmic multiplies), max error against float64.   -> profiles/r06_winograd.txt"""
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import awr_amd  # noqa: E402
from awr_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
PROBES = "--probes" in sys.argv      # timing-only variants: 1008 = input window loaded once, 2008 = weights loaded once, 3008 = nothing loaded after stage 0,
                                     # 4008 = additionally no transform / LDS writes (MFMA + fragment reads + barriers)


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


def run(name, B, H, cin, cout, err_b=2):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    xin = ops.nhwc(x).to(dev).contiguous()
    spec = ops.ConvSpec("conv", cin, cout, 3, 1, 1)
    prob = spec.fwd_problem(H, H)
    wp = ops.pack_weight(w.to(dev), spec.fwd_pack())
    out_d = torch.empty(B, H, H, cout, device=dev)
    a = ops.make_conv_args(prob, B, xin, wp, out_d, T=spec.T)
    t_direct = time_us(lambda: L.call("awr_conv_gemm", C.byref(a), L.stream()))
    U = torch.empty(16, cin, cout, device=dev)
    wd = w.to(dev).contiguous()
    L.call("awr_wino_weights", L.ptr(wd), cout, cin, cout, cin, 0, L.ptr(U), L.stream())
    out_w = torch.empty(B, H, H, cout, device=dev)
    res = {}
    for kb in (8, 4, 108, 104) + ((1008, 2008, 3008, 4008) if PROBES else ()):
        res[kb] = time_us(lambda: L.call("awr_wino_conv3x3", L.ptr(xin), L.ptr(U), None, L.ptr(out_w), B, H, H, cin, cout, 0, kb, L.stream()))
    pow2 = H & (H - 1) == 0
    if pow2:
        L.call("awr_set_conv_winograd", 8)           # 32-channel tiles only
        res[2] = time_us(lambda: L.call("awr_wino2_conv3x3", L.ptr(xin), L.ptr(U), None, None, None, 0, L.ptr(out_w), None, 0, B, H, H, cin, cout, 0, L.stream()))
        if cout % 64 == 0:           # the 64-channel tile form (16 waves, one workgroup per CU)
            L.call("awr_set_conv_winograd", 0)       # automatic: the 64-channel tile form where it fills the chip
            res[3] = time_us(lambda: L.call("awr_wino2_conv3x3", L.ptr(xin), L.ptr(U), None, None, None, 0, L.ptr(out_w), None, 0, B, H, H, cin, cout, 0, L.stream()))
            L.call("awr_set_conv_winograd", 0)
    best = min((k for k in res if k < 1000), key=res.get)
    if best in (2, 3):
        L.call("awr_set_conv_winograd", 0 if best == 3 else 8)
        L.call("awr_wino2_conv3x3", L.ptr(xin), L.ptr(U), None, None, None, 0, L.ptr(out_w), None, 0, B, H, H, cin, cout, 0, L.stream())
    else:
        L.call("awr_wino_conv3x3", L.ptr(xin), L.ptr(U), None, L.ptr(out_w), B, H, H, cin, cout, 0, best, L.stream())
    torch.cuda.synchronize()
    L.call("awr_set_conv_winograd", 0)
    ref = torch.nn.functional.conv2d(x[:err_b].double(), w.double(), padding=1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    e_d = float((out_d[:err_b].cpu().double() - ref).abs().max()) / scale
    e_w = float((out_w[:err_b].cpu().double() - ref).abs().max()) / scale
    rms_d = float((out_d[:err_b].cpu().double() - ref).pow(2).mean().sqrt()) / scale
    rms_w = float((out_w[:err_b].cpu().double() - ref).pow(2).mean().sqrt()) / scale
    fl = 2.0 * 9 * cin * cout * B * H * H
    print("%-28s B=%3d %3dx%-3d %3d->%-3d | direct %7.1f us %6.1f TF | winograd(form %3d; 2 = v2, 3 = v2 wide) %7.1f us  %6.1f TF algorithmic  %6.1f TF executed | "
          "x%.2f | max err / max|y|: direct %.2e  wino %.2e (x%.1f) | rms: %.2e  %.2e (x%.1f) | all kb: %s" % (
              name, B, H, H, cin, cout, t_direct, fl / t_direct / 1e6, best, res[best], fl / res[best] / 1e6, fl * 16 / 36 / res[best] / 1e6,
              t_direct / res[best], e_d, e_w, e_w / e_d, rms_d, rms_w, rms_w / rms_d, {k: round(v, 1) for k, v in res.items()}), flush=True)


if __name__ == "__main__":
    print("Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32 vs the direct LDS-DMA implicit GEMM (both plain epilogues, no bias), HIP events, 20 reps")
    run("hg 3x3 128->128 @64 (HG-1)", 64, 64, 128, 128)
    run("hg 3x3 128->128 @64 B=128", 128, 64, 128, 128)
    run("layer1 64->64 @32 (R18)", 64, 32, 64, 64)
    run("layer1 64->64 @64", 64, 64, 64, 64)
    run("layer2 128->128 @16 (R18)", 64, 16, 128, 128)
    run("layer3 256->256 @8 (R18)", 64, 8, 256, 256)
    run("layer3 256->256 @16", 64, 16, 256, 256)
    run("layer4 512->512 @4 (R18)", 64, 4, 512, 512)
    run("hg2 64->64 @128 (cfg5)", 32, 128, 64, 64)
    run("layer1 B=256 (cfg4)", 256, 32, 64, 64)
