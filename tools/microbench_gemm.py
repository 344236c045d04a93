#!/usr/bin/env python3
"""Microbenchmark of awr_conv_gemm / awr_conv_wgrad on single layers (GPU box only).

    python tools/microbench_gemm.py ksweep      # time vs number of K-slices at fixed M,N: fixed cost per workgroup
    python tools/microbench_gemm.py layers      # the ResNet18 layer shapes at --batch
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import awr_amd  # noqa: E402,F401
from awr_amd import _lib as L  # noqa: E402
from awr_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def run_fwd(spec, B, H, tile=None, stats=False, affine=False, presplit=False):
    dev = torch.device("cuda:0")
    x = torch.randn(B, H, H, spec.cin_pad, device=dev)
    wshape = (spec.cout, spec.cin, spec.k, spec.k) if spec.kind == "conv" else (spec.cin, spec.cout, spec.k, spec.k)
    w = torch.randn(*wshape, device=dev) * 0.05
    wp = ops.pack_weight(w, spec.fwd_pack())
    prob = spec.fwd_problem(H, H)
    out = torch.empty(B, prob["Hout"], prob["Wout"], prob["N"], device=dev)
    st = torch.zeros(16, 2, prob["N"], device=dev, dtype=torch.float64) if stats else None
    aff = {}
    if affine:      # the un-materialised BatchNorm + ReLU input
        aff = dict(in_scale=torch.rand(spec.cin_pad, device=dev) + 0.5, in_shift=torch.randn(spec.cin_pad, device=dev) * 0.1, relu_in=True)
    if presplit:      # split-operand mode, activation image cut once by the producer (LDS-DMA kernel)
        aff = dict(in_split=ops.split_act(x, aff.get("in_scale"), aff.get("in_shift"), aff.get("relu_in", False)))
    a = ops.make_conv_args(prob, B, x, wp, out, stats=st, T=spec.T, **aff)
    if tile:
        L.call("awr_debug_force_tile", *tile)
    s = L.stream()
    t = timeit(lambda: L.check(L.lib.awr_conv_gemm(C.byref(a), s)))
    L.call("awr_debug_force_tile", 0, 0)
    macs = B * prob["Hout"] * prob["Wout"] * spec.cout * spec.cin * (spec.T if spec.kind == "conv" else spec.T / 4)
    return t, 2 * macs / t / 1e12


def run_split_act(C_, B, H, affine=False):
    """the producer-side cut on its own: one pass, 4 B read + 6 B written per element"""
    dev = torch.device("cuda:0")
    x = torch.randn(B, H, H, C_, device=dev)
    out = torch.empty(x.numel() * 3, device=dev, dtype=torch.int16)
    sc = torch.rand(C_, device=dev) + 0.5 if affine else None
    sh = torch.randn(C_, device=dev) if affine else None
    s = L.stream()
    t = timeit(lambda: L.call("awr_split_act", L.ptr(x), L.ptr(sc), L.ptr(sh), int(affine), x.numel() // C_, C_, L.ptr(out), s))
    return t, x.numel() * 10 / t / 1e9


def run_wgrad(spec, B, H, tile=None, algo=0, blocks=0, affine=False):
    dev = torch.device("cuda:0")
    prob = spec.wgrad_problem(H, H)
    ho, wo = spec.out_hw(H, H)
    x = torch.randn(B, H, H, spec.cin_pad, device=dev)
    dy = torch.randn(B, ho, wo, spec.cout_pad, device=dev)
    D, G = (dy, x) if prob["D"] == "dy" else (x, dy)
    R = torch.zeros(prob["Cd"], len(prob["taps"]), prob["Cg"], device=dev)
    aff = {}
    if affine:      # the un-materialised BatchNorm+ReLU loader on the layer-input operand
        aff = {"g_affine" if prob["D"] == "dy" else "d_affine": (torch.rand(spec.cin_pad, device=dev) + 0.5, torch.randn(spec.cin_pad, device=dev) * 0.1, True)}
    a = ops.make_wgrad_args(prob, B, D, G, R, prob["Cg"], algo=algo, **aff)
    a.target_blocks = blocks
    if tile and isinstance(tile, tuple) and len(tile) == 2 and algo != 2:
        a.tile_m, a.tile_n = tile
    s = L.stream()
    t = timeit(lambda: L.check(L.lib.awr_conv_wgrad(C.byref(a), s)))
    L.call("awr_debug_force_tile", 0, 0)
    macs = B * ho * wo * spec.cout * spec.cin * (spec.T if spec.kind == "conv" else spec.T / 4)
    return t, 2 * macs / t / 1e12


def run_split(spec, B, H, tile_a, tile_b):
    """Two half-batch launches of the same layer with DIFFERENT tiles on two streams vs one full launch."""
    dev = torch.device("cuda:0")
    x = torch.randn(B, H, H, spec.cin_pad, device=dev)
    wshape = (spec.cout, spec.cin, spec.k, spec.k) if spec.kind == "conv" else (spec.cin, spec.cout, spec.k, spec.k)
    w = torch.randn(*wshape, device=dev) * 0.05
    wp = ops.pack_weight(w, spec.fwd_pack())
    prob = spec.fwd_problem(H, H)
    out = torch.empty(B, prob["Hout"], prob["Wout"], prob["N"], device=dev)
    hb = B // 2
    a0 = ops.make_conv_args(prob, hb, x[:hb], wp, out[:hb], T=spec.T)
    a1 = ops.make_conv_args(prob, B - hb, x[hb:], wp, out[hb:], T=spec.T)
    a0.tile_m, a0.tile_n = tile_a
    a1.tile_m, a1.tile_n = tile_b
    side = torch.cuda.Stream()

    def fn():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        L.check(L.lib.awr_conv_gemm(C.byref(a0), main.cuda_stream))
        L.check(L.lib.awr_conv_gemm(C.byref(a1), side.cuda_stream))
        main.wait_stream(side)
    t = timeit(fn)
    macs = B * prob["Hout"] * prob["Wout"] * spec.cout * spec.cin * (spec.T if spec.kind == "conv" else spec.T / 4)
    return t, 2 * macs / t / 1e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["ksweep", "layers", "split", "tiles", "wgrad", "fwdset", "wgradset", "rowset", "splitset", "smallset"])
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    if args.mode == "ksweep":
        print("1x1 conv, M = 64*64*64 = 262144 rows; time = a + b * (K/32)")
        for tile, cout in (((2, 2), 128), ((2, 1), 64), ((1, 1), 64)):
            pts = []
            for nk in (1, 2, 4, 8, 16, 32, 64):
                t, tf = run_fwd(ops.ConvSpec("conv", 32 * nk, cout, 1, 1, 0), 64, 64, tile)
                pts.append((nk, t))
                print("tile %s N=%3d  K-slices %3d : %8.1f us  %6.1f TF" % (tile, cout, nk, t * 1e6, tf))
            (n0, t0), (n1, t1) = pts[-2], pts[-1]
            b = (t1 - t0) / (n1 - n0)
            print("   -> per-slice %.2f us, fixed %.1f us  (= %.1f slices)" % (b * 1e6, (t1 - b * n1) * 1e6, (t1 - b * n1) / b))
    elif args.mode == "splitset":    # split-operand mode: the in-kernel cut (rounds 1-4) against the LDS-DMA kernel on a pre-cut activation image
        B = args.batch
        L.call("awr_set_gemm_products", 6)
        shapes = [("layer1 3x3 64->64 @64", ops.ConvSpec("conv", 64, 64, 3, 1, 1), 64), ("layer2 3x3 128->128 @32", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 32),
                  ("layer3 3x3 256->256 @16", ops.ConvSpec("conv", 256, 256, 3, 1, 1), 16), ("layer4 3x3 512->512 @8", ops.ConvSpec("conv", 512, 512, 3, 1, 1), 8),
                  ("deconv 512->256 @8", ops.ConvSpec("deconv", 512, 256, 4, 2, 1), 8), ("deconv 256->256 @16", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 16),
                  ("deconv 256->256 @32", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 32), ("head 1x1 256->64 @64", ops.ConvSpec("conv", 256, 64, 1, 1, 0), 64),
                  ("hg 1x1 256->128 @64", ops.ConvSpec("conv", 256, 128, 1, 1, 0), 64), ("hg 1x1 128->256 @64", ops.ConvSpec("conv", 128, 256, 1, 1, 0), 64),
                  ("hg 3x3 128->128 @64", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 64)]
        print("split-operand mode, batch %d: TF-equivalent per tile: in-kernel cut plain|stats|stats+affine  ->  pre-cut image + LDS-DMA plain|stats" % B)
        for name, spec, H in shapes:
            res = []
            for tile in ((1, 1), (2, 1), (1, 2), (2, 2)):
                if spec.cout <= 64 and tile[1] == 2:
                    continue
                res.append("%s %5.1f|%5.1f|%5.1f -> %5.1f|%5.1f" % (tile, run_fwd(spec, B, H, tile)[1], run_fwd(spec, B, H, tile, stats=True)[1],
                                                                 run_fwd(spec, B, H, tile, stats=True, affine=True)[1],
                                                                 run_fwd(spec, B, H, tile, presplit=True)[1], run_fwd(spec, B, H, tile, stats=True, presplit=True)[1]))
            print("%-26s %s" % (name, "  ".join(res)), flush=True)
        print("awr_split_act alone (one pass, 10 B per element): us, GB/s -- plain | with BatchNorm affine + ReLU")
        for C_, H in ((64, 64), (128, 32), (256, 16), (512, 8), (256, 64), (128, 64)):
            t0, g0 = run_split_act(C_, B, H)
            t1, g1 = run_split_act(C_, B, H, True)
            print("  C=%3d @%2d: %7.1f us %6.0f GB/s | %7.1f us %6.0f GB/s" % (C_, H, t0 * 1e6, g0, t1 * 1e6, g1))
        L.call("awr_set_gemm_products", 1)
    elif args.mode == "smallset":    # launches that cannot fill the chip: the deep pipeline (AWR_DEEP=1, four stage buffers) against two buffers (0)
        shapes = [("r18 layer1 3x3 64 @64 B4", ops.ConvSpec("conv", 64, 64, 3, 1, 1), 64, 4), ("r18 layer2 3x3 128 @32 B4", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 32, 4),
                  ("r18 layer3 3x3 256 @16 B4", ops.ConvSpec("conv", 256, 256, 3, 1, 1), 16, 4), ("r18 deconv 256->256 @32 B4", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 32, 4),
                  ("r18 layer4 3x3 512 @8 B64", ops.ConvSpec("conv", 512, 512, 3, 1, 1), 8, 64), ("r18 layer3 3x3 256 @16 B16", ops.ConvSpec("conv", 256, 256, 3, 1, 1), 16, 16),
                  ("hg 3x3 128 @16 B64", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 16, 64), ("hg 3x3 128 @8 B64", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 8, 64),
                  ("hg 3x3 128 @4 B64", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 4, 64), ("hg 1x1 256->128 @16 B64", ops.ConvSpec("conv", 256, 128, 1, 1, 0), 16, 64),
                  ("hg 1x1 128->256 @8 B64", ops.ConvSpec("conv", 128, 256, 1, 1, 0), 8, 64), ("hg 1x1 256->128 @4 B64", ops.ConvSpec("conv", 256, 128, 1, 1, 0), 4, 64)]
        print("AWR_DEEP: us per launch, plain | statistics | statistics + affine -- two stage buffers -> deep (four)")
        for name, spec, H, B in shapes:
            res = []
            for tile in ((1, 1), (2, 1), (1, 2)):
                if spec.cout <= 64 and tile[1] == 2:
                    continue
                cell = []
                for deep in ("0", "1"):
                    os.environ["AWR_DEEP"] = deep
                    cell.append("%5.1f|%5.1f|%5.1f" % (run_fwd(spec, B, H, tile)[0] * 1e6, run_fwd(spec, B, H, tile, stats=True)[0] * 1e6, run_fwd(spec, B, H, tile, stats=True, affine=True)[0] * 1e6))
                res.append("%s %s -> %s" % (tile, cell[0], cell[1]))
            print("%-28s %s" % (name, "   ".join(res)), flush=True)
        os.environ.pop("AWR_DEEP", None)
    elif args.mode == "fwdset":      # forward / data-gradient GEMM only, every tile, plain and with the BatchNorm statistics epilogue: the
        B = args.batch                # same-box A/B of the staging variants (AWR_DMA=0..3, one process each; tools/gpu_r4_a.sh)
        shapes = [("layer1 3x3 64->64 @64", ops.ConvSpec("conv", 64, 64, 3, 1, 1), 64), ("layer2 3x3 128->128 @32", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 32),
                  ("layer3 3x3 256->256 @16", ops.ConvSpec("conv", 256, 256, 3, 1, 1), 16), ("layer4 3x3 512->512 @8", ops.ConvSpec("conv", 512, 512, 3, 1, 1), 8),
                  ("deconv 512->256 @8", ops.ConvSpec("deconv", 512, 256, 4, 2, 1), 8), ("deconv 256->256 @16", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 16),
                  ("deconv 256->256 @32", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 32), ("head 1x1 256->64 @64", ops.ConvSpec("conv", 256, 64, 1, 1, 0), 64),
                  ("hg 1x1 256->128 @64", ops.ConvSpec("conv", 256, 128, 1, 1, 0), 64), ("hg 1x1 128->256 @64", ops.ConvSpec("conv", 128, 256, 1, 1, 0), 64),
                  ("hg 3x3 128->128 @64", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 64), ("hg 1x1 256->128 @16", ops.ConvSpec("conv", 256, 128, 1, 1, 0), 16)]
        print("AWR_DMA=%s batch %d: TF per tile, plain | with statistics epilogue | statistics + fused input affine" % (os.environ.get("AWR_DMA", "2"), B))
        for name, spec, H in shapes:
            res = []
            for tile in ((1, 1), (2, 1), (1, 2), (2, 2)):
                if spec.cout <= 64 and tile[1] == 2:
                    continue
                res.append("%s %5.1f|%5.1f|%5.1f" % (tile, run_fwd(spec, B, H, tile)[1], run_fwd(spec, B, H, tile, stats=True)[1], run_fwd(spec, B, H, tile, stats=True, affine=True)[1]))
            print("%-26s %s" % (name, "  ".join(res)), flush=True)
    elif args.mode == "wgradset":    # workgroup-per-tap weight gradient, every tile x split-K candidate of the plan autotuner; plain and with the
        B = args.batch                # un-materialised BatchNorm loader on the layer input (AWR_WGRAD_DMA / AWR_WGRAD_KP: one process per variant)
        shapes = [("layer1 3x3 64->64 @64", ops.ConvSpec("conv", 64, 64, 3, 1, 1), 64), ("layer2 3x3 128->128 @32", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 32),
                  ("layer3 3x3 256->256 @16", ops.ConvSpec("conv", 256, 256, 3, 1, 1), 16), ("layer4 3x3 512->512 @8", ops.ConvSpec("conv", 512, 512, 3, 1, 1), 8),
                  ("layer2.0 3x3s2 64->128 @64", ops.ConvSpec("conv", 64, 128, 3, 2, 1), 64), ("deconv 512->256 @8", ops.ConvSpec("deconv", 512, 256, 4, 2, 1), 8),
                  ("deconv 256->256 @32", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 32), ("hg 3x3 128->128 @64", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 64),
                  ("hg 1x1 256->128 @64", ops.ConvSpec("conv", 256, 128, 1, 1, 0), 64), ("hg 1x1 128->256 @64", ops.ConvSpec("conv", 128, 256, 1, 1, 0), 64)]
        print("AWR_WGRAD_DMA=%s AWR_WGRAD_KP=%s batch %d: TF per (tile/blocks), plain | affine loader" % (os.environ.get("AWR_WGRAD_DMA", "1"), os.environ.get("AWR_WGRAD_KP", "0"), B))
        for name, spec, H in shapes:
            res = []
            pr = spec.wgrad_problem(H, H)
            for tile, blocks in (((1, 1), 2048), ((1, 1), 3072), ((1, 1), 4096), ((2, 1), 1536), ((2, 1), 2048), ((1, 2), 2048), ((2, 2), 768), ((2, 2), 1024), ((2, 2), 1536)):
                if (tile[0] == 2 and pr["Cd"] <= 64) or (tile[1] == 2 and pr["Cg"] <= 64):
                    continue
                res.append("%s/%d %5.1f|%5.1f" % (tile, blocks, run_wgrad(spec, B, H, tile, algo=1, blocks=blocks)[1], run_wgrad(spec, B, H, tile, algo=1, blocks=blocks, affine=True)[1]))
            if spec.kind == "conv" and spec.k == 3 and spec.stride == 1:      # one workgroup per kernel row (algo 3): split-K candidates
                for blocks in (512, 768, 1024, 1536, 2048, 3072):
                    res.append("row/%d %5.1f|%5.1f" % (blocks, run_wgrad(spec, B, H, None, algo=3, blocks=blocks)[1], run_wgrad(spec, B, H, None, algo=3, blocks=blocks, affine=True)[1]))
            print("%-28s %s" % (name, "  ".join(res)), flush=True)
    elif args.mode == "rowset":      # kernel-row weight gradient only (algo 3) over its split-K depths, plain | fused BatchNorm loader; the wave-per-tap
        B = args.batch                # kernel (algo 2) beside it -- same-box A/B of builds through AWR_LIB_PATH (tools/gpu_r4_p.sh)
        shapes = [("layer1 3x3 64->64 @64", ops.ConvSpec("conv", 64, 64, 3, 1, 1), 64), ("hg pre.1 3x3 64->64 @128", ops.ConvSpec("conv", 64, 64, 3, 1, 1), 128),
                  ("layer2 3x3 128->128 @32", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 32), ("hg 3x3 128->128 @64", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 64),
                  ("layer3 3x3 256->256 @16", ops.ConvSpec("conv", 256, 256, 3, 1, 1), 16), ("layer4 3x3 512->512 @8", ops.ConvSpec("conv", 512, 512, 3, 1, 1), 8)]
        print("%s batch %d: TF plain|affine loader" % (os.environ.get("AWR_LIB_PATH", "in-tree"), B))
        for name, spec, H in shapes:
            res = []
            for blocks in (768, 1024, 1280, 1536, 2048, 2560, 3072):
                res.append("row/%d %5.1f|%5.1f" % (blocks, run_wgrad(spec, B, H, None, algo=3, blocks=blocks)[1], run_wgrad(spec, B, H, None, algo=3, blocks=blocks, affine=True)[1]))
            if L.lib.awr_conv_wgrad_algo_ok is not None and spec.cin == 128 and H >= 64:
                res.append("taps %5.1f|%5.1f" % (run_wgrad(spec, B, H, None, algo=2)[1], run_wgrad(spec, B, H, None, algo=2, affine=True)[1]))
            print("%-28s %s" % (name, "  ".join(res)), flush=True)
    elif args.mode == "tiles":       # forward only, every tile, a few representative layers (used by tools/probe_gemm.sh)
        B = args.batch
        shapes = [("layer1 3x3 64->64 @64", ops.ConvSpec("conv", 64, 64, 3, 1, 1), 64), ("layer3 3x3 256->256 @16", ops.ConvSpec("conv", 256, 256, 3, 1, 1), 16),
                  ("deconv 256->256 @32", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 32)]
        for name, spec, H in shapes:
            res = []
            for tile in ((1, 1), (2, 1), (1, 2), (2, 2)):
                if spec.cout <= 64 and tile[1] == 2:
                    continue
                res.append("%s %.0fTF" % (tile, run_fwd(spec, B, H, tile)[1]))
            print("%-26s %s" % (name, "  ".join(res)))
    elif args.mode == "wgrad":       # weight gradient: workgroup-per-tap (algo 1, its tile / split-K candidates) vs wave-per-tap (algo 2)
        B = args.batch
        shapes = [("layer1 3x3 64->64 @64", ops.ConvSpec("conv", 64, 64, 3, 1, 1), 64), ("layer2 3x3 128->128 @32", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 32),
                  ("layer3 3x3 256->256 @16", ops.ConvSpec("conv", 256, 256, 3, 1, 1), 16), ("layer4 3x3 512->512 @8", ops.ConvSpec("conv", 512, 512, 3, 1, 1), 8),
                  ("layer2.0 3x3s2 64->128 @64", ops.ConvSpec("conv", 64, 128, 3, 2, 1), 64), ("layer4.0 3x3s2 256->512 @16", ops.ConvSpec("conv", 256, 512, 3, 2, 1), 16),
                  ("deconv 512->256 @8", ops.ConvSpec("deconv", 512, 256, 4, 2, 1), 8), ("deconv 256->256 @16", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 16),
                  ("deconv 256->256 @32", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 32),
                  ("hg 3x3 128->128 @64", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 64), ("hg 3x3 128->128 @8", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 8)]
        for name, spec, H in shapes:
            old = []
            for tile, blocks in (((1, 1), 2048), ((1, 1), 3072), ((1, 1), 4096), ((2, 1), 1536), ((2, 1), 2048), ((1, 2), 2048)):
                if (tile[0] == 2 and spec.wgrad_problem(H, H)["Cd"] <= 64) or (tile[1] == 2 and spec.wgrad_problem(H, H)["Cg"] <= 64):
                    continue
                t, tf = run_wgrad(spec, B, H, tile, algo=1, blocks=blocks)
                old.append((tf, "%s/%d" % (tile, blocks)))
            new = []
            for blocks in (256, 512, 768, 1024, 1536, 2048, 3072):
                t, tf = run_wgrad(spec, B, H, None, algo=2, blocks=blocks)
                new.append((tf, str(blocks)))
            ta, tfa = run_wgrad(spec, B, H, None, algo=2, blocks=max(new)[1] and int(max(new)[1]), affine=True)
            print("%-30s wg-per-tap best %5.1f TF (%s) | wave-per-tap %s | best %5.1f TF (+affine %5.1f)" % (
                name, max(old)[0], max(old)[1], " ".join("%s:%.0f" % (b, tf) for tf, b in new), max(new)[0], tfa), flush=True)
    elif args.mode == "split":
        B = args.batch
        shapes = [("layer1 3x3 64->64 @64", ops.ConvSpec("conv", 64, 64, 3, 1, 1), 64), ("layer2 3x3 128->128 @32", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 32),
                  ("layer3 3x3 256->256 @16", ops.ConvSpec("conv", 256, 256, 3, 1, 1), 16), ("deconv 256->256 @32", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 32)]
        for name, spec, H in shapes:
            t, tf = run_fwd(spec, B, H)
            res = ["full(auto) %.0fus %.0fTF" % (t * 1e6, tf)]
            tiles = [(1, 1), (2, 1)] + ([(1, 2), (2, 2)] if spec.cout > 64 else [])
            for ta in tiles:
                for tb in tiles:
                    if ta <= tb:
                        t2, tf2 = run_split(spec, B, H, ta, tb)
                        res.append("%s|%s %.0fTF" % (ta, tb, tf2))
            print("%-26s %s" % (name, "  ".join(res)))
    else:
        B = args.batch
        shapes = [("layer1 3x3 64->64 @64", ops.ConvSpec("conv", 64, 64, 3, 1, 1), 64), ("layer2 3x3 128->128 @32", ops.ConvSpec("conv", 128, 128, 3, 1, 1), 32),
                  ("layer3 3x3 256->256 @16", ops.ConvSpec("conv", 256, 256, 3, 1, 1), 16), ("layer4 3x3 512->512 @8", ops.ConvSpec("conv", 512, 512, 3, 1, 1), 8),
                  ("deconv 512->256 @8", ops.ConvSpec("deconv", 512, 256, 4, 2, 1), 8), ("deconv 256->256 @32", ops.ConvSpec("deconv", 256, 256, 4, 2, 1), 32)]
        for name, spec, H in shapes:
            best = []
            for tile in ((2, 2), (2, 1), (1, 2), (1, 1)):
                if spec.cout <= 64 and tile[1] == 2:
                    continue
                t, tf = run_fwd(spec, B, H, tile)
                alt = []
                for v in (6,):
                    L.call("awr_set_gemm_products", v)
                    alt.append("x%d %.0fTF" % (v, run_fwd(spec, B, H, tile)[1]))
                L.call("awr_set_gemm_products", 1)
                tw, tfw = run_wgrad(spec, B, H, tile)
                L.call("awr_set_gemm_products", 6)
                tw6, tfw6 = run_wgrad(spec, B, H, tile)
                L.call("awr_set_gemm_products", 1)
                best.append("%s fwd %.0fus %.0fTF  [%s] | wgrad %.0fus %.0fTF [x6 %.0fTF]" % (tile, t * 1e6, tf, " ".join(alt), tw * 1e6, tfw, tfw6))
            t, tf = run_fwd(spec, B, H)
            print("%-28s auto fwd %.0f TF" % (name, tf))
            for b_ in best:
                print("      " + b_)


if __name__ == "__main__":
    main()
