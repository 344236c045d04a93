"""1x1 conv (the Hourglass residual's conv3 / conv1 shapes, batch 64, 64x64 maps: M = 262 144, 17.2 GFLOP) with each epilogue / loader option on its
own, every workgroup tile: where do the 62-83 TF of these launches in the training step come from?  -> profiles/r03_microbench_1x1_epilogue.txt"""
import ctypes as C, torch, time, sys
sys.path.insert(0, '/root/repo')
import awr_amd
from awr_amd import ops, _lib as L
dev = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H = 64, 64
for cin, cout in ((128, 256), (256, 128)):
    spec = ops.ConvSpec("conv", cin, cout, 1, 1, 0)
    x = torch.randn(B, H, H, cin, device=dev); w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    wp = ops.pack_weight(w, spec.fwd_pack())
    res = torch.randn(B, H, H, cout, device=dev)
    y = torch.randn(B, H, H, cout, device=dev)
    coef4 = torch.rand(4, cout, device=dev) + 0.5
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.rand(cin, device=dev)
    out = torch.empty(B, H, H, cout, device=dev)
    prob = spec.fwd_problem(H, H)
    gf = 2.0 * B * H * H * cin * cout * 1e-9
    for tile in ((1, 1), (2, 1), (1, 2), (2, 2)):
        row = []
        for name, kw, extra in (("plain", {}, None), ("affine_in", dict(in_scale=sc, in_shift=sh, relu_in=True), None), ("res", dict(res=res), None),
                                ("stats", dict(stats="S"), None), ("res+stats", dict(res=res, stats="S"), None), ("bnr", dict(stats="S"), "bnr")):
            st = torch.zeros(16, 2, cout, device=dev, dtype=torch.float64)
            kw2 = {k: (st if v == "S" else v) for k, v in kw.items()}
            a = ops.make_conv_args(prob, B, x, wp, out, T=spec.T, **kw2)
            a.tile_m, a.tile_n = tile
            if extra == "bnr":
                a.bnr_y, a.bnr_coef = L.ptr(y), L.ptr(coef4)
            t = bench(lambda: L.call("awr_conv_gemm", C.byref(a), L.stream()))
            row.append("%s %.0f us %.0f TF" % (name, t, gf / t * 1e3))
        print("%d->%d tile %s: " % (cin, cout, tile) + " | ".join(row))
