"""Statistics epilogue, accumulator-layout form (EM 5, AWR_FAST_STATS=1) against the row-layout form (0) and float64 sums of the stored tensor, over the
launch shapes of the reference nets (GPU box; tools/gpu_session.sh statspaths)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '/root/repo')
import awr_amd
from awr_amd import ops, _lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [("conv", 64, 64, 1, 1, 0, 2, 32), ("conv", 64, 256, 1, 1, 0, 2, 32), ("conv", 256, 64, 1, 1, 0, 2, 32), ("conv", 128, 128, 3, 2, 1, 2, 32),
         ("conv", 256, 512, 1, 2, 0, 2, 32), ("deconv", 256, 256, 4, 2, 1, 2, 8), ("conv", 64, 64, 3, 1, 1, 2, 32), ("conv", 128, 128, 3, 1, 1, 2, 64),
         ("conv", 256, 128, 1, 1, 0, 2, 64), ("conv", 512, 512, 3, 1, 1, 8, 4), ("conv", 64, 128, 3, 2, 1, 4, 16)]
for kind, cin, cout, k, stride, pad, B, H in cases:
    spec = ops.ConvSpec(kind, cin, cout, k, stride, pad)
    x = torch.randn(B, H, H, spec.cin_pad, device=dev) * 0.7 + 0.3
    w = torch.randn(*((cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)), device=dev) * 0.05
    wp = ops.pack_weight(w, spec.fwd_pack())
    prob = spec.fwd_problem(H, H)
    for with_bias in (False, True):
        bias = (torch.randn(prob["N"], device=dev) + 3.0) if with_bias else None
        for tile in ((0, 0), (1, 1), (2, 1), (1, 2), (2, 2)):
            if tile[1] == 2 and prob["N"] <= 64:
                continue
            res = {}
            for fast in ("0", "1"):
                os.environ["AWR_FAST_STATS"] = fast
                out = torch.full((B, prob["Hout"], prob["Wout"], prob["N"]), float("nan"), device=dev)
                st = torch.zeros(16, 2, prob["N"], device=dev, dtype=torch.float64)
                a = ops.make_conv_args(prob, B, x, wp, out, bias=bias, stats=st, T=spec.T)
                L.call("awr_debug_force_tile", *tile)
                L.call("awr_conv_gemm", C.byref(a), L.stream())
                L.call("awr_debug_force_tile", 0, 0)
                torch.cuda.synchronize()
                res[fast] = (out.clone(), st.sum(0).clone())
            ref = res["0"][0].double().reshape(-1, prob["N"])
            n = ref.shape[0]
            line = "%-6s %3d->%3d k%d s%d B%d H%2d bias=%d tile=%s same_out=%d" % (kind, cin, cout, k, stride, B, H, with_bias, tile, torch.equal(res["0"][0], res["1"][0]))
            for f in ("0", "1"):
                s1, s2 = res[f][1]
                mean, var = s1 / n, s2 / n - (s1 / n) ** 2
                line += "  | fast=%s mean %.1e var %.1e" % (f, float(((mean - ref.mean(0)).abs() / ref.std(0)).max()), float(((var - ref.var(0, unbiased=False)).abs() / ref.var(0, unbiased=False)).max()))
            print(line, flush=True)
