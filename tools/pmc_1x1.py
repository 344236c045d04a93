"""PMC study of the short-K 1x1 launches (profiles/r03_pmc_1x1.txt): the Hourglass conv3 shape (128 -> 256, 64x64 maps, batch 64) plain and with
residual + statistics, beside the 3x3 128 -> 128 conv of the same residual, 12 launches each.  Run under rocprofv3 --pmc (tools/gpu_pmc_1x1.sh)."""
import ctypes as C, sys, torch
sys.path.insert(0, '/root/repo')
import awr_amd
from awr_amd import ops, _lib as L
dev = torch.device("cuda:0")
B, H = 64, 64
def run(spec, cin, cout, k, res, stats, tile, n=12):
    x = torch.randn(B, H, H, cin, device=dev); w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wp = ops.pack_weight(w, spec.fwd_pack())
    out = torch.empty(B, H, H, cout, device=dev)
    r = torch.randn(B, H, H, cout, device=dev) if res else None
    st = torch.zeros(16, 2, cout, device=dev, dtype=torch.float64) if stats else None
    a = ops.make_conv_args(spec.fwd_problem(H, H), B, x, wp, out, T=spec.T, res=r, stats=st)
    a.tile_m, a.tile_n = tile
    for _ in range(n):
        L.call("awr_conv_gemm", C.byref(a), L.stream())
    torch.cuda.synchronize()
run(ops.ConvSpec("conv", 128, 256, 1, 1, 0), 128, 256, 1, False, False, (2, 2))      # -> conv_gemm_kernel<2, 2, ...>
run(ops.ConvSpec("conv", 128, 256, 1, 1, 0), 128, 256, 1, True, True, (1, 2))        # -> <1, 2, ..., EPRE>
run(ops.ConvSpec("conv", 128, 128, 3, 1, 1), 128, 128, 3, False, True, (2, 1))       # -> <2, 1, ...>
