"""PMC study of the weight-gradient kernel (profiles/r03_pmc_wgrad.txt): layer1 (64 -> 64 @ 64x64), layer3 (256 -> 256 @ 16x16) 3x3 shapes at batch 64,
with and without the fused input affine (un-materialised BatchNorm + ReLU).  Run under rocprofv3 --pmc (tools/gpu_pmc_wgrad.sh)."""
import ctypes as C, sys, torch
sys.path.insert(0, '/root/repo')
import awr_amd
from awr_amd import ops, _lib as L
dev = torch.device("cuda:0")
B = 64
for cin, cout, H, aff in ((64, 64, 64, False), (64, 64, 64, True), (256, 256, 16, True)):
    spec = ops.ConvSpec("conv", cin, cout, 3, 1, 1)
    x = torch.randn(B, H, H, cin, device=dev); dy = torch.randn(B, H, H, cout, device=dev)
    prob = spec.wgrad_problem(H, H)
    R = torch.zeros(prob["Cd"], len(prob["taps"]), prob["Cg"], device=dev)
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.rand(cin, device=dev)
    a = ops.make_wgrad_args(prob, B, dy, x, R, prob["Cg"], **({"g_affine": (sc, sh, True)} if aff else {}))
    for _ in range(12):
        L.call("awr_conv_wgrad", C.byref(a), L.stream())
    torch.cuda.synchronize()
    print("done", cin, cout, H, aff)
