"""PMC study of the Winograd-domain weight gradient (profiles/r06_winograd.txt): 128 -> 128 @ 64 x 64 x 64 behind the fused affine + ReLU.
Run under rocprofv3 --pmc (tools/gpu_pmc_wino_wgrad.sh)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import awr_amd  # noqa: F401
from awr_amd import _lib as L
dev = torch.device("cuda:0")
B, H, cin, cout = 64, 64, 128, 128
x = torch.randn(B, H, H, cin, device=dev); dy = torch.randn(B, H, H, cout, device=dev)
sc, sh = torch.rand(cin, device=dev) + 0.5, torch.rand(cin, device=dev)
scratch = torch.empty(int(L.lib.awr_wino_wgrad_scratch(B, H, H, cin, cout)), device=dev)
R = torch.empty(cout, 9, cin, device=dev)
for _ in range(12):
    L.call("awr_wino_wgrad", L.ptr(x), L.ptr(dy), L.ptr(sc), L.ptr(sh), 1, B, H, H, cin, cout, L.ptr(scratch), L.ptr(R), cin, None, L.stream())
torch.cuda.synchronize()
print("done")
