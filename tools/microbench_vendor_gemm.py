"""Reference point: the vendor library's FP32 GEMM (torch.mm -> hipBLASLt / rocBLAS, no TF32 on gfx950) on the plain-matrix shapes of
the implicit GEMMs of the ResNet18 step at batch 64 -- what a well-tuned FP32-MFMA GEMM reaches WITHOUT gather, fused prologue /
epilogue or BatchNorm statistics.  Usage: tools/microbench_vendor_gemm.py"""
import torch

dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
shapes = {"layer1 conv (M=262144, N=64, K=576)": (262144, 64, 576), "layer2 conv (65536, 128, 1152)": (65536, 128, 1152),
          "layer3 conv (16384, 256, 2304)": (16384, 256, 2304), "layer4 conv (4096, 512, 4608)": (4096, 512, 4608),
          "deconv 256->256 @32->64, all phases (262144, 256, 1024)": (262144, 256, 1024), "hourglass 3x3 128->128 @64 (262144, 128, 1152)": (262144, 128, 1152),
          "hourglass 1x1 256->128 @64 (262144, 128, 256)": (262144, 128, 256), "square 8192": (8192, 8192, 8192)}
for name, (M, N, K) in shapes.items():
    a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    c = torch.empty(M, N, device=dev)
    for _ in range(3):
        torch.mm(a, b.t(), out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20 if M * N * K < 4e11 else 5
    e0.record()
    for _ in range(reps):
        torch.mm(a, b.t(), out=c)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print("%-62s %8.1f us  %6.1f TFLOP/s" % (name, t * 1e6, 2.0 * M * N * K / t / 1e12))
