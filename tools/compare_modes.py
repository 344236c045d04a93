"""Train the same network on the same synthetic batches in the FP32-MFMA mode and in the split-operand mode and compare the
loss trajectories (both are fp32-accurate per GEMM; Adam amplifies rounding-level differences, so the curves are compared
statistically, and against a third run that only differs by a re-seeded tile autotuning = a different summation order)."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import awr_amd  # noqa: E402
import awr_oracle as O  # noqa: E402  (synthetic inputs only)
from awr_amd.trainer import TrainEngine  # noqa: E402


def run(products, steps, B, autotune, net_name="resnet_18"):
    awr_amd.set_gemm_products(products)
    torch.manual_seed(0)
    net = (awr_amd.get_deconv_net(18, 14, 2) if net_name.startswith("resnet") else awr_amd.PoseNet(net_name, 14)).cuda()
    eng = TrainEngine(net, B, 128, 1.0 if net_name.startswith("resnet") else 0.4, coord_weight=1.0, dense_weight=1.0, lr=1e-3, use_graph=False, autotune=autotune)
    losses = []
    for s in range(steps):
        img, jt = O.synth_batch(B, 128, 14, seed=100 + s % 8)
        l, _ = eng.step(img.cuda(), jt.cuda())
        losses.append(float(l[2]))
    awr_amd.set_gemm_products(1)
    return losses


def main():
    steps, B = int(os.environ.get("STEPS", "120")), 16
    a = run(1, steps, B, False)
    b = run(6, steps, B, False)
    c = run(1, steps, B, True)          # FP32 MFMA with autotuned tiles: same arithmetic, different summation order
    tail = slice(steps - 20, steps)
    mean = lambda v: sum(v) / len(v)    # noqa: E731
    rel = lambda x, y: max(abs(p - q) / q for p, q in zip(x, y))     # noqa: E731
    out = {"steps": steps, "batch": B, "loss_step0": [a[0], b[0], c[0]], "final20_mean_loss": {"f32_mfma": mean(a[tail]), "split6": mean(b[tail]), "f32_mfma_other_tiles": mean(c[tail])},
           "max_rel_diff_first10": {"split6_vs_f32": rel(b[:10], a[:10]), "other_tiles_vs_f32": rel(c[:10], a[:10])},
           "max_rel_diff_all": {"split6_vs_f32": rel(b, a), "other_tiles_vs_f32": rel(c, a)},
           "curves_every10": {"f32_mfma": a[::10], "split6": b[::10], "f32_mfma_other_tiles": c[::10]}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
