"""BASELINE config 5 on one GPU: Hourglass-2, 256x256 input, 21 joints -- step time, plan size, MFMA fraction (and, for B = 2, the
oracle loss on the trained weights).  `--layers` prints the slowest GEMM launches of a serialised pass and the GEMM / step totals.
Usage: tools/check_hg2_256.py [--layers] [batch sizes...]"""
import sys
import time

import torch

sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import awr_amd, awr_oracle as O
from awr_amd.trainer import TrainEngine

dev = torch.device('cuda:0')
J, H = 21, 256
args = [a for a in sys.argv[1:] if not a.startswith("--")]
layers = "--layers" in sys.argv
for B in ([int(b) for b in args] or [2, 8, 16]):
    net = awr_amd.PoseNet('hourglass_2', J).cuda()
    eng = TrainEngine(net, B, H, 0.4, coord_weight=1.0, use_graph=False)
    img, jt = O.synth_batch(B, H, J, seed=3)
    img, jt = img.to(dev), jt.to(dev)
    for _ in range(3): l, _ = eng.step(img, jt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): l, _ = eng.step(img, jt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    macs = sum(eng.plan.macs.values())
    print("HG-2 J=21 256x256 B=%d: %.1f ms/step, %.1f img/s, plan %.1f GB, mfma frac %.3f, loss %.5f" % (B, dt * 1e3, B / dt, eng.plan.bytes / 1e9, 2 * macs / dt / 157.3e12, float(l[2])))
    if layers:
        per = eng.timed_core()
        tot = sum(per.values())
        print("  serial GEMM-family time %.1f ms (%.1f TF average); step %.1f ms" % (tot * 1e3, 2 * macs / tot / 1e12, dt * 1e3))
        for n, t in sorted(per.items(), key=lambda kv: -kv[1])[:40]:
            print("  %-58s %9.1f us %7.1f TF" % (n, t * 1e6, 2 * eng.plan.macs.get(n, 0) / t / 1e12))
    if B == 2:
        sd = net.state_dict()
        o = O.loss_and_grads('hourglass_2', {k: v.cpu().clone() for k, v in sd.items()}, img.cpu(), jt.cpu(), 0.4, 1.0, 1.0, J=J)
        print("  oracle loss on the trained weights (train-mode fwd):", float(o[0]))
    del eng, net; torch.cuda.empty_cache()
