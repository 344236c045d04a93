"""Localise a difference between the direct and the Winograd plan of one training step: tensor by tensor in build order."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "tests"))
import awr_amd, awr_oracle as O
from awr_amd.trainer import TrainEngine
dev = torch.device("cuda:0")
net, cw = sys.argv[1] if len(sys.argv) > 1 else "hourglass_1", float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
J, B = 14, 2
ks = 1.0 if net.startswith("resnet") else 0.4
img, jt_gt = O.synth_batch(B, 128, J, seed=23)
sd = O.reference_init_state(net, J, seed=9)
snap = {}
for w in (0, 1, 2):
    awr_amd.set_conv_winograd({0: False, 1: True, 2: "full"}[w])
    if net.startswith("resnet"):
        m = awr_amd.get_deconv_net(int(net.split("_")[1]), J, 2)
    else:
        m = awr_amd.PoseNet(net, J)
    m.load_state_dict(sd); m = m.cuda(); m.train(True)
    eng = TrainEngine(m, B, 128, ks, coord_weight=cw, dense_weight=1.0, lr=1e-3, autotune=False)
    eng.step(img.to(dev), jt_gt.to(dev)); torch.cuda.synchronize()
    T = eng.plan.tensors(lazy=False)
    snap[w] = ({k: (v[0].clone().cpu(), None if v[1] is None else v[1].clone().cpu()) for k, v in T.items()},
               {k: m.grad_view(k).clone().cpu() for k, _ in m.named_parameters() if k not in m._unused}, eng.plan.n_winograd)
    print("mode", w, "winograd launches", eng.plan.n_winograd, flush=True)
def rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))
for w in (1, 2):
    print("== mode", w, "against direct")
    for k, (v, g) in snap[0][0].items():
        rv = rel(snap[w][0][k][0], v)
        rg = rel(snap[w][0][k][1], g) if g is not None else -1.0
        flag = " <<<" if rv > 2e-5 or rg > 2e-5 else ""
        print("  %-44s value %.2e  grad %.2e%s" % (k, rv, rg, flag))
    worst = sorted(((rel(snap[w][1][k], g), k) for k, g in snap[0][1].items()), reverse=True)[:8]
    for r, k in worst:
        print("  param grad %-40s %.2e" % (k, r))
