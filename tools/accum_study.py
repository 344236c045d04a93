"""Accumulation modes of the forward / data-gradient GEMMs (include/awr_hip.h: awr_set_gemm_accum) side by side: ordered, auto at
several K thresholds, blocked -- distance of the joints from the reference-generated golden and from float64 on the ResNet18
training-mode fixtures, and the cost on the BASELINE batch-64 train step.  GPU box only; prints one JSON object.

    python tools/accum_study.py [--fast-stats]      -> profiles/r06_accum_modes.json (copy by hand)
"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import awr_amd  # noqa: E402
import awr_oracle as O  # noqa: E402
from awr_amd import _lib as L  # noqa: E402
from awr_amd.trainer import TrainEngine  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
dev = torch.device("cuda:0")


def make_net(net, J, sd):
    m = awr_amd.get_deconv_net(int(net.split("_")[1]), J, 2) if net.startswith("resnet") else awr_amd.PoseNet(net, J)
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def fp64_joints(net, sd, img, ks):
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    O.HIGH_PRECISION = True
    try:
        with torch.no_grad():
            o = O.backbone_forward(net, sd64, img.double(), training=True)
            return O.offset2joint_softmax(o[-1], img.double(), ks)
    finally:
        O.HIGH_PRECISION = False


def mm(a, b):
    d = np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=-1) * 150.0
    return float(d.mean()), float(d.max())


def fixtures(mode):
    out = {}
    for net in ("resnet_18", "hourglass_1"):
        g = np.load(os.path.join(GOLD, "%s_train.npz" % net))
        img, jt_gt = torch.from_numpy(g["img"]), torch.from_numpy(g["jt_gt"])
        J, ks = int(g["J"]), float(g["ks"])
        man = O.manifest_for(net, J)
        j64 = fp64_joints(net, O.procedural_state(man, seed=1), img, ks)
        with torch.no_grad():
            o32 = O.backbone_forward(net, O.procedural_state(man, seed=1), img, training=True)
            j32 = O.offset2joint_softmax(o32[-1], img, ks)
        out["%s/oracle_vs_fp64" % net] = mm(j32, j64)
        for tag, cw in (("c0", 0.0), ("c1", 1.0)):
            m = make_net(net, J, O.procedural_state(man, seed=1))
            eng = TrainEngine(m, img.shape[0], 128, ks, coord_weight=cw, dense_weight=1.0, lr=1e-3, use_graph=False, accum=mode)
            _, jt = eng.step(img.to(dev), jt_gt.to(dev))
            jt = jt.cpu().numpy()
            out["%s/%s/vs_golden" % (net, tag)] = mm(jt, g[tag + "_jt0"])
            out["%s/%s/vs_fp64" % (net, tag)] = mm(jt, j64.numpy())
    b8 = os.path.join(GOLD, "resnet_18_train_b8.npz")
    if os.path.exists(b8):
        g = np.load(b8)
        J, ks, B = int(g["J"]), float(g["ks"]), int(g["B"])
        img, jt_gt = O.synth_batch(B, 128, J, seed=int(g["img_seed"]))
        for tag, cw in (("c0", 0.0), ("c1", 1.0)):
            m = make_net("resnet_18", J, O.reference_init_state("resnet_18", J, seed=int(g["w_seed"])))
            eng = TrainEngine(m, B, 128, ks, coord_weight=cw, dense_weight=1.0, lr=1e-3, use_graph=False, accum=mode)
            _, jt = eng.step(img.to(dev), jt_gt.to(dev))
            out["resnet_18_b8/%s/vs_golden" % tag] = mm(jt.cpu().numpy(), g[tag + "_jt0"])
    return out


def step_ms(mode, net_name="resnet_18", B=64, steps=20, warmup=8):
    torch.manual_seed(0)
    ks = 1.0 if net_name.startswith("resnet") else 0.4
    net = make_net(net_name, 14, O.reference_init_state(net_name, 14, seed=0))
    eng = TrainEngine(net, B, 128, ks, coord_weight=0.0, dense_weight=1.0, lr=1e-3, use_graph=False, accum=mode)
    img, jt = O.synth_batch(B, 128, 14, seed=5)
    img, jt = img.to(dev), jt.to(dev)
    for _ in range(warmup):
        eng.step(img, jt)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step(img, jt)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)

    del eng, net
    torch.cuda.empty_cache()
    return best


def main():
    res = {"fast_stats": os.environ.get("AWR_FAST_STATS", "0")}
    modes = [("ordered", None, 0), ("auto", 2304, 0), ("auto", 1152, 0), ("auto", 1024, 0), ("auto", 576, 0), ("auto", 1152, 1), ("blocked", None, 0)]
    for mode, k, dg in modes:
        if k:
            L.call("awr_set_gemm_accum_auto", k, dg)
        name = mode + ("_k%d%s" % (k, "_dgrad" if dg else "") if k else "")
        ent = {"fixtures": fixtures(mode)}
        ent["r18_b64_ms"] = step_ms(mode)
        if "--hg" in sys.argv:
            ent["hg1_b64_ms"] = step_ms(mode, "hourglass_1")
        res[name] = ent
        print(name, json.dumps(ent), flush=True)
    L.call("awr_set_gemm_accum_auto", 1152, 0)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
