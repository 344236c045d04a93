"""GPU box: does the drop-in module path (net(x), training mode, no_grad) pick up the process-wide accumulation mode?  Prints a checksum of
the dense map for ordered / blocked plans of the same net and input."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import awr_amd  # noqa: E402
import awr_oracle as O  # noqa: E402
from awr_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
img, _ = O.synth_batch(2, 128, 14, seed=5)
man = O.manifest_for("resnet_18", 14)
for mode in (("env",) if "--env" in sys.argv else ("ordered", "blocked", "ordered")):
    if mode != "env":
        awr_amd.set_gemm_accum(mode)
    m = awr_amd.get_deconv_net(18, 14, 2)
    m.load_state_dict(O.procedural_state(man, seed=0))
    m = m.cuda()
    for train in (True, False):
        m.train(train)
        with torch.no_grad():
            out = m(img.to(dev))
        out = out[-1] if isinstance(out, (list, tuple)) else out
        print(mode, "train" if train else "eval", "process mode", L.lib.awr_get_gemm_accum(), "checksum %.10e" % float(out.double().abs().sum()))

# the golden forward fixture of tests/test_nets_gpu.py::test_backbone_forward_golden[resnet_18], training-mode BatchNorm, both modes
import numpy as np  # noqa: E402
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "resnet_18_fwd.npz"))
gimg = torch.from_numpy(g["img"])
fm = awr_amd.FeatureModule()
for mode in ("ordered", "blocked"):
    awr_amd.set_gemm_accum(mode)
    m = awr_amd.get_deconv_net(18, int(g["J"]), 2)
    m.load_state_dict(O.procedural_state(O.manifest_for("resnet_18", int(g["J"])), seed=0))
    m = m.cuda().train()
    with torch.no_grad():
        o = m(gimg.to(dev))
    o = o[-1] if isinstance(o, (list, tuple)) else o
    jt = fm.offset2joint_softmax(o, gimg.to(dev), float(g["ks"])).cpu().numpy()
    d = np.linalg.norm(jt.astype(np.float64) - g["train_s0_jt"].astype(np.float64), axis=-1) * 150.0
    print("golden fwd fixture, train mode,", mode, ": joints mean %.4e mm, max %.4e mm; per joint (image 0):" % (d.mean(), d.max()), np.array2string(d[0], precision=2))
