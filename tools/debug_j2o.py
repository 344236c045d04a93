#!/usr/bin/env python3
"""Why is the GT dense map (joint2offset) not bit-identical to the torch-CPU oracle?  Compares, on the box this runs on:
the HIP kernel, a numpy float32 restatement that uses only IEEE-correct operations (numpy's sqrt / divide are correctly rounded),
the torch-CPU oracle, and torch's own sqrt against the correctly rounded value."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import awr_oracle as O      # noqa: E402


def main():
    import awr_amd
    B, J, H, ks = 2, 14, 128, 0.4
    img, jt = O.synth_batch(B, H, J, seed=21)
    F = H // 2
    ref = O.joint2offset(jt, img, ks, F).numpy()
    ieee = O.joint2offset_ieee(jt, img, ks, F)
    hip = awr_amd.FeatureModule().joint2offset(jt.cuda(), img.cuda(), ks, F).cpu().numpy()
    x = (np.random.RandomState(0).rand(1000000).astype(np.float32) * 2 + 1e-3)
    exact = np.sqrt(x.astype(np.float64)).astype(np.float32)
    print("cpu:", [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0])
    print("torch.sqrt (CPU) not correctly rounded in %d of 1e6 values; numpy.sqrt in %d" % (int((torch.sqrt(torch.from_numpy(x)).numpy() != exact).sum()), int((np.sqrt(x) != exact).sum())))
    print("elements:", ref.size, " non-zero:", int((ref != 0).sum()))
    print("HIP   vs IEEE numpy restatement: %d differ (max %g)" % (int((hip != ieee).sum()), float(np.abs(hip - ieee).max())))
    print("HIP   vs torch-CPU oracle      : %d differ (max %g)" % (int((hip != ref).sum()), float(np.abs(hip - ref).max())))
    print("IEEE  vs torch-CPU oracle      : %d differ (max %g)" % (int((ieee != ref).sum()), float(np.abs(ieee - ref).max())))
    g = np.load(os.path.join(REPO, "tests", "golden", "j2o_j14_ks04.npz"))
    print("golden keys:", list(g.keys())[:8])


if __name__ == "__main__":
    main()
