#!/usr/bin/env python3
"""Histogram of the launches in a training plan's forward / backward lists (GPU): tools/plan_ops.py [net] [batch] [H] [J]"""
import collections
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import awr_amd                      # noqa: E402
from awr_amd.trainer import TrainEngine      # noqa: E402

net_name = sys.argv[1] if len(sys.argv) > 1 else "resnet_18"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
J = int(sys.argv[4]) if len(sys.argv) > 4 else 14
net = (awr_amd.get_deconv_net(int(net_name.split("_")[1]), J, 2) if net_name.startswith("resnet") else awr_amd.PoseNet(net_name, J)).cuda()
eng = TrainEngine(net, B, H, 1.0 if net_name.startswith("resnet") else 0.4, autotune=False)
for which in ("fwd", "bwd"):
    ops = eng.plan.op_names(which)
    h = collections.Counter(o.split(":")[0] for o in ops)
    print(net_name, which, len(ops), "ops:", ", ".join("%s x%d" % kv for kv in sorted(h.items(), key=lambda kv: -kv[1])))
