import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import awr_amd, awr_oracle as O
from awr_amd import _lib as L
dev = torch.device("cuda:0")
g = np.load(os.path.join(REPO, "tests/golden/resnet_18_fwd.npz"))
img = torch.from_numpy(g["img"]); J, ks = int(g["J"]), float(g["ks"])
man = O.manifest_for("resnet_18", J)
fm = awr_amd.FeatureModule()
res = {}
for mode in ("ordered", "auto", "blocked"):
    for fs in (0, 1):
        awr_amd.set_gemm_accum(mode)
        L.call("awr_debug_set_knob", b"fast_stats", fs)
        m = awr_amd.get_deconv_net(18, J, 2); m.load_state_dict(O.procedural_state(man, seed=0)); m = m.cuda(); m.train(True)
        with torch.no_grad():
            o = m(img.to(dev))
        jt = fm.offset2joint_softmax(o, img.to(dev), ks).cpu().numpy()
        d = np.linalg.norm(jt.astype(np.float64) - g["train_s0_jt"].astype(np.float64), axis=-1) * 150
        plan = m.get_plan(2, 128, True)
        res[(mode, fs)] = jt
        print(mode, "fast_stats", fs, "process mode", awr_amd.get_gemm_accum(), "plan.accum", plan.accum, "vs golden %.4e" % d.mean(), flush=True)
awr_amd.set_gemm_accum("auto"); L.call("awr_debug_set_knob", b"fast_stats", 1)
print("ordered==auto (fs1)?", np.array_equal(res[("ordered", 1)], res[("auto", 1)]), "auto==blocked?", np.array_equal(res[("auto", 1)], res[("blocked", 1)]),
      "fs0==fs1 (ordered)?", np.array_equal(res[("ordered", 0)], res[("ordered", 1)]))
