import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
import awr_amd, awr_oracle as O
dev = torch.device("cuda:0")
J = 14
img, jt_gt = O.synth_batch(2, 128, J, seed=61)
man = O.manifest_for("resnet_18", J)
m = awr_amd.get_deconv_net(18, J, 2); m.load_state_dict(O.procedural_state(man, seed=6)); m = m.cuda(); m.train()
plan = m.get_plan(2, 128, True, supervised=(0,), n_buckets=4)
print("buckets", plan.buckets)
names = plan.op_names("bwd")
for i, n in enumerate(names):
    if n in ("__bucket__", "awr_unpack_wgrads_batched") or "layer3.1.conv2" in n or "layer4.0" in n or "layer4.1.conv1" in n:
        print(i, n)
snaps = []
plan.bucket_hook = lambda lo, hi: snaps.append((lo, hi, m.flat_grads()[lo:hi].clone()))
m.sync_weights(plan, force=True)
plan.img.copy_(img.to(dev))
plan.forward()
plan.grad_outs[0].normal_(0, 1e-3)
plan.backward()
torch.cuda.synchronize()
for lo, hi, s in snaps:
    d = (s != m.flat_grads()[lo:hi])
    print(lo, hi, "mismatch count", int(d.sum()), "first mismatch at", (int(d.nonzero()[0]) + lo) if d.any() else None, "last", (int(d.nonzero()[-1]) + lo) if d.any() else None)
print("nan report")
for k, s, kd in m._layout:
    if kd in ("conv_w", "deconv_w", "conv_b", "bn_w", "bn_b"):
        g = m.grad_view(k)
        n = int(torch.isnan(g).sum())
        if n:
            print(k, tuple(g.shape), "nan", n, "of", g.numel())
print("outputs nan", int(torch.isnan(plan.outputs[0]).sum()))
from awr_amd import _lib as L
print("L.stream()", L.stream(), "default stream handle", torch.cuda.default_stream().cuda_stream)
