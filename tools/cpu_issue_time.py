import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
import awr_amd, awr_oracle as O
from awr_amd.trainer import TrainEngine
net = awr_amd.get_deconv_net(18, 14, 2).cuda()
eng = TrainEngine(net, 64, 128, 1.0, lr=1e-3, use_graph=False)
img, jt = O.synth_batch(64, 128, 14, seed=1); img, jt = img.cuda(), jt.cuda()
for _ in range(8): eng.step(img, jt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): eng.step(img, jt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("CPU issue time per step: %.2f ms; wall per step: %.2f ms" % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(30): eng.step(img, jt)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr).sort_stats("cumulative")
    st.print_stats(28)
    print("ops: fwd %d bwd %d" % (len(eng.plan.op_names("fwd")), len(eng.plan.op_names("bwd"))))
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(img, jt)
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("issue time of ONE step into an empty queue: min %.2f ms, median %.2f ms" % (min(ts) * 1e3, sorted(ts)[5] * 1e3))
