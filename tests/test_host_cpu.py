"""CPU: host-side logic -- conv geometry / tap tables (emulated gather-GEMM vs torch convs),
weight packing recipes, checkpoint layout and arena plumbing of the drop-in modules."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

import awr_oracle as O


@pytest.fixture(scope="module")
def amd():
    import awr_amd
    return awr_amd


def emulate_pack(w, recipe):
    d0, d1, T, tr, rows, ld = recipe
    w3 = w.reshape(d0, d1, T)
    out = torch.zeros(rows, T, ld, dtype=w.dtype)
    if tr:
        out[:d1, :, :d0] = w3.permute(1, 2, 0)
    else:
        out[:d0, :, :d1] = w3.permute(0, 2, 1)
    return out


def emulate_gemm(prob, x, wp):
    """Pure-torch statement of awr_conv_gemm's contract (include/awr_hip.h): x NHWC, wp [n][T][Cin]."""
    B = x.shape[0]
    out = torch.zeros(B, prob["Hout"], prob["Wout"], prob["N"], dtype=x.dtype)
    xp = x
    for py, px, taps in prob["phases"]:
        for qy in range(prob["Hq"]):
            for qx in range(prob["Wq"]):
                acc = torch.zeros(B, prob["N"], dtype=x.dtype)
                for dy, dx, wt in taps:
                    iy, ix = qy * prob["si"] + dy, qx * prob["si"] + dx
                    if 0 <= iy < prob["Hin"] and 0 <= ix < prob["Win"]:
                        acc += xp[:, iy, ix, :] @ wp[:prob["N"], wt, :].T
                out[:, qy * prob["so"] + py, qx * prob["so"] + px, :] = acc
    return out


CASES = [("conv", 4, 6, 3, 1, 1, 6), ("conv", 4, 6, 3, 2, 1, 8), ("conv", 4, 6, 1, 2, 0, 8), ("conv", 4, 6, 1, 1, 0, 5),
         ("conv", 3, 5, 5, 1, 2, 7), ("deconv", 4, 6, 4, 2, 1, 4)]


@pytest.mark.parametrize("kind,cin,cout,k,s,p,H", CASES)
def test_geometry_forward_and_dgrad(amd, kind, cin, cout, k, s, p, H):
    from awr_amd import ops
    ops_K = ops.K_ALIGN
    ops.K_ALIGN = 1                      # the 32-channel granularity is a kernel constraint, not a geometry one
    try:
        spec = ops.ConvSpec(kind, cin, cout, k, s, p)
    finally:
        ops.K_ALIGN = ops_K
    g = torch.Generator().manual_seed(0)
    w = torch.randn((cout, cin, k, k) if kind == "conv" else (cin, cout, k, k), generator=g, dtype=torch.float64)
    x = torch.randn(2, cin, H, H, generator=g, dtype=torch.float64, requires_grad=True)
    y_ref = (TF.conv2d if kind == "conv" else TF.conv_transpose2d)(x, w, None, s, p)
    gy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    (gx_ref,) = torch.autograd.grad(y_ref, x, gy)
    xn = x.detach().permute(0, 2, 3, 1)
    y = emulate_gemm(spec.fwd_problem(H, H), xn, emulate_pack(w, spec.fwd_pack()))
    assert torch.allclose(y.permute(0, 3, 1, 2), y_ref.detach(), atol=1e-10)
    dp = spec.dgrad_problem(H, H)
    gx = emulate_gemm(dp, gy.permute(0, 2, 3, 1), emulate_pack(w, spec.dgrad_pack()))
    assert torch.allclose(gx.permute(0, 3, 1, 2), gx_ref, atol=1e-10)
    # wgrad contract: R[d0][t][d1] = sum_m D[m][d0] * G[gather(m,t)][d1]
    wp = spec.wgrad_problem(H, H)
    D, G = (gy, x.detach()) if wp["D"] == "dy" else (x.detach(), gy)
    Dn, Gn = D.permute(0, 2, 3, 1), G.permute(0, 2, 3, 1)
    R = torch.zeros(wp["Cd"], len(wp["taps"]), wp["Cg"], dtype=torch.float64)
    for t, (dy, dx) in enumerate(wp["taps"]):
        for yy in range(wp["Hd"]):
            for xx in range(wp["Wd"]):
                gy_, gx_ = yy * wp["sg"] + dy, xx * wp["sg"] + dx
                if 0 <= gy_ < wp["Hg"] and 0 <= gx_ < wp["Wg"]:
                    R[:, t, :] += Dn[:, yy, xx, :].T @ Gn[:, gy_, gx_, :]
    wd = w.clone().requires_grad_(True)
    (gw_ref,) = torch.autograd.grad((TF.conv2d if kind == "conv" else TF.conv_transpose2d)(x.detach(), wd, None, s, p), wd, gy)
    assert torch.allclose(R.permute(0, 2, 1).reshape(gw_ref.shape), gw_ref, atol=1e-9)


def test_scatter_phases_partition_the_taps(amd):
    from awr_amd.ops import ConvSpec
    spec = ConvSpec("deconv", 32, 32, 4, 2, 1)
    ph = spec._scatter_phases()
    assert len(ph) == 4 and all(len(t) == 4 for _, _, t in ph)
    assert sorted(wt for _, _, taps in ph for _, _, wt in taps) == list(range(16))
    spec1 = ConvSpec("conv", 32, 32, 1, 2, 0)
    assert [(py, px, len(t)) for py, px, t in spec1._scatter_phases()] == [(0, 0, 1)]     # only even pixels get gradient
    assert not spec1.dgrad_problem(8, 8)["full"]
    spec3 = ConvSpec("conv", 32, 32, 3, 2, 1)
    assert sorted(len(t) for _, _, t in spec3._scatter_phases()) == [1, 2, 2, 4]


@pytest.mark.parametrize("net,J", [("resnet_18", 14), ("hourglass_1", 14), ("hourglass_2", 21)])
def test_module_checkpoint_layout(amd, golden_dir, net, J):
    man = json.load(open(os.path.join(golden_dir, "statedict_manifest.json")))["%s_J%d" % (net, J)]
    m = amd.get_deconv_net(18, J, 2) if net.startswith("resnet") else amd.PoseNet(net, J)
    sd = m.state_dict()
    assert list(sd.keys()) == [e[0] for e in man]
    assert [list(v.shape) for v in sd.values()] == [e[1] for e in man]
    assert [str(v.dtype).replace("torch.", "") for v in sd.values()] == [e[2] for e in man]
    # strict load of a reference-layout state dict; values land in the flat arena
    ref = O.procedural_state(O.manifest_for(net, J), seed=3)
    m.load_state_dict(ref, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k]), k
    n_params = sum(v.numel() for k, v in ref.items() if v.dtype == torch.float32 and "running" not in k)
    assert sum(p.numel() for p in m.parameters()) == n_params
    assert m.flat_params().numel() >= n_params
    # parameters are views of one arena: writing through the arena is visible in state_dict()
    m.flat_params().mul_(2.0)
    k0 = next(iter(ref))
    assert torch.equal(m.state_dict()[k0], ref[k0] * 2)
    # stock optimiser over net.parameters() round-trips its state (train.py:67, :84, :168)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt.load_state_dict(opt.state_dict())


def test_hourglass_unused_parameters_sit_at_the_arena_tail(amd, golden_dir):
    g = np.load(os.path.join(golden_dir, "hourglass_1_train.npz"))
    m = amd.PoseNet("hourglass_1", 14)
    assert sorted(m._unused) == sorted(str(k) for k in g["nograd"])        # same 30 tensors the reference never trains
    lo = min(m._poff[k][0] for k in m._unused)
    assert lo >= m.n_active and all(m._poff[k][0] < m.n_active for k in m._poff if k not in m._unused)
    assert sum(m._poff[k][1] for k in m._unused) == 986880


def test_reference_init_distributions(amd):
    torch.manual_seed(0)
    m = amd.get_deconv_net(18, 14, 2)
    sd = m.state_dict()
    assert abs(float(sd["layer1.0.conv1.weight"].std()) - (2.0 / (9 * 64)) ** 0.5) < 2e-3     # resnet_deconv.py:95-97
    assert abs(float(sd["deconv_layers.0.weight"].std()) - 0.001) < 1e-4                       # :103-104
    assert float(sd["final1.bias"].abs().max()) == 0 and float(sd["pre.1.weight"].min()) == 1  # :110, :99
    h = amd.PoseNet("hourglass_1", 14).state_dict()
    w = h["pre.1.conv2.conv.weight"]
    assert float(w.abs().max()) <= 1.0 / (w.shape[1] * 9) ** 0.5 + 1e-7                       # kaiming_uniform(a=sqrt(5))


def test_gradient_bucket_planner(amd):
    """Data-parallel overlap: buckets tile the arena from its end downwards, in backward (ready) order."""
    from awr_amd.engine import plan_buckets
    # 6 tensors laid out in forward order; backward finalises them in reverse, with a local swap (offset 300 before 400)
    writes = [(0, 100, 50), (100, 300, 40), (300, 400, 25), (400, 700, 30), (700, 900, 10), (900, 1000, 5)]
    b = plan_buckets(writes, 1000, 3)
    assert b[0][1] == 1000 and b[-1][0] == 0
    assert all(x[0] == y[1] for x, y in zip(b, b[1:]))                 # contiguous, descending
    assert [r for _, _, r in b] == sorted(r for _, _, r in b)          # launch order follows the backward
    for lo, hi, ready in b:
        assert ready >= max(r for (wlo, whi, r) in writes if lo <= wlo < hi)    # nothing is reduced before it is final
    assert plan_buckets(writes, 1000, 1) == [(0, 1000, 50)]
    assert len(plan_buckets(writes, 1000, 16)) <= 6
    # the last bucket -- final only when the backward ends -- is the small one: ResNet18-like sizes (stem + layer1 = 1 % of the arena)
    sizes = [1728, 74000, 74000, 520000, 2100000, 8400000, 2100000, 1050000, 1050000, 14000]
    w2, lo = [], 0
    for i, n in enumerate(sizes):
        w2.append((lo, lo + n, 100 - 10 * i))
        lo += n
    b4 = plan_buckets(w2, lo, 4)
    assert 3 <= len(b4) <= 4 and b4[-1] == (0, 1728 + 74000 + 74000, 100) and b4[0][1] == lo
    assert all(x[0] == y[1] for x, y in zip(b4, b4[1:])) and [r for _, _, r in b4] == sorted(r for _, _, r in b4)
    assert (b4[-1][1] - b4[-1][0]) <= lo // 100


def test_entry_point_overrides_and_dataset_lookup(tmp_path):
    """train.py / test.py entry points: `--set` parsing, config validation, and the reference's behaviour of building the
    NYU datasets from the config (train.py:58-61) -- here: a clear error when the directory is absent."""
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("awr_train_entry", os.path.join(repo, "train.py"))
    entry = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(entry)
    ov = entry.parse_overrides(["net=resnet_18", "kernel_size=1", "cube=[250,250,250]", "lr=5e-4", "exp_id=run_a"])
    assert ov == {"net": "resnet_18", "kernel_size": 1, "cube": [250, 250, 250], "lr": 5e-4, "exp_id": "run_a"}
    from awr_amd.config import Config
    cfg = Config(data_dir=str(tmp_path), dataset="nyu", **ov)
    assert (cfg.jt_num, cfg.step, cfg.max_epoch) == (14, 30, 40) and cfg.net == "resnet_18"
    assert Config(dataset="msra").jt_num == 21 and Config(dataset="icvl", max_epoch=3).max_epoch == 3
    with pytest.raises(AttributeError):
        Config(no_such_entry=1)
    with pytest.raises(ValueError):
        Config(dataset="bighand")
    from awr_amd.trainer import Trainer
    with pytest.raises(FileNotFoundError):
        Trainer(cfg)


def test_visualisation_and_pck_plot_without_cv2_or_matplotlib(tmp_path):
    """util/vis_tool.py:17-60 and util/eval_tool.py:124-135 restated with PIL: files are written, the overlay paints the
    prediction in red and the ground truth in blue at the joint pixels, the PCK curve is a rising line inside the axes."""
    from PIL import Image
    from awr_amd.vis_tool import VisualUtil, plot_pck
    from awr_amd.evaluator import EvalUtil
    rng = np.random.RandomState(0)
    img = np.full((1, 128, 128), 1.0, np.float32)
    img[0, 40:90, 40:90] = -0.2
    pred = np.concatenate([rng.uniform(30, 100, (14, 2)), np.zeros((14, 1))], 1).astype(np.float32)
    gt = pred + np.array([6.0, -5.0, 0.0], np.float32)
    path = os.path.join(str(tmp_path), "overlay.png")
    VisualUtil("nyu").plot(img, path, pred)
    rgb = np.asarray(Image.open(path).convert("RGB")).astype(np.int64)
    assert rgb.shape == (128, 128, 3)
    for j in range(14):
        u, v = int(pred[j, 0]), int(pred[j, 1])
        assert rgb[v, u, 0] > rgb[v, u, 2], j                         # red dominates at predicted joints
    VisualUtil("nyu").plot(img, path, pred, gt)                        # ground truth is painted on top, in blue
    rgb = np.asarray(Image.open(path).convert("RGB")).astype(np.int64)
    for j in range(14):
        u, v = int(gt[j, 0]), int(gt[j, 1])
        assert rgb[v, u, 2] > rgb[v, u, 0], j
    assert tuple(rgb[2, 2]) == (200, 200, 200)                          # background depth 1.0 -> (1 + 1) * 100
    with pytest.raises(ValueError):
        VisualUtil("unknown-set")
    assert len(VisualUtil("hands17").fingers) == 5 and VisualUtil("hands17").fingers[4][0][-1] == 0
    thr = np.linspace(0, 50, 100)
    pck = 1.0 - np.exp(-thr / 10.0)
    p2 = os.path.join(str(tmp_path), "pck.png")
    plot_pck(p2, pck, thr)
    EvalUtil(128, (588.03, 587.07, 320.0, 240.0), -1, 14).plot_pck(os.path.join(str(tmp_path), "pck2.png"), pck, thr)
    chart = np.asarray(Image.open(p2).convert("RGB")).astype(np.int64)
    blue = (chart[:, :, 2] > 150) & (chart[:, :, 0] < 80)
    cols = np.where(blue.any(0))[0]
    first_row = np.array([np.where(blue[:, c])[0].min() for c in cols])
    assert cols.size > 200 and first_row[0] > first_row[-1] + 100      # the curve climbs from the bottom-left to the top-right


def test_bench_rank_affinity_helpers(monkeypatch):
    """bench.py, N > 1: a rank is pinned to the CPUs of its GPU's NUMA node, or -- when /sys reports none -- to an equal contiguous share of
    the allowed CPUs.  Pure host logic: exercised here without a GPU (the device query is stubbed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("awr_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and bench._cpulist("") == set()
    allowed = sorted(os.sched_getaffinity(0))

    class Props:          # a device whose PCI address has no sysfs entry: the equal-share fallback
        pci_bus_id, pci_device_id, pci_domain_id = 0xfe, 0x1f, 0xffff
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda dev: Props())
    pinned = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: pinned.append(set(cpus)))
    nlocal = 2 if len(allowed) >= 2 else 1
    shares = [bench._pin_cpu_affinity(None, r, nlocal) for r in range(nlocal)]
    assert all("error" not in s and s["numa_node"] is None and s["n_cpus"] >= 1 for s in shares), shares
    assert len(pinned) == nlocal and all(p <= set(allowed) for p in pinned)
    if nlocal == 2:
        assert not (pinned[0] & pinned[1])                     # disjoint shares
    # a failing query is reported, never raised (an unpinned rank is slower, not wrong)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda dev: (_ for _ in ()).throw(RuntimeError("no device")))
    assert "error" in bench._pin_cpu_affinity(None, 0, 1)
