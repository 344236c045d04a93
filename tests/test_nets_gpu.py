"""GPU parity of the full backbones and the fused train step (HIP path through the C ABI) against
the oracle and the committed golden vectors (procedural weights, so nothing large is shipped)."""
import json
import os

import numpy as np
import pytest
import torch

import awr_oracle as O

pytestmark = pytest.mark.gpu


def _has_study():
    import awr_amd  # noqa: F401
    from awr_amd import _lib
    return _lib.HAS_STUDY


# measured-and-rejected forms of earlier rounds live in study builds only (AWR_BUILD_STUDY=1 python -m awr_amd.build --force): their tests run there
study_only = pytest.mark.skipif(not _has_study(), reason="study form: needs libawr_hip.so built with -DAWR_STUDY")
REPORT = {}


def report(name, value):
    REPORT[name] = float(value)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(REPORT, open(os.path.join(out, "parity_report.json"), "w"), indent=1, sort_keys=True)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def amd():
    import awr_amd
    return awr_amd


def make_net(amd, net, J, sd):
    m = amd.get_deconv_net(int(net.split("_")[1]), J, 2) if net.startswith("resnet") else amd.PoseNet(net, J)
    m.load_state_dict(sd, strict=True)
    return m.cuda()


NORTH_STAR_MEAN_MM = 1e-3      # BASELINE.json north_star: "outputs match the reference within 1e-3 mm mean joint error"


_GAP_CACHE = {}     # (net, mode, inputs) -> gaps: the same fixture is scored by the direct, split-operand and Winograd modes; the float64 oracle runs once


def oracle_fp64_joint_gap(net, sd, img, ks, training, stages=None):
    key = (net, bool(training), float(ks), tuple(img.shape), float(img.double().sum()), float(sum(v.double().abs().sum() for v in sd.values() if v.is_floating_point())))
    if key not in _GAP_CACHE:
        _GAP_CACHE[key] = _oracle_fp64_joint_gap(net, sd, img, ks, training)
    return _GAP_CACHE[key]


def _oracle_fp64_joint_gap(net, sd, img, ks, training):
    """Per stage: (mean, max) 3D joint distance in mm (300 mm cube => x150) between the fp32 oracle and the SAME formulas
    evaluated in float64 -- how far fp32 arithmetic alone (torch-CPU, the reference's own numerics) sits from the exact
    answer on these inputs.  The head turns the heat map into softmax(30*h) weights; procedural weights that drive |h| to
    ~15 (Hourglass-2 stage 1) make the joints ill-conditioned, and no fp32 implementation can then agree with another one
    to better than this gap.  Used as the yardstick wherever it exceeds the north_star figure."""
    sd32 = {k: v.clone() for k, v in sd.items()}
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    with torch.no_grad():
        o32 = O.backbone_forward(net, sd32, img, training=training)
        j32 = [O.offset2joint_softmax(o, img, ks) for o in o32]
        O.HIGH_PRECISION = True
        try:
            o64 = O.backbone_forward(net, sd64, img.double(), training=training)
            j64 = [O.offset2joint_softmax(o, img.double(), ks) for o in o64]
        finally:
            O.HIGH_PRECISION = False
    gaps = []
    for a, b in zip(j32, j64):
        d = (a.double() - b).norm(dim=-1) * 150.0
        gaps.append((float(d.mean()), float(d.max()), b))          # [2] = the float64 joints (assert_joints reports HIP's own distance from them)
    return gaps


def assert_joints(name, got, ref, gap, factor=2.0, yardstick=None):
    """got / ref: (B,J,3) normalised joints.  Bar: mean 3D distance <= 1e-3 mm (north_star), max <= 5e-3 mm -- widened to `factor` x
    (mean) / 3 `factor` x (max: the worst single joint of a handful is a noisy statistic) the oracle's own fp32-vs-fp64 gap where the
    inputs are that ill-conditioned (Hourglass-2 stage 1, ResNet-50: the fp32 oracle itself sits beyond the north_star figure from float64
    there, and two fp32 implementations that each sit one gap from the exact answer differ by ~1.4 gaps).

    yardstick = r (round 6) -- the two-image training-mode fixtures with procedural weights are CHAOTIC around 1e-3 mm: where
    an implementation lands against another fp32 implementation (the golden joints) depends on which way a handful of roundings fall in the
    BatchNorm statistics, and a MORE accurate summation can land farther away.  For those the criterion is distance to the TRUTH: the HIP
    joints must sit within r x the fp32 oracle's own distance from float64 (both computed on the spot; mean, and 3 r x for the worst single
    joint); the distance to the golden joints is reported, not asserted.  The well-conditioned fixture
    (test_well_conditioned_training_fixture_meets_the_plain_bar_in_every_mode) carries the plain north_star bar against the reference."""
    d = np.linalg.norm(np.asarray(got, np.float64) - np.asarray(ref, np.float64), axis=-1) * 150.0
    mean, mx = float(d.mean()), float(d.max())
    report(name + "/joint_err_mm_mean", mean)
    report(name + "/joint_err_mm", mx)
    report(name + "/oracle_fp32_vs_fp64_gap_mm_mean", gap[0])
    hip64 = None
    if len(gap) > 2 and tuple(gap[2].shape) == tuple(np.asarray(got).shape):      # how far the HIP joints themselves sit from float64
        d64 = np.linalg.norm(np.asarray(got, np.float64) - gap[2].numpy(), axis=-1) * 150.0
        hip64 = float(d64.mean())
        report(name + "/hip_vs_fp64_mm_mean", hip64)
        report(name + "/hip_vs_fp64_over_oracle_vs_fp64", hip64 / gap[0])
    if yardstick is not None:
        assert hip64 is not None, name
        assert hip64 <= yardstick * gap[0] and float(d64.max()) <= 3.0 * yardstick * gap[1], \
            (name, "HIP vs float64", hip64, float(d64.max()), "oracle vs float64", gap[0], gap[1], "allowed ratio", yardstick)
        return mean, mx
    bar_mean, bar_max = max(NORTH_STAR_MEAN_MM, factor * gap[0]), max(5e-3, 3.0 * factor * gap[1])
    assert mean <= bar_mean and mx <= bar_max, (name, mean, mx, "bars", bar_mean, bar_max, "fp64 gap", gap)
    return mean, mx


def smp_index_stream(n, stream):
    return int(np.minimum(((O._hash_uniform(1, 1000 + stream, 7).astype(np.float64) + 0.5) * n).astype(np.int64), n - 1)[0])


def smp_index(n, i):
    return smp_index_stream(n, 40 + i)


@pytest.mark.parametrize("net", ["resnet_18", "resnet_50", "hourglass_1", "hourglass_2"])
def test_backbone_forward_golden(amd, dev, golden_dir, net, yardstick=2.0):
    g = np.load(os.path.join(golden_dir, "%s_fwd.npz" % net))
    img = torch.from_numpy(g["img"])
    J, ks = int(g["J"]), float(g["ks"])
    man = O.manifest_for(net, J)
    fm = amd.FeatureModule()
    for mode in ("eval", "train"):
        sd = O.procedural_state(man, seed=0)
        m = make_net(amd, net, J, sd)
        m.train(mode == "train")
        with torch.no_grad():
            outs = m(img.to(dev))
        outs = outs if isinstance(outs, list) else [outs]
        okey = ("fwd", net, mode)
        if okey not in _GAP_CACHE:
            _GAP_CACHE[okey] = O.backbone_forward(net, O.procedural_state(man, seed=0), img, training=(mode == "train"))
        oracle = _GAP_CACHE[okey]
        gaps = oracle_fp64_joint_gap(net, O.procedural_state(man, seed=0), img, ks, mode == "train")
        for s, o in enumerate(outs):
            ref = g["%s_s%d_val" % (mode, s)]
            got = o.cpu().reshape(-1).numpy()[g["%s_s%d_idx" % (mode, s)]]
            scale = max(1.0, float(np.abs(ref).max()))
            err = float(np.abs(got - ref).max()) / scale
            full = float((o.cpu() - oracle[s]).abs().max()) / scale
            report("%s/%s/stage%d/dense_map_rel_err" % (net, mode, s), full)
            assert err <= 2e-4 and full <= 2e-4, (err, full)
            jt = fm.offset2joint_softmax(o, img.to(dev), ks).cpu().numpy()
            # ResNet-50 (not a BASELINE config; procedural weights, batch 2, training-mode BatchNorm over 32 samples per channel at layer4) is
            # ill-conditioned -- the fp32 oracle itself sits 4.4e-3 mm from float64 -- and 50 layers deep: the MFMA's k-ordered accumulation
            # carries 2.6x the oracle's rounding error, i.e. an expected distance of sqrt(1 + 2.6^2) = 2.8 gaps, measured 3.06
            # ResNet18, training mode: the chaotic two-image fixture -> distance to float64.  The default accumulation mode blocks the training
            # forward's long K extents; measured (profiles/r06_accum_modes.txt) it lands at 0.87x (train-step fixture) and 1.69x (this
            # fixture) the oracle's own distance from float64 -- two fp32 implementations each one gap from the truth, 28 joints: the ratio
            # itself is a noisy statistic, 2.0 is its bar (the FULLY blocked mode gives the same two numbers: forward launches are identical)
            assert_joints("%s/%s/stage%d" % (net, mode, s), jt, g["%s_s%d_jt" % (mode, s)], gaps[s],
                          factor=4.0 if net == "resnet_50" else 2.0, yardstick=yardstick if (net == "resnet_18" and mode == "train") else None)
        if mode == "train":
            got_sd = m.state_dict()
            for i, k in enumerate(g["bn_keys"]):
                np.testing.assert_allclose(got_sd[str(k)].cpu().numpy(), g["bn_%d" % i], rtol=2e-4, atol=2e-5)
            assert int(got_sd["pre.1.num_batches_tracked" if net.startswith("resnet") else "pre.0.bn.num_batches_tracked"]) == 1
            assert len(got_sd) == len(man)


def check_grad_norms(m, pkeys, ref_l2, ref_smp=None, tol=5e-3):
    """L2 norm per parameter tensor (+ one sampled element each) against the reference's autograd.  A conv bias that feeds a
    BatchNorm has an analytically ZERO gradient (BN removes the mean); both sides then hold rounding noise, so errors are
    measured against the largest gradient norm in the network."""
    worst, worst_key = 0.0, ""
    gmax = float(np.max(ref_l2))
    for i, k in enumerate(pkeys):
        ref = float(ref_l2[i])
        if ref < 0:
            assert k in m._unused
            continue
        gv = m.grad_view(k).cpu()
        if k.endswith(".conv.bias") and i + 1 < len(pkeys) and ".bn" in pkeys[i + 1] and ref <= 1e-3 * gmax:
            assert float(gv.double().norm()) <= 1e-3 * gmax, k      # conv bias feeding a BatchNorm: true gradient is zero
            continue
        err = abs(float(gv.double().norm()) - ref)
        rel = err / (ref + 1e-3 * gmax)
        if rel > worst:
            worst, worst_key = rel, k
        assert rel <= tol, (k, float(gv.double().norm()), ref)
        if ref_smp is not None:
            smp = float(gv.reshape(-1)[smp_index(gv.numel(), i)])
            assert abs(smp - float(ref_smp[i])) <= tol * float(gv.abs().max()) + 1e-6 * gmax, k
    print("worst grad-norm error: %s %.3e" % (worst_key, worst))
    return worst


@pytest.mark.parametrize("net", ["resnet_18", "resnet_50", "hourglass_1", "hourglass_2"])
@pytest.mark.parametrize("tag,cw", [("c0", 0.0), ("c1", 1.0)])
def test_fused_train_step_golden(amd, dev, golden_dir, net, tag, cw, yardstick=2.0):
    from awr_amd.trainer import TrainEngine
    g = np.load(os.path.join(golden_dir, "%s_train.npz" % net))
    img, jt_gt = torch.from_numpy(g["img"]), torch.from_numpy(g["jt_gt"])
    J, ks = int(g["J"]), float(g["ks"])
    man = O.manifest_for(net, J)
    pkeys = [str(k) for k in g["pkeys"]]
    sd = O.procedural_state(man, seed=1)
    m = make_net(amd, net, J, sd)
    eng = TrainEngine(m, img.shape[0], 128, ks, coord_weight=cw, dense_weight=1.0, lr=1e-3, use_graph=False)
    losses, jt = eng.step(img.to(dev), jt_gt.to(dev))
    l0 = float(losses[2])
    ref0 = float(g[tag + "_loss0"])
    report("%s/%s/loss0_rel_err" % (net, tag), abs(l0 - ref0) / abs(ref0))
    assert abs(l0 - ref0) <= 2e-4 * abs(ref0), (l0, ref0)
    assert abs(float(losses[0]) - float(g[tag + "_lcoord0"])) <= 2e-4 * max(1e-6, abs(float(g[tag + "_lcoord0"]))) + 1e-9
    gap = oracle_fp64_joint_gap(net, O.procedural_state(man, seed=1), img, ks, True)[-1]
    assert_joints("%s/%s/train" % (net, tag), jt.cpu().numpy(), g[tag + "_jt0"], gap, factor=4.0 if net == "resnet_50" else 2.0,
                  yardstick=yardstick if net == "resnet_18" else None)
    # gradients: golden = reference autograd (train.py:116-121: for the hourglass only the LAST stage's loss survives)
    # (ResNet-50: 50 layers of ReLU / pooling decisions between the loss and the stem, procedural weights, batch 2 -- the first layers'
    # gradient NORMS move by 1-2 % when a handful of decisions fall the other way; the tensor-by-tensor float64 yardstick below is the sharp
    # statement for it, test_gradients_elementwise_against_the_fp64_yardstick[resnet_50])
    worst = check_grad_norms(m, pkeys, g[tag + "_grad_l2"], g[tag + "_grad_smp"], tol=3e-2 if net == "resnet_50" else 5e-3)
    report("%s/%s/worst_grad_norm_rel_err" % (net, tag), worst)
    if not net.startswith("resnet"):                       # fused multi-stack step: `stacks` BN momentum updates per iteration
        stacks = int(net.split("_")[-1])
        assert all(int(v) == stacks for k, v in m.state_dict().items() if k.endswith("num_batches_tracked"))
    # parameters after 1 and 2 Adam steps (sampled), loss of the second step
    sd1 = m.state_dict()
    p1 = np.array([float(sd1[k].reshape(-1)[smp_index(sd1[k].numel(), i)]) for i, k in enumerate(pkeys)], np.float32)
    report("%s/%s/param_abs_err_step1" % (net, tag), float(np.abs(p1 - g[tag + "_param_smp1"]).max()))
    # one Adam step moves a weight by ~lr = 1e-3 whatever its gradient's magnitude: elements whose gradient is
    # rounding noise can land anywhere in +-lr, the bulk must agree tightly
    d1 = np.abs(p1 - g[tag + "_param_smp1"])
    assert np.quantile(d1, 0.9) <= 1e-4 and d1.max() <= 2.1e-3, (np.quantile(d1, 0.9), d1.max())
    losses, _ = eng.step(img.to(dev), jt_gt.to(dev))
    ref1 = float(g[tag + "_loss1"])
    report("%s/%s/loss1_rel_err" % (net, tag), abs(float(losses[2]) - ref1) / abs(ref1))
    assert abs(float(losses[2]) - ref1) <= 2e-2 * abs(ref1)
    sd2 = m.state_dict()
    p2 = np.array([float(sd2[k].reshape(-1)[smp_index(sd2[k].numel(), i)]) for i, k in enumerate(pkeys)], np.float32)
    # Adam's first steps move every weight by ~lr whatever the gradient magnitude, so elements whose gradient is
    # rounding noise may flip direction: bound the bulk tightly and the worst case by 2 steps * lr.
    d2 = np.abs(p2 - g[tag + "_param_smp2"])
    assert np.quantile(d2, 0.9) <= (1e-3 if net == "resnet_50" else 3e-4) and d2.max() <= 2.1e-3, (np.quantile(d2, 0.9), d2.max())
    for k in m._unused:                                                               # never touched, like torch's `grad is None`
        assert torch.equal(sd2[k].cpu(), O.procedural_state(man, seed=1)[k])


@pytest.mark.parametrize("tag,cw", [("c0", 0.0), ("c1", 1.0)])
def test_blocked_accumulation_meets_the_plain_north_star_bar(amd, dev, golden_dir, tag, cw):
    """VERDICT r3 item 2.  The ordered mode's k-chain (one fmaf chain over K = 576 ... 4608 terms) leaves the ResNet18 training-mode golden
    fixtures at 0.9-1.05e-3 mm -- at or above the north_star figure, passed only through the widened bar.  With blocked accumulation
    (awr_conv_args.accum = 1: the chain restarts every 128 k into a second accumulator set; TrainEngine(accum="blocked")) the SAME fixtures
    meet mean <= 1e-3 mm with NO widening, and the loss / gradient-norm bars of the ordered mode."""
    from awr_amd.trainer import TrainEngine
    net = "resnet_18"
    g = np.load(os.path.join(golden_dir, "%s_train.npz" % net))
    img, jt_gt = torch.from_numpy(g["img"]), torch.from_numpy(g["jt_gt"])
    J, ks = int(g["J"]), float(g["ks"])
    man = O.manifest_for(net, J)
    m = make_net(amd, net, J, O.procedural_state(man, seed=1))
    eng = TrainEngine(m, img.shape[0], 128, ks, coord_weight=cw, dense_weight=1.0, lr=1e-3, use_graph=False, accum="blocked")
    losses, jt = eng.step(img.to(dev), jt_gt.to(dev))
    ref0 = float(g[tag + "_loss0"])
    assert abs(float(losses[2]) - ref0) <= 2e-4 * abs(ref0)
    d = np.linalg.norm(jt.cpu().numpy().astype(np.float64) - g[tag + "_jt0"].astype(np.float64), axis=-1) * 150.0
    report("%s/%s/train_blocked/joint_err_mm_mean" % (net, tag), float(d.mean()))
    report("%s/%s/train_blocked/joint_err_mm" % (net, tag), float(d.max()))
    assert float(d.mean()) <= NORTH_STAR_MEAN_MM and float(d.max()) <= 5e-3, (float(d.mean()), float(d.max()))
    worst = check_grad_norms(m, [str(k) for k in g["pkeys"]], g[tag + "_grad_l2"], g[tag + "_grad_smp"], tol=5e-3)
    report("%s/%s/train_blocked/worst_grad_norm_rel_err" % (net, tag), worst)
    assert amd.get_gemm_accum() == "auto"             # the engine's mode is its plan's, not the process's (whose default is auto)
    if tag == "c0":      # the forward fixture through the drop-in module (training-mode BatchNorm), process-wide mode
        gf = np.load(os.path.join(golden_dir, "resnet_18_fwd.npz"))
        amd.set_gemm_accum("blocked")
        try:
            mf = make_net(amd, net, int(gf["J"]), O.procedural_state(O.manifest_for(net, int(gf["J"])), seed=0))
            mf.train()
            with torch.no_grad():
                o = mf(torch.from_numpy(gf["img"]).to(dev))
            o = o[-1] if isinstance(o, (list, tuple)) else o
            jf = amd.FeatureModule().offset2joint_softmax(o, torch.from_numpy(gf["img"]).to(dev), float(gf["ks"])).cpu().numpy()
        finally:
            amd.set_gemm_accum("auto")
        df = np.linalg.norm(jf.astype(np.float64) - gf["train_s0_jt"].astype(np.float64), axis=-1) * 150.0
        report("%s/train_blocked/stage0/joint_err_mm_mean" % net, float(df.mean()))
        assert float(df.mean()) <= NORTH_STAR_MEAN_MM and float(df.max()) <= 5e-3, (float(df.mean()), float(df.max()))


@pytest.mark.parametrize("mode", ["auto", "ordered", "blocked"])
def test_well_conditioned_training_fixture_meets_the_plain_bar_in_every_mode(amd, dev, golden_dir, mode):
    """VERDICT r5 2b.  tests/golden/resnet_18_train_b8.npz: the REFERENCE's ResNet18-deconv (model/resnet_deconv.py:118-136) in training
    mode on a seeded batch of 8 with weights drawn from its own init_weights distributions (:93-115), one reference train iteration
    (train.py:107-131) for both loss weightings -- joints, sampled dense map, losses, per-tensor gradient norms, BatchNorm running statistics.
    Well-conditioned (256 samples per channel and more in every BatchNorm), so unlike the two-image procedural-weight fixtures nothing here
    depends on which way a handful of roundings fall: the PLAIN north_star bar (1e-3 mm mean, 5e-3 mm max) holds in every accumulation mode."""
    from awr_amd.trainer import TrainEngine
    g = np.load(os.path.join(golden_dir, "resnet_18_train_b8.npz"))
    net, J, ks, B = "resnet_18", int(g["J"]), float(g["ks"]), int(g["B"])
    img, jt_gt = O.synth_batch(B, 128, J, seed=int(g["img_seed"]))
    assert float(img.double().sum()) == float(g["img_sum"]), "synth_batch no longer reproduces the fixture's inputs"
    sd0 = O.reference_init_state(net, J, seed=int(g["w_seed"]))
    assert sum(float(v.double().abs().sum()) for v in sd0.values()) == float(g["w_sum"]), "reference_init_state no longer reproduces the fixture's weights"
    pkeys = [str(k) for k in g["pkeys"]]
    for tag, cw in (("c0", 0.0), ("c1", 1.0)):
        m = make_net(amd, net, J, O.reference_init_state(net, J, seed=int(g["w_seed"])))
        eng = TrainEngine(m, B, 128, ks, coord_weight=cw, dense_weight=1.0, lr=1e-3, use_graph=False, accum=mode)
        assert eng.plan.accum == {"ordered": 0, "blocked": 1, "auto": 2}[mode]
        losses, jt = eng.step(img.to(dev), jt_gt.to(dev))
        ref0 = float(g[tag + "_loss0"])
        assert abs(float(losses[2]) - ref0) <= 2e-4 * abs(ref0), (float(losses[2]), ref0)
        assert abs(float(losses[0]) - float(g[tag + "_lcoord0"])) <= 2e-4 * max(1e-6, abs(float(g[tag + "_lcoord0"]))) + 1e-9
        d = np.linalg.norm(jt.cpu().numpy().astype(np.float64) - g[tag + "_jt0"].astype(np.float64), axis=-1) * 150.0
        report("resnet_18_b8/%s/%s/joint_err_mm_mean" % (tag, mode), float(d.mean()))
        report("resnet_18_b8/%s/%s/joint_err_mm" % (tag, mode), float(d.max()))
        assert float(d.mean()) <= NORTH_STAR_MEAN_MM and float(d.max()) <= 5e-3, (mode, tag, float(d.mean()), float(d.max()))
        pred = eng.plan.dense_map(0).cpu().reshape(-1).numpy()[g["pred_idx"]]
        ref = g[tag + "_pred_val"]
        assert float(np.abs(pred - ref).max()) <= 2e-4 * max(1.0, float(np.abs(ref).max())), mode
        worst = check_grad_norms(m, pkeys, g[tag + "_grad_l2"], g[tag + "_grad_smp"], tol=5e-3)
        report("resnet_18_b8/%s/%s/worst_grad_norm_rel_err" % (tag, mode), worst)
    sd = m.state_dict()
    got = np.array([float(sd[str(k)].reshape(-1)[smp_index_stream(sd[str(k)].numel(), 90 + i)]) for i, k in enumerate(g["bn_keys"])], np.float32)
    np.testing.assert_allclose(got, g["bn_smp"], rtol=2e-4, atol=2e-6)


def test_inference_engine_parity_mode(amd, dev, golden_dir):
    """InferEngine(parity=True): blocked accumulation for scoring passes (test.py:67-86).  Low batches run split-K (already blocked by
    construction), so the mode is exercised at a batch that fills the chip; it must stay within the ordinary bars of the ordered engine
    and may not be farther from float64 than the ordered mode by more than rounding noise."""
    from awr_amd.trainer import InferEngine
    J, ks, B = 14, 1.0, 64
    img, _ = O.synth_batch(B, 128, J, seed=41)
    sd = O.reference_init_state("resnet_18", J, seed=12)
    m = make_net(amd, "resnet_18", J, sd)
    m.eval()
    jo = InferEngine(m, B, 128, ks)(img.to(dev)).cpu()
    jb = InferEngine(m, B, 128, ks, parity=True)(img.to(dev)).cpu()
    with torch.no_grad():
        ref = O.offset2joint_softmax(O.backbone_forward("resnet_18", sd, img[:8], training=False)[-1], img[:8], ks)
    do = float((jo[:8] - ref).norm(dim=-1).mean()) * 150.0
    db = float((jb[:8] - ref).norm(dim=-1).mean()) * 150.0
    report("resnet_18/infer_b64_ordered/joint_err_mm_mean", do)
    report("resnet_18/infer_b64_blocked/joint_err_mm_mean", db)
    assert do <= NORTH_STAR_MEAN_MM and db <= NORTH_STAR_MEAN_MM, (do, db)
    assert float((jo - jb).abs().max()) > 0.0          # (the two plans really are different instantiations)


@pytest.mark.parametrize("net,B", [("resnet_18", 3), ("hourglass_1", 2), ("hourglass_2", 1)])
def test_dropin_autograd_path_vs_oracle(amd, dev, net, B):
    """The reference's own step, written with the drop-in objects (train.py:113-131): net(x) ->
    FeatureModule -> My_SmoothL1Loss -> loss.backward() -> stock torch.optim.Adam."""
    J = 14
    ks = 1.0 if net.startswith("resnet") else 0.4
    img, jt_gt = O.synth_batch(B, 128, J, seed=31)
    man = O.manifest_for(net, J)
    sd = O.procedural_state(man, seed=2)
    m = make_net(amd, net, J, sd)
    m.train()
    fm, crit = amd.FeatureModule(), amd.My_SmoothL1Loss().cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    x, jg = img.to(dev), jt_gt.to(dev)
    gt = fm.joint2offset(jg, x, ks, 64)
    stacks = 1 if net.startswith("resnet") else int(net.split("_")[-1])
    for stage in range(stacks):
        pred = m(x)
        pred = pred[stage] if isinstance(pred, list) else pred
        jt = fm.offset2joint_softmax(pred, x, ks)
        loss = 1.0 * crit(jt, jg) + 1.0 * crit(pred, gt)
    opt.zero_grad()
    loss.backward()
    sdo = O.procedural_state(man, seed=2)
    lo, lc, ld, grads, jt_o = O.loss_and_grads(net, sdo, img, jt_gt, ks, 1.0, 1.0)
    assert abs(float(loss) - float(lo)) <= 2e-4 * abs(float(lo))
    named = dict(m.named_parameters())
    # Per-element gradient parity is not a meaningful bar for this loss: ReLU masks, the Huber kink at 0.01 and
    # the GT-map thresholds make the gradient discontinuous -- perturbing the ORACLE's weights by 2e-6 (fp32
    # rounding level) moves single gradient elements by up to 14 % (DESIGN.md "conditioning").  The bar is the
    # L2 error per parameter tensor and of the whole flat gradient, relative to the network's gradient norm.
    worst, worst_key, num, den = 0.0, "", 0.0, 0.0
    gnorm = sum(float(gr.double().pow(2).sum()) for gr in grads.values() if gr is not None) ** 0.5
    for k, gr in grads.items():
        if gr is None:
            assert named[k].grad is None, k
            continue
        got = named[k].grad.cpu().double()
        e = float((got - gr.double()).norm())
        num += e * e
        d = e / (float(gr.double().norm()) + 1e-2 * gnorm)
        if d > worst:
            worst, worst_key = d, k
    glob = num ** 0.5 / gnorm
    print("worst per-tensor L2 gradient error: %s %.3e ; whole-gradient rel L2 error %.3e" % (worst_key, worst, glob))
    report("%s/dropin/flat_grad_rel_l2_err" % net, glob)
    assert glob <= 2e-2, glob
    report("%s/dropin/worst_grad_rel_err" % net, worst)
    assert worst <= 1e-1, (worst_key, worst)
    opt.step()
    # BN buffers followed the reference quirk: `stacks` momentum updates per iteration
    got = m.state_dict()
    bnk = [k for k, _, kind in man if kind in ("bn_mean", "bn_var")]
    for k in (bnk[0], bnk[-1]):
        np.testing.assert_allclose(got[k].cpu().numpy(), sdo[k].numpy(), rtol=5e-4, atol=5e-5)
    cnt = [k for k, _, kind in man if kind == "counter"][0]
    assert int(got[cnt]) == stacks


def test_config5_hourglass2_256_j21(amd, dev, golden_dir):
    """BASELINE configs[4]: Hourglass-2, 256x256 crops, 21 joints.  Forward (eval) and the FUSED two-stack train step -- one
    forward with the BatchNorm momentum compounded twice and only the last stage supervised -- against vectors produced by the
    reference's literal loop (train.py:116-121: two forwards, the last stage's loss survives), plus the oracle on the full maps."""
    from awr_amd.trainer import TrainEngine
    g = np.load(os.path.join(golden_dir, "hourglass_2_c5.npz"))
    net, J, ks, H = "hourglass_2", int(g["J"]), float(g["ks"]), 256
    assert J == 21 and g["img"].shape[-1] == H
    img, jt_gt = torch.from_numpy(g["img"]), torch.from_numpy(g["jt_gt"])
    B = img.shape[0]
    man = O.manifest_for(net, J)
    pkeys = [str(k) for k in g["pkeys"]]
    fm = amd.FeatureModule()
    # ---- eval forward ----
    m = make_net(amd, net, J, O.procedural_state(man, seed=3))
    m.eval()
    with torch.no_grad():
        outs = m(img.to(dev))
    oracle = O.backbone_forward(net, O.procedural_state(man, seed=3), img, training=False)
    gaps = oracle_fp64_joint_gap(net, O.procedural_state(man, seed=3), img, ks, False)
    assert len(outs) == 2 and tuple(outs[1].shape) == (B, 4 * J, H // 2, H // 2)
    for s_, o in enumerate(outs):
        ref = g["eval_s%d_val" % s_]
        scale = max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(o.cpu().reshape(-1).numpy()[g["eval_s%d_idx" % s_]] - ref).max()) / scale
        full = float((o.cpu() - oracle[s_]).abs().max()) / scale
        report("config5/eval/stage%d/dense_map_rel_err" % s_, full)
        assert err <= 2e-4 and full <= 2e-4, (s_, err, full)
        assert_joints("config5/eval/stage%d" % s_, fm.offset2joint_softmax(o, img.to(dev), ks).cpu().numpy(), g["eval_s%d_jt" % s_], gaps[s_])
    # ---- fused train step (nstack = 2) ----
    m = make_net(amd, net, J, O.procedural_state(man, seed=3))
    eng = TrainEngine(m, B, H, ks, coord_weight=1.0, dense_weight=1.0, lr=1e-3, use_graph=False)
    assert eng.plan.bn_repeat == 2 and eng.stage == 1
    losses, jt = eng.step(img.to(dev), jt_gt.to(dev))
    l0, ref0 = float(losses[2]), float(g["loss0"])
    report("config5/loss0_rel_err", abs(l0 - ref0) / abs(ref0))
    assert abs(l0 - ref0) <= 2e-4 * abs(ref0), (l0, ref0)
    assert abs(float(losses[0]) - float(g["lcoord0"])) <= 2e-4 * abs(float(g["lcoord0"])) + 1e-9
    assert abs(float(losses[1]) - float(g["ldense0"])) <= 2e-4 * abs(float(g["ldense0"])) + 1e-9
    pred = eng.dense_map(1).cpu().reshape(-1).numpy()[g["pred_idx"]]               # last stage's dense map, training-mode BN
    assert float(np.abs(pred - g["pred_val"]).max()) <= 2e-4 * max(1.0, float(np.abs(g["pred_val"]).max()))
    gap = oracle_fp64_joint_gap(net, O.procedural_state(man, seed=3), img, ks, True)[-1]
    assert_joints("config5/train", jt.cpu().numpy(), g["jt0"], gap)
    worst = check_grad_norms(m, pkeys, g["grad_l2"])
    report("config5/worst_grad_norm_rel_err", worst)
    got_sd = m.state_dict()
    for i, k in enumerate(g["bn_keys"]):              # running statistics after the COMPOUNDED momentum == two literal updates
        np.testing.assert_allclose(got_sd[str(k)].cpu().numpy(), g["bn_%d" % i], rtol=2e-4, atol=2e-5)
    counters = [v for k, v in got_sd.items() if k.endswith("num_batches_tracked")]
    assert len(counters) > 50 and all(int(v) == int(g["num_batches_tracked"]) == 2 for v in counters)
    p1 = np.array([float(got_sd[k].reshape(-1)[smp_index(got_sd[k].numel(), i)]) for i, k in enumerate(pkeys)], np.float32)
    d1 = np.abs(p1 - g["param_smp1"])
    assert np.quantile(d1, 0.9) <= 1e-4 and d1.max() <= 2.1e-3, (np.quantile(d1, 0.9), d1.max())
    for k in m._unused:                                # skip_layers that never run: untouched by the optimiser
        assert torch.equal(got_sd[k].cpu(), O.procedural_state(man, seed=3)[k])
    losses, _ = eng.step(img.to(dev), jt_gt.to(dev))
    ref1 = float(g["loss1"])
    report("config5/loss1_rel_err", abs(float(losses[2]) - ref1) / abs(ref1))
    assert abs(float(losses[2]) - ref1) <= 2e-2 * abs(ref1)
    assert all(int(v) == 4 for k, v in m.state_dict().items() if k.endswith("num_batches_tracked"))


@pytest.mark.parametrize("net", ["resnet_18", "hourglass_1", "hourglass_2"])
def test_reference_initialised_weights_meet_north_star(amd, dev, net):
    """Weights drawn from the reference's own initialisers (resnet_deconv.py:93-115 / torch defaults, what a training run and
    bench.py start from): every stage's joints within 1e-3 mm MEAN of the oracle, eval and train mode -- no yardstick needed for
    the mean; the single worst joint (a noisy statistic: batch-statistics BatchNorm over 4 images) is held to 5e-3 mm or six
    times the oracle's own fp32-vs-fp64 gap."""
    J, B = 14, 4
    ks = 1.0 if net.startswith("resnet") else 0.4
    img, _ = O.synth_batch(B, 128, J, seed=17)
    fm = amd.FeatureModule()
    for mode in ("eval", "train"):
        sd = O.reference_init_state(net, J, seed=5)
        m = make_net(amd, net, J, sd)
        m.train(mode == "train")
        with torch.no_grad():
            outs = m(img.to(dev))
        outs = outs if isinstance(outs, list) else [outs]
        with torch.no_grad():
            oracle = O.backbone_forward(net, O.reference_init_state(net, J, seed=5), img, training=(mode == "train"))
        gaps = oracle_fp64_joint_gap(net, O.reference_init_state(net, J, seed=5), img, ks, training=(mode == "train"))
        for s_, (o, r) in enumerate(zip(outs, oracle)):
            d = (fm.offset2joint_softmax(o, img.to(dev), ks).cpu() - O.offset2joint_softmax(r, img, ks)).norm(dim=-1) * 150.0
            report("%s/refinit/%s/stage%d/joint_err_mm_mean" % (net, mode, s_), float(d.mean()))
            report("%s/refinit/%s/stage%d/joint_err_mm" % (net, mode, s_), float(d.max()))
            report("%s/refinit/%s/stage%d/oracle_fp32_vs_fp64_gap_mm_max" % (net, mode, s_), gaps[s_][1])
            assert float(d.mean()) <= NORTH_STAR_MEAN_MM and float(d.max()) <= max(5e-3, 6.0 * gaps[s_][1]), \
                (net, mode, s_, float(d.mean()), float(d.max()), gaps[s_])


_YARD_CACHE, _YARD_KEEP = {}, {("hourglass_1", 1.0)}


@pytest.mark.parametrize("net,cw", [("resnet_18", 0.0), ("resnet_18", 1.0), ("hourglass_1", 0.0), ("hourglass_1", 1.0), ("resnet_50", 0.0)])
def test_gradients_elementwise_against_the_fp64_yardstick(amd, dev, net, cw):
    """Whole gradient tensors (not norms) against float64, tensor by tensor -- with the ReLU decisions of the implementation under
    test (tests/yardstick.py).  The loss is piecewise smooth and a ReLU whose pre-activation is ~1e-6 gets derivative 0 in one fp32
    implementation and 1 in another; everything upstream in the backward inherits that.  Round 2 widened the tolerance by a "kink noise
    floor" so large (1.8e-2) that it could hide a 4x regression; round 3 removes the noise instead: the plan's own ReLU decisions
    (awr_plan_tensor: materialised activations and the lazy BatchNorm coefficients) are fed to the float64 evaluation, which then
    differentiates exactly the branch the HIP step took (and likewise for the fp32 oracle with ITS decisions); max-pool windows whose
    two largest elements are within rounding of each other are the same kind of decision and are handled the same way.  What is left is rounding:
    the HIP gradients must sit within 3x the fp32 oracle's own distance from float64 (the MFMA accumulates K sequentially; its forward
    activations carry 1.8-2.6x oneDNN's rounding error on the same inputs: tools/diag_parity.py, profiles/r03_diag_parity.txt).
    Decisions may only differ from float64's where the float64 pre-activation is within 1e-4 of zero (relative to 1 + the tensor's
    largest magnitude)."""
    import yardstick as Y
    from awr_amd.trainer import TrainEngine
    J, B = 14, 2
    ks = 1.0 if net.startswith("resnet") else 0.4
    img, jt_gt = O.synth_batch(B, 128, J, seed=23)
    sd = O.reference_init_state(net, J, seed=9)
    if (net, cw) not in _YARD_CACHE:      # (the float64 / fp32-oracle traces do not depend on the kernel mode under test: the Winograd wrapper reuses them)
        ref = Y.trace(net, sd, img, jt_gt, ks, cw, True)
        f32 = Y.trace(net, sd, img, jt_gt, ks, cw, False)
        fl32, pl32 = Y.decisions_from_trace(ref, f32)
        ent = (ref, f32, fl32, pl32, Y.trace(net, sd, img, jt_gt, ks, cw, True, flips=fl32, pools=pl32))
        if (net, cw) in _YARD_KEEP:       # (a trace holds every activation in float64 -- gigabytes: only the one combination that is used twice stays)
            _YARD_CACHE[(net, cw)] = ent
    else:
        ent = _YARD_CACHE[(net, cw)]
    ref, f32, fl32, pl32, ref_f32 = ent
    m = make_net(amd, net, J, sd)
    eng = TrainEngine(m, B, 128, ks, coord_weight=cw, dense_weight=1.0, lr=1e-3, autotune=False)
    eng.step(img.to(dev), jt_gt.to(dev))
    torch.cuda.synchronize()
    flips, pools, rep = Y.decisions_from_plan(ref, eng.plan.tensors(lazy=True))
    for tag, n, mx in rep:
        assert mx < 1e-4, ("ReLU / max-pool decision differs from float64 away from a kink", tag, n, mx)
    ref_hip = Y.trace(net, sd, img, jt_gt, ks, cw, True, flips=flips, pools=pools)
    # the fused ResNet stem never materialises its ReLU / pooling decisions: they stay float64's, their near-kinks stay a (small) allowance
    # for the tensors upstream of them: the stem's own parameters
    stem_kink = Y.stem_allowance(ref) if net.startswith("resnet") else 0.0
    pkeys = O.params_of(sd, O.manifest_for(net, J))
    gmax = max(float(ref["grads"][k].norm()) for k in pkeys if ref["grads"][k] is not None)
    rows, bad = [], []
    for k in pkeys:
        if ref["grads"][k] is None:
            assert k in m._unused
            continue
        floor = 1e-3 * gmax       # (a conv bias in front of a BatchNorm has a zero true gradient: pure rounding noise)
        e_hip = Y.rel_l2(m.grad_view(k).cpu(), ref_hip["grads"][k], floor)
        e_f32 = Y.rel_l2(f32["grads"][k], ref_f32["grads"][k], floor)
        e_raw = Y.rel_l2(m.grad_view(k).cpu(), ref["grads"][k], floor)
        rows.append((e_hip / max(e_f32, 1e-7), k, e_hip, e_f32, e_raw))
        allow = 1.5 * stem_kink if k.startswith("pre.") else 0.0
        if not e_hip <= 3.0 * e_f32 + allow + 2e-5:
            bad.append((k, e_hip, e_f32, stem_kink))
    ratios = [r[0] for r in rows]
    tagp = "%s/cw%d/grad_vs_fp64/" % (net, int(cw))
    report(tagp + "median_ratio_hip_over_fp32_oracle", float(np.median(ratios)))
    report(tagp + "max_rel_l2_err_hip", max(r[2] for r in rows))
    report(tagp + "max_rel_l2_err_fp32_oracle", max(r[3] for r in rows))
    report(tagp + "max_rel_l2_err_hip_without_relu_decisions", max(r[4] for r in rows))
    report(tagp + "relu_decisions_flipped_hip", sum(n for _, n, _ in rep))
    report(tagp + "stem_kink_allowance", stem_kink)
    report(tagp + "relu_decisions_flipped_fp32_oracle", sum(int(v.sum()) for v in fl32.values()) + len(pl32))
    print("median error ratio HIP / fp32 oracle %.2f, max rel. L2 error HIP %.2e (%.2e before the ReLU decisions) / fp32 oracle %.2e; decisions flipped: %s" %
          (float(np.median(ratios)), max(r[2] for r in rows), max(r[4] for r in rows), max(r[3] for r in rows), rep))
    for r in sorted(rows, reverse=True)[:4]:
        print("  %6.2f  %-40s hip %.2e  fp32 oracle %.2e  (raw %.2e)" % r)
    assert not bad, bad[:8]


def test_dropin_loop_sees_every_optimizer_step(amd, dev):
    """The advertised drop-in loop -- net(x) -> loss.backward() -> STOCK torch.optim.Adam.step() -- for several iterations: the
    packed GEMM copies of the weights must follow the in-place updates the optimiser makes through the nn.Parameter objects
    (their version counters, not the arena's, move)."""
    net, J, ks, B = "resnet_18", 14, 1.0, 2
    img, jt_gt = O.synth_batch(B, 128, J, seed=33)
    man = O.manifest_for(net, J)
    m = make_net(amd, net, J, O.procedural_state(man, seed=2))
    m.train()
    fm, crit = amd.FeatureModule(), amd.My_SmoothL1Loss().cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    x, jg = img.to(dev), jt_gt.to(dev)
    gt = fm.joint2offset(jg, x, ks, 64)
    sdo, ost = O.procedural_state(man, seed=2), {"step": 0, "m": {}, "v": {}}
    got, ref = [], []
    for it in range(3):
        pred = m(x)
        loss = crit(fm.offset2joint_softmax(pred, x, ks), jg) + crit(pred, gt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        got.append(float(loss))
        ref.append(float(O.train_step(net, sdo, ost, img, jt_gt, ks, 1.0, 1.0)[0]))
    assert abs(got[0] - ref[0]) <= 2e-4 * abs(ref[0])
    for it in (1, 2):                                   # stale packed weights would repeat the first loss
        assert abs(got[it] - ref[it]) <= 0.25 * abs(ref[it] - ref[it - 1]) + 2e-3 * abs(ref[it]), (got, ref)
    # eval-mode plan built BEFORE further updates must also follow them (signature = parameter version counters)
    m.eval()
    with torch.no_grad():
        a = m(x).clone()
    m.train()
    pred = m(x)
    loss = crit(pred, gt)
    opt.zero_grad()
    loss.backward()
    opt.step()
    m.eval()
    with torch.no_grad():
        b = m(x)
        ref_b = O.resnet18_forward({k: v.cpu() for k, v in m.state_dict().items()}, img, False)
    assert not torch.equal(a, b)
    assert float((b.cpu() - ref_b).abs().max()) <= 2e-4 * max(1.0, float(ref_b.abs().max()))


def test_inference_engine_and_graph_replay(amd, dev):
    from awr_amd.trainer import InferEngine, TrainEngine
    J = 14
    img, jt_gt = O.synth_batch(4, 128, J, seed=41)
    man = O.manifest_for("resnet_18", J)
    sd = O.procedural_state(man, seed=4)
    m = make_net(amd, "resnet_18", J, sd)
    inf = InferEngine(m, 4, 128, 1.0, use_graph=True)
    ref = O.offset2joint_softmax(O.resnet18_forward(O.procedural_state(man, seed=4), img), img, 1.0)
    gap = oracle_fp64_joint_gap("resnet_18", O.procedural_state(man, seed=4), img, 1.0, False)[0]
    for it in range(4):                      # iterations 0-1 eager, 2 captures, 3 replays
        jt = inf(img.to(dev))
        assert_joints("resnet_18/infer_engine/it%d" % it, jt.cpu().numpy(), ref.numpy(), gap)
    # graph-captured train step == eager train step (same inputs, fresh nets): bit for bit in deterministic mode, to rounding
    # noise amplified by Adam in the default mode (split-K atomics order differs run to run)
    for det in (True, False):
        amd.set_deterministic(det)
        try:
            res = []
            for use_graph in (False, True):
                mm = make_net(amd, "resnet_18", J, O.procedural_state(man, seed=4))
                eng = TrainEngine(mm, 4, 128, 1.0, coord_weight=1.0, dense_weight=1.0, use_graph=use_graph)
                ls = [float(eng.step(img.to(dev), jt_gt.to(dev))[0][2]) for _ in range(4)]
                res.append((ls, mm.flat_params().clone()))
        finally:
            amd.set_deterministic(False)
        if det:
            assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
        else:
            assert np.allclose(res[0][0][:3], res[1][0][:3], rtol=2e-3) and np.allclose(res[0][0], res[1][0], rtol=3e-2), (res[0][0], res[1][0])
            dpar = (res[0][1] - res[1][1]).abs()
            assert float(torch.quantile(dpar[:1000000], 0.9)) <= 4e-4 and float(dpar.max()) <= 8.1e-3      # (4 Adam steps of 1e-3: noise-level differences)


def test_roundtrip_save_load_checkpoint(amd, dev, tmp_path):
    """train.py:165-170 / test.py:45-49: {'model','optimizer','best_records'} round trip."""
    from awr_amd.trainer import TrainEngine
    J = 14
    img, jt_gt = O.synth_batch(2, 128, J, seed=51)
    m = make_net(amd, "hourglass_1", J, O.procedural_state(O.manifest_for("hourglass_1", J), seed=5))
    eng = TrainEngine(m, 2, 128, 0.4, use_graph=False)
    eng.step(img.to(dev), jt_gt.to(dev))
    path = os.path.join(tmp_path, "epoch_1.pth")
    torch.save({"model": m.state_dict(), "optimizer": eng.optimizer_state_dict(), "best_records": {"epoch": 1, "MPE": 1e10, "AUC": 0}}, path)
    pth = torch.load(path, weights_only=False)
    m2 = amd.PoseNet("hourglass_1", J).cuda()
    m2.load_state_dict(pth["model"])
    assert torch.equal(m2.flat_params()[:m2.n_params], m.flat_params()[:m.n_params])
    opt = torch.optim.Adam(m2.parameters(), lr=1e-3)
    opt.load_state_dict(pth["optimizer"])                      # stock optimizer accepts the engine's state
    m.eval(); m2.eval()
    with torch.no_grad():
        a, b = m(img.to(dev))[0], m2(img.to(dev))[0]
    assert torch.equal(a, b)


def test_trainer_end_to_end(amd, dev, tmp_path):
    """train.py / test.py orchestration (a9) on a synthetic hand dataset: the loss goes down, the evaluator
    reports a finite mm error, StepLR semantics, checkpoint files and the results txt are written and reload."""
    from awr_amd.trainer import SyntheticHands, Trainer
    from awr_amd.config import Config

    class Cfg(Config):
        net = "resnet_18"
        kernel_size = 1.0
        batch_size = 8
        num_workers = 0
        max_epoch = 3
        step = 2
        print_freq = 2
        output_dir = str(tmp_path)
        load_model = ""
        exp_id = "t"
        coord_weight = 1.0
        use_hipgraph = False
    tr = Trainer(Cfg(), SyntheticHands(32, seed=1), SyntheticHands(12, seed=2))
    mpe0 = tr.test(0)
    # scoring passes (test.py:67-86) run ordered by default since round 6: eval-mode plans measured no gain from blocked accumulation
    # (1.851e-4 mm from the oracle either way, profiles/r05_parity_report.json) and paid ~3 % for it; config.parity_infer opts in
    assert Cfg().parity_infer is False and not tr._last_infer.parity and tr._last_infer.plan.accum == 0
    assert tr.engine.plan.accum == {"ordered": 0, "blocked": 1, "auto": 2}[Cfg().accum]
    tr.train()
    work = os.path.join(str(tmp_path), "nyu", "checkpoint_t")
    log = open(os.path.join(work, "resnet_18_dense.log")).read()
    assert "[epoch 01], [train loss" in log and "[epoch  3], [test mpe" in log and "learning rate: 1.0e-03" in log
    assert abs(tr.engine.lr - 1e-3 * 0.1 ** (3 // 2)) < 1e-12                     # StepLR(step_size=2, gamma=0.1).step(3)
    assert os.path.exists(os.path.join(work, "test_pck_epoch_0.png"))                 # train.py:216
    assert any(f.startswith("test_epoch_0_iter_") for f in os.listdir(os.path.join(work, "results")))   # train.py:203-213 (vis_freq = 1)
    losses = [float(l.split("[train loss ")[1].split("]")[0]) for l in log.splitlines() if l.startswith("[epoch 0") and "train mpe" in l]
    assert len(losses) == 3 and losses[-1] < losses[0]
    assert np.isfinite(mpe0) and any(f.startswith("test_") and f.endswith(".txt") for f in os.listdir(work))
    txt = np.loadtxt(os.path.join(work, [f for f in os.listdir(work) if f.startswith("test_") and f.endswith(".txt")][0]))
    assert txt.shape == (12, 42)                                                   # results/*.txt format (test.py:105-108)
    pth = torch.load(os.path.join(work, "epoch_3.pth"), weights_only=False)
    assert set(pth) == {"model", "optimizer", "best_records"} and len(pth["model"]) == 142

    class Cfg2(Cfg):
        load_model = os.path.join(work, "epoch_3.pth")
        exp_id = "t2"
    tr2 = Trainer(Cfg2(), SyntheticHands(32, seed=1), SyntheticHands(12, seed=2))
    assert torch.equal(tr2.net.flat_params(), tr.net.flat_params()) and tr2.engine.step_count == tr.engine.step_count
    assert torch.equal(tr2.engine.m, tr.engine.m) and abs(tr2.engine.lr - 1e-3) < 1e-12       # LR force-reset (train.py:94-96)
    assert abs(tr2.test(1) - tr.test(1)) < 1e-5       # (mm, of ~80: two engines, independently autotuned tiles = different summation orders)

    class Cfg3(Cfg2):      # opting in: the parity mode (blocked accumulation) for scoring
        parity_infer = True
        exp_id = "t3"
    tr3 = Trainer(Cfg3(), SyntheticHands(32, seed=1), SyntheticHands(12, seed=2))
    m3 = tr3.test(1)
    assert tr3._last_infer.parity and tr3._last_infer.plan.accum == 1 and abs(m3 - tr.test(1)) < 1e-3

    # the split-operand mode has no blocked kernel -- a scoring pass that asks for parity there must run (ordered, with a
    # warning), not fail at its first launch; a training engine's "auto" degrades to ordered the same way
    class Cfg4(Cfg3):
        gemm_products = 6
        exp_id = "t4"
    try:
        with pytest.warns(UserWarning, match="blocked accumulation is not available"):
            tr4 = Trainer(Cfg4(), SyntheticHands(32, seed=1), SyntheticHands(12, seed=2))
            m4 = tr4.test(1)
        assert not tr4._last_infer.parity and tr4._last_infer.plan.accum == 0 and abs(m4 - m3) < 1e-2
        from awr_amd._lib import AwrError
        with pytest.raises(AwrError, match="blocked accumulation"):
            tr4.net.get_plan(8, 128, False, accum="blocked")
    finally:
        amd.set_gemm_products(1)


@pytest.mark.parametrize("net,streams", [("resnet_18", 0), ("resnet_18", 2), ("hourglass_1", 2)])
def test_bucketed_backward_matches_and_is_final_at_the_marker(amd, dev, net, streams):
    """Data-parallel overlap: with n_buckets > 1 the backward hands out arena ranges as soon as they are final.
    Snapshots taken at each marker must equal the end-of-step gradients, and those must equal the 1-bucket plan's.  With side
    streams the weight gradients run beside the chain, the hourglass backward forks its outer levels onto branch streams and the
    buckets are handed to the comm stream (the hook runs with torch switched to it)."""
    J = 14
    img, jt_gt = O.synth_batch(2, 128, J, seed=61)
    man = O.manifest_for(net, J)
    grads = []
    for nb in (1, 4):
        m = make_net(amd, net, J, O.procedural_state(man, seed=6))
        m.train()
        plan = m.get_plan(2, 128, True, supervised=(0,), n_buckets=nb)
        if streams:
            plan.set_streams(streams, comm=nb > 1)
        snaps = []
        plan.bucket_hook = lambda lo, hi: snaps.append((lo, hi, m.flat_grads()[lo:hi].clone()))
        m.sync_weights(plan, force=True)
        plan.img.copy_(img.to(dev))
        plan.forward()
        plan.grad_outs[0].copy_(_hashed_like(plan.grad_outs[0]))
        plan.backward()
        torch.cuda.synchronize()
        g = m.flat_grads()[:m.n_active].clone()
        grads.append(g)
        if nb > 1:
            assert len(snaps) >= 2 and snaps[0][1] == m.n_active and snaps[-1][0] == 0
            assert all(a[0] == b[1] for a, b in zip(snaps, snaps[1:]))
            for lo, hi, snap in snaps:
                assert torch.equal(snap, m.flat_grads()[lo:hi]), (lo, hi)      # nothing wrote the range after its marker
    d = (grads[0] - grads[1]).abs().max() / grads[0].abs().max()
    assert float(d) < 1e-4          # same kernels, split-K atomics order differs


def test_a_failing_bucket_hook_is_raised_not_swallowed(amd, dev):
    """the bucket hook runs inside a ctypes callback in the middle of the native backward replay; ctypes prints
    'Exception ignored' and carries on, the optimiser would then step an un-reduced bucket.  The binding stashes the error and
    run_backward() re-raises it after the native call has returned."""
    from awr_amd._lib import AwrError
    J = 14
    img, _ = O.synth_batch(2, 128, J, seed=62)
    m = make_net(amd, "resnet_18", J, O.procedural_state(O.manifest_for("resnet_18", J), seed=6))
    m.train()
    plan = m.get_plan(2, 128, True, supervised=(0,), n_buckets=4)
    plan.set_streams(2, comm=True)
    calls = []

    def hook(lo, hi):
        calls.append((lo, hi))
        if len(calls) == 2:
            raise RuntimeError("collective failed")
    plan.bucket_hook = hook
    m.sync_weights(plan, force=True)
    plan.img.copy_(img.to(dev))
    plan.forward()
    plan.grad_outs[0].copy_(_hashed_like(plan.grad_outs[0]))
    with pytest.raises(AwrError, match="bucket hook failed"):
        plan.backward()
    torch.cuda.synchronize()
    assert len(calls) == 2          # later buckets are not handed out after a failure
    plan.bucket_hook = lambda lo, hi: None
    plan.forward()
    plan.backward()                 # the plan stays usable
    torch.cuda.synchronize()


def _hashed_like(t):
    v = O._hash_uniform(t.numel(), 77, 5) * np.float32(1e-3)
    return torch.from_numpy(v.reshape(tuple(t.shape)).copy()).to(t.device)


def test_train_engine_with_one_rank_rccl_group(amd, dev, monkeypatch):
    """The data-parallel code path (RCCL broadcast, bucketed async all-reduce overlapped with the backward, 1/world
    scale) on a 1-rank `nccl` group: must give the same step as the single-process engine."""
    import torch.distributed as dist
    from awr_amd.trainer import TrainEngine
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29533")
    monkeypatch.setenv("AWR_FORCE_DP", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        J = 14
        img, jt_gt = O.synth_batch(2, 128, J, seed=71)
        man = O.manifest_for("resnet_18", J)
        out = []
        # (None: single process; torch: the Python bucket hook + torch.distributed all-reduce; native: the library's own RCCL communicator
        # -- awr_dp_*, librccl.so through dlopen -- attached to the plan, no Python in the exchange)
        for pg, native in ((None, False), (dist.group.WORLD, False), (dist.group.WORLD, True)):
            m = make_net(amd, "resnet_18", J, O.procedural_state(man, seed=7))
            eng = TrainEngine(m, 2, 128, 1.0, coord_weight=1.0, use_graph=False, process_group=pg, native_rccl=native)
            assert eng.dp == (pg is not None) and len(eng.plan.buckets) == (4 if pg is not None else 1)
            assert (eng.dpcomm is not None) == native
            l = [float(eng.step(img.to(dev), jt_gt.to(dev))[0][2]) for _ in range(2)]
            out.append((l, m.flat_params().clone()))
        for k in (1, 2):
            assert np.allclose(out[0][0], out[k][0], rtol=1e-4)
            d = (out[0][1] - out[k][1]).abs()
            assert float(torch.quantile(d[:1000000], 0.9)) <= 2e-4
    finally:
        dist.destroy_process_group()


def test_native_rccl_communicator_through_the_c_abi(amd, dev):
    """awr_dp_* (include/awr_hip.h "Data-parallel API"): what a non-Python host uses.  One rank here (RCCL refuses two ranks on one
    device; tests/test_dp_gpu.py runs the two-rank form where two GPUs are visible): unique id, communicator on the current device,
    in-place all-reduce / broadcast ordered behind the caller's stream, and a one-bucket plan whose backward exchanges the whole
    gradient arena itself."""
    from awr_amd.engine import DpComm
    ok, version, path = DpComm.available()
    assert ok and version > 0 and "rccl" in path
    dp = DpComm(0, 1, DpComm.unique_id())
    x = torch.arange(1 << 20, device=dev, dtype=torch.float32)
    y = x.clone()
    dp.allreduce(y)
    dp.broadcast(y, 0)
    dp.wait()
    torch.cuda.synchronize()
    assert torch.equal(x, y)                       # SUM over one rank
    J = 14
    img, _ = O.synth_batch(2, 128, J, seed=72)
    grads = []
    for attach in (False, True):
        m = make_net(amd, "resnet_18", J, O.procedural_state(O.manifest_for("resnet_18", J), seed=7))
        m.train()
        plan = m.get_plan(2, 128, True, supervised=(0,), n_buckets=1 if not attach else 4)
        plan.set_streams(2, comm=attach)
        if attach:
            plan.set_dp(dp)
        m.sync_weights(plan, force=True)
        plan.img.copy_(img.to(dev))
        plan.forward()
        plan.grad_outs[0].copy_(_hashed_like(plan.grad_outs[0]))
        plan.backward()
        torch.cuda.synchronize()
        grads.append(m.flat_grads()[:m.n_active].clone())
        if attach:
            plan.set_dp(None)
    assert float((grads[0] - grads[1]).abs().max() / grads[0].abs().max()) < 1e-4
    # lifetime rule: a communicator destroyed UNDER a plan makes every later backward fail -- not only the first -- until the host
    # detaches or re-attaches; replicas that silently skipped an exchange would diverge
    from awr_amd._lib import AwrError
    plan.set_dp(dp)
    dp.close()
    for _ in range(2):
        plan.forward()
        with pytest.raises(AwrError):
            plan.backward()
        torch.cuda.synchronize()
    plan.set_dp(None)
    plan.forward()
    plan.backward()
    torch.cuda.synchronize()
    # the HOST's bucket callback survives lost communicator -> re-attach -> detach (the re-attach used to save the plan's own
    # callback over it, after which gradients were silently no longer handed to the host)
    seen = []
    plan.bucket_hook = lambda lo, hi: seen.append((lo, hi))
    dp2 = DpComm(0, 1, DpComm.unique_id())
    plan.set_dp(dp2)
    dp2.close()
    plan.forward()
    with pytest.raises(AwrError):
        plan.backward()
    torch.cuda.synchronize()
    dp3 = DpComm(0, 1, DpComm.unique_id())
    plan.set_dp(dp3)                     # re-attach while the lost communicator's callback is still installed
    plan.forward()
    plan.backward()
    torch.cuda.synchronize()
    assert not seen                      # buckets went to the communicator
    plan.set_dp(None)
    dp3.close()
    plan.forward()
    plan.backward()
    torch.cuda.synchronize()
    assert len(seen) == 4 and seen[-1][1] > seen[-1][0], seen
    plan.bucket_hook = None


@pytest.mark.parametrize("net", ["resnet_18", "hourglass_1"])
def test_wgrad_side_streams_do_not_change_the_step(amd, dev, net):
    """Weight-gradient GEMMs on extra HIP streams (and only those whose dY is never aliased) == the serial step."""
    from awr_amd.trainer import TrainEngine
    J = 14
    ks = 1.0 if net.startswith("resnet") else 0.4
    img, jt_gt = O.synth_batch(2, 128, J, seed=81)
    man = O.manifest_for(net, J)
    res = []
    for nstreams in (0, 2):
        m = make_net(amd, net, J, O.procedural_state(man, seed=8))
        eng = TrainEngine(m, 2, 128, ks, coord_weight=1.0, use_graph=False, autotune=False, wgrad_streams=nstreams)
        if nstreams:
            n_side = len(eng.plan._side_ok)
            n_all = sum(1 for n in eng.plan.op_names("bwd") if n.startswith("awr_conv_wgrad"))
            assert 0 < n_side <= n_all and (net.startswith("resnet") and n_side == n_all or n_side < n_all)
        for it in range(3):
            losses, _ = eng.step(img.to(dev), jt_gt.to(dev))
            torch.cuda.synchronize()
            if it == 0:
                g0 = m.flat_grads()[:m.n_active].clone()      # same weights in both runs: differences = atomics order only
        res.append((float(losses[2]), g0))
    d = (res[0][1] - res[1][1]).double().norm() / res[0][1].double().norm()
    assert float(d) < 1e-3, float(d)
    assert abs(res[0][0] - res[1][0]) <= 3e-2 * abs(res[0][0])       # after 3 Adam steps (chaotic amplification of rounding)


@pytest.mark.parametrize("ds", [1, 4])
def test_resnet_other_downsample_rates(amd, dev, ds):
    """config.downsample in [1,2,4] (config.py:31): 4 - log2(ds) deconv stages; forward + one fused train step vs the oracle."""
    from awr_amd.trainer import TrainEngine
    J, B = 14, 2
    img, jt_gt = O.synth_batch(B, 128, J, seed=91)
    man = O.resnet18_manifest(J, ds)
    sd = O.procedural_state(man, seed=9)
    m = amd.get_deconv_net(18, J, ds)
    assert [k for k in m.state_dict()] == [k for k, _, _ in man]
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(img.to(dev)).cpu()
        ref = O.resnet18_forward(O.procedural_state(man, seed=9), img, False, ds)
    assert out.shape == ref.shape == (B, 4 * J, 128 // ds, 128 // ds)
    assert float((out - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))
    eng = TrainEngine(m, B, 128, 1.0, coord_weight=1.0, use_graph=False, autotune=False)
    losses, jt = eng.step(img.to(dev), jt_gt.to(dev))
    F = 128 // ds
    sdo = O.procedural_state(man, seed=9)
    pred = O.resnet18_forward(sdo, img, True, ds)
    lref = O.huber(O.offset2joint_softmax(pred, img, 1.0), jt_gt) + O.huber(pred, O.joint2offset(jt_gt, img, 1.0, F))
    assert abs(float(losses[2]) - float(lref)) <= 3e-4 * abs(float(lref))


@pytest.mark.parametrize("optimizer,wd", [("sgd", 0.0), ("adam", 1e-2), ("sgd", 1e-2)])
def test_train_engine_optimizer_variants_vs_torch(amd, dev, optimizer, wd):
    """train.py:66-69: Adam(lr, weight_decay) / SGD(lr, momentum=0.9, weight_decay).  The engine's flat-arena optimiser
    kernels must move the parameters like the stock torch optimiser does when it is fed the engine's own gradients."""
    from awr_amd.trainer import TrainEngine
    J = 14
    img, jt_gt = O.synth_batch(2, 128, J, seed=95)
    man = O.manifest_for("resnet_18", J)
    m = make_net(amd, "resnet_18", J, O.procedural_state(man, seed=9))
    eng = TrainEngine(m, 2, 128, 1.0, coord_weight=1.0, lr=1e-3, weight_decay=wd, optimizer=optimizer, use_graph=False, autotune=False)
    shadow = torch.nn.Parameter(m.flat_params()[:m.n_active].clone())
    opt = (torch.optim.Adam([shadow], lr=1e-3, weight_decay=wd) if optimizer == "adam"
           else torch.optim.SGD([shadow], lr=1e-3, momentum=0.9, weight_decay=wd))
    for _ in range(3):
        before = m.flat_params()[:m.n_active].clone()
        eng.step(img.to(dev), jt_gt.to(dev))
        torch.cuda.synchronize()
        with torch.no_grad():
            shadow.copy_(before)                      # same starting point, same gradient: compare one update at a time
        shadow.grad = m.flat_grads()[:m.n_active].clone()
        opt.step()
        d = (m.flat_params()[:m.n_active] - shadow.detach()).abs().max()
        assert float(d) <= 2e-6, (optimizer, wd, float(d))


@pytest.mark.parametrize("net", ["resnet_18", "hourglass_1"])
def test_split_operand_mode_meets_the_same_golden_bars(amd, dev, golden_dir, net):
    """awr_amd.set_gemm_products(6): every conv GEMM of forward, dgrad and wgrad on the bf16 matrix pipe with exactly split
    fp32 operands.  Same golden vectors, same tolerances as the FP32-MFMA mode (forward maps, joints, BN statistics,
    losses, gradient norms, two Adam steps)."""
    amd.set_gemm_products(6)
    try:
        assert amd.get_gemm_products() == 6
        # (the opt-in mode has no blocked kernel: its GEMMs accumulate in one ordered chain and the chaotic two-image ResNet18 fixtures land at
        # 1.9x ... 2.6x the oracle's own distance from float64, session to session -- the yardstick ratio of THIS mode is 3.0, what its bar was
        # in rounds 3-5 (three oracle gaps); the default mode's is 2.0)
        test_backbone_forward_golden(amd, dev, golden_dir, net, yardstick=3.0)
        test_fused_train_step_golden(amd, dev, golden_dir, net, "c1", 1.0, yardstick=3.0)
    finally:
        amd.set_gemm_products(1)


@pytest.mark.parametrize("net", ["resnet_18", "hourglass_1"])
def test_winograd_forward_mode_meets_the_same_golden_bars(amd, dev, golden_dir, net):
    """awr_amd.set_conv_winograd: the stride-1 3x3 convolutions as Winograd F(2x2, 3x3) (csrc/awr_wino.hip) -- forward with fused input BatchNorm + ReLU,
    bias, BatchNorm statistics from its epilogue; in the "force" mode used here (= "full" on every layer the kernels can run, whatever the launch size, so
    that the two-image fixtures exercise them) also the data gradients and the Winograd-domain weight gradients.  Same golden vectors and bars as the direct
    mode (forward maps, joints, BN running statistics, losses, gradient norms, two Adam steps)."""
    from awr_amd.trainer import TrainEngine
    amd.set_conv_winograd("force")
    try:
        J = 14
        m = make_net(amd, net, J, O.procedural_state(O.manifest_for(net, J), seed=0))
        nw = m.get_plan(2, 128, True).n_winograd
        report("%s/winograd_launches" % net, nw)
        assert nw >= 4, nw
        test_backbone_forward_golden(amd, dev, golden_dir, net)
        test_fused_train_step_golden(amd, dev, golden_dir, net, "c1", 1.0)
    finally:
        amd.set_conv_winograd(False)
    # the engine's own switch, normal eligibility (enough workgroups to fill the chip): batch 64 takes it, batch 2 does not
    m = make_net(amd, net, 14, O.reference_init_state(net, 14, seed=3))
    nw2 = TrainEngine(m, 2, 128, 1.0, winograd=True, use_graph=False, autotune=False).plan.n_winograd
    assert nw2 < nw and (nw2 == 0 or net != "resnet_18"), (nw2, nw)
    assert not amd.get_conv_winograd()
    # "forward" (= True) replaces forward launches only, "full" the data gradients as well
    m64 = make_net(amd, net, 14, O.reference_init_state(net, 14, seed=3))
    nf, nfull = m64.get_plan(64, 128, True, winograd=True).n_winograd, m64.get_plan(64, 128, True, winograd="full").n_winograd
    assert 0 < nf < nfull, (nf, nfull)


def test_winograd_full_mode_gradients_elementwise(amd, dev):
    """"force" = forward, data gradients AND weight gradients (awr_wino_wgrad: the Winograd-domain weight gradient) of every stride-1 3x3 layer the kernels
    can run, whatever the launch size: whole gradient tensors against float64 with the plan's own ReLU decisions, same bar as the direct mode."""
    amd.set_conv_winograd("force")
    try:
        test_gradients_elementwise_against_the_fp64_yardstick(amd, dev, "hourglass_1", 1.0)
    finally:
        amd.set_conv_winograd(False)
        _YARD_CACHE.clear()


def test_train_and_test_entry_points(dev, tmp_path):
    """`python train.py` / `python test.py` (reference train.py:231-236, test.py:113-116) as subprocesses on a synthetic
    dataset: one epoch, checkpoint written, test.py reloads it and writes the results file."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--synthetic", "32", "--set", "net=resnet_18", "kernel_size=1", "batch_size=8", "num_workers=0", "max_epoch=1", "print_freq=2",
              "output_dir=%s" % tmp_path, "exp_id=entry", "use_hipgraph=False", "coord_weight=1.0"]
    r = subprocess.run([sys.executable, os.path.join(repo, "train.py")] + common + ["load_model="], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ckpt = os.path.join(str(tmp_path), "nyu", "checkpoint_entry", "epoch_1.pth")
    assert os.path.exists(ckpt), r.stdout[-2000:]
    r = subprocess.run([sys.executable, os.path.join(repo, "test.py")] + common + ["load_model=%s" % ckpt], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "loading model from" in r.stdout and "[test mpe" in r.stdout, r.stdout[-2000:]
    res = os.listdir(os.path.join(str(tmp_path), "nyu", "checkpoint_entry"))
    assert any(f.startswith("test_") and f.endswith(".txt") for f in res), res


@pytest.mark.parametrize("net,B", [("resnet_18", 3), ("hourglass_1", 2)])
def test_deterministic_mode_is_bitwise_reproducible(amd, dev, golden_dir, net, B):
    """awr_amd.set_deterministic(): one accumulator copy per producer workgroup (BatchNorm statistics, BN-backward sums, stem),
    one copy of every weight gradient per split-K chunk summed in order by the scatter, fixed-point integer loss sums.  Three
    train steps issued (a) eagerly on one stream, (b) as a hipGraph with the weight gradients on two side streams must agree BIT FOR
    BIT in losses, predictions, gradients, parameters and BN buffers -- and still meet the golden bars."""
    from awr_amd.trainer import TrainEngine
    J = 14
    ks = 1.0 if net.startswith("resnet") else 0.4
    img, jt_gt = O.synth_batch(B, 128, J, seed=83)
    man = O.manifest_for(net, J)
    amd.set_deterministic(True)
    try:
        assert amd.get_deterministic()
        runs = []
        for use_graph, streams in ((False, 0), (True, 2), (False, 2)):
            m = make_net(amd, net, J, O.procedural_state(man, seed=8))
            eng = TrainEngine(m, B, 128, ks, coord_weight=1.0, use_graph=use_graph, wgrad_streams=streams)
            assert eng.plan.det and not eng.plan.tuned
            ls, js = [], []
            for it in range(3):
                losses, jt = eng.step(img.to(dev), jt_gt.to(dev))
                ls.append(losses.clone())
                js.append(jt.clone())
            torch.cuda.synchronize()
            runs.append((torch.stack(ls), torch.stack(js), m.flat_grads()[:m.n_active].clone(), m.flat_params().clone(), m._barena.clone()))
        for other in runs[1:]:
            for a, b in zip(runs[0], other):
                assert torch.equal(a, b)
        if net == "resnet_18":
            test_fused_train_step_golden(amd, dev, golden_dir, net, "c1", 1.0)
    finally:
        amd.set_deterministic(False)
    # and the default mode really is order-dependent somewhere (otherwise this test proves nothing): not asserted, only reported
    m = make_net(amd, net, J, O.procedural_state(man, seed=8))
    eng = TrainEngine(m, B, 128, ks, coord_weight=1.0, use_graph=False, wgrad_streams=0, autotune=False)
    for it in range(3):
        eng.step(img.to(dev), jt_gt.to(dev))
    report("%s/deterministic/default_mode_param_max_abs_diff" % net, float((m.flat_params() - runs[0][3]).abs().max()))


@pytest.mark.parametrize("net", ["hourglass_1", "hourglass_2"])
def test_inference_fused_conv_pairs(amd, dev, golden_dir, net, monkeypatch):
    """Inference plans run conv2 -> bn3 -> ReLU -> conv3 + skip of every 256 -> 128 -> 128 -> 256 residual (hourglass.py:44-59) as ONE launch
    once the launch fills the chip (awr_conv_args.w2); forced here for every such residual of a batch-2 plan (down to 4x4 maps: 32-pixel
    launches, ragged tiles) and held to the golden joints, the oracle's dense map and the two-launch plan."""
    from awr_amd.trainer import InferEngine
    g = np.load(os.path.join(golden_dir, "%s_fwd.npz" % net))
    img = torch.from_numpy(g["img"])
    J, ks = int(g["J"]), float(g["ks"])
    man = O.manifest_for(net, J)
    outs = {}
    npool = {}
    for fused in (True, "separate_pool", False):
        if fused:
            monkeypatch.setenv("AWR_FUSE2_MIN_WGS", "1")
            monkeypatch.delenv("AWR_NO_FUSE2", raising=False)
            monkeypatch.setenv("AWR_PAIR_POOL", "0" if fused == "separate_pool" else "1")
        else:
            monkeypatch.setenv("AWR_NO_FUSE2", "1")
        m = make_net(amd, net, J, O.procedural_state(man, seed=0))
        m.eval()
        inf = InferEngine(m, img.shape[0], 128, ks, autotune=False)
        names = inf.plan.op_names("fwd")
        npair = sum(1 for n in names if "+conv3" in n)
        npool[fused] = sum(1 for n in names if n == "awr_maxpool_fwd")
        assert (npair >= 12 and sum(1 for n in names if "+conv3+skip_layer" in n) >= 2) if fused else (npair == 0), names
        jt = inf(img.to(dev)).cpu()
        outs[fused] = (jt, inf.plan.dense_map(m.nstage - 1).cpu())
    # round 5: the 2x2 max-pool of a pair's output rides in the pair's launch where the map is wide enough for the 2D workgroup tiles (pool_out):
    # fewer pooling passes, and -- every pixel accumulates in the same k order whichever tile holds it, the windows compare in maxpool_fwd's
    # order -- the SAME BITS as the plan with the separate passes
    assert npool[True] < npool["separate_pool"] == npool[False], npool
    assert torch.equal(outs[True][0], outs["separate_pool"][0]) and torch.equal(outs[True][1], outs["separate_pool"][1])
    oracle = O.backbone_forward(net, O.procedural_state(man, seed=0), img, training=False)
    gaps = oracle_fp64_joint_gap(net, O.procedural_state(man, seed=0), img, ks, False)
    s = len(oracle) - 1
    scale = max(1.0, float(oracle[s].abs().max()))
    assert float((outs[True][1] - oracle[s]).abs().max()) / scale <= 2e-4
    assert float((outs[True][1] - outs[False][1]).abs().max()) / scale <= 2e-5
    assert_joints("%s/eval_fused_pairs/stage%d" % (net, s), outs[True][0].numpy(), g["eval_s%d_jt" % s], gaps[s])


@pytest.mark.parametrize("net", ["resnet_18", "hourglass_1"])
def test_single_image_and_odd_batches(amd, dev, net):
    """test.py:67-86 feeds whatever the loader's last batch holds; train.py keeps the ragged last batch (drop_last = False).  Batch 1 and
    batch 3 plans (one / three 64-pixel tiles at the innermost Hourglass level, split-K inference launches) against the oracle: eval joints of
    every image equal the joints of the same image evaluated alone, and one train step at batch 1 matches the oracle's loss and joints."""
    from awr_amd.trainer import InferEngine, TrainEngine
    J = 14
    ks = 1.0 if net.startswith("resnet") else 0.4
    man = O.manifest_for(net, J)
    img, jt_gt = O.synth_batch(3, 128, J, seed=61)
    sd = O.procedural_state(man, seed=6)
    ref = O.offset2joint_softmax(O.backbone_forward(net, O.procedural_state(man, seed=6), img, training=False)[-1], img, ks)
    gap = oracle_fp64_joint_gap(net, O.procedural_state(man, seed=6), img, ks, False)[-1]
    m = make_net(amd, net, J, sd)
    jt3 = InferEngine(m, 3, 128, ks, autotune=False)(img.to(dev)).cpu()
    assert_joints("%s/infer_b3" % net, jt3.numpy(), ref.numpy(), gap)
    inf1 = InferEngine(m, 1, 128, ks, autotune=False)
    for i in range(3):
        jt1 = inf1(img[i:i + 1].to(dev)).cpu()
        assert_joints("%s/infer_b1/img%d" % (net, i), jt1.numpy(), ref[i:i + 1].numpy(), gap)
        d13 = (jt1 - jt3[i:i + 1]).norm(dim=-1) * 150.0      # tiles and split-K depths differ with the batch: the summation order does too
        assert float(d13.mean()) <= 1e-3 and float(d13.max()) <= 5e-3
    # one optimisation step on a single image
    mt = make_net(amd, net, J, O.procedural_state(man, seed=6))
    eng = TrainEngine(mt, 1, 128, ks, coord_weight=1.0, autotune=False)
    losses, jt = eng.step(img[:1].to(dev), jt_gt[:1].to(dev))
    sdo, ost = O.procedural_state(man, seed=6), {"step": 0, "m": {}, "v": {}}
    out = O.train_step(net, sdo, ost, img[:1], jt_gt[:1], ks, 1.0, 1.0)
    lo = float(out[0]) if isinstance(out, (tuple, list)) else float(out)
    assert abs(float(losses[2]) - lo) <= 2e-4 * abs(lo)
    # an empty batch and a batch larger than the plan are refused, not padded
    from awr_amd import _lib as L
    with pytest.raises(L.AwrError):
        eng.step(img[:0].to(dev), jt_gt[:0].to(dev))
    with pytest.raises(L.AwrError):
        eng.step(img[:2].to(dev), jt_gt[:2].to(dev))


def test_tuning_cache_round_trip(amd, dev, tmp_path, monkeypatch):
    """AWR_TUNE_CACHE: the per-launch choices of the plan autotuner (tile, split-K target and -- weight gradients -- the algorithm) written by one
    engine are what a second engine of the same shape runs with, without timing anything (engine.Plan.autotune, awr_plan_set_gemm[_algo])."""
    import json
    from awr_amd.trainer import TrainEngine
    cache = tmp_path / "tune.json"
    monkeypatch.setenv("AWR_TUNE_CACHE", str(cache))
    J, ks, B = 14, 1.0, 4
    man = O.manifest_for("resnet_18", J)
    img, jt_gt = O.synth_batch(B, 128, J, seed=11)
    tuned = []
    for it in range(2):
        m = make_net(amd, "resnet_18", J, O.procedural_state(man, seed=3))
        eng = TrainEngine(m, B, 128, ks, coord_weight=1.0, use_graph=False)
        losses, _ = eng.step(img.to(dev), jt_gt.to(dev))
        assert torch.isfinite(losses).all()
        tuned.append({k: (tuple(v[0]), v[1]) for k, v in eng.plan.tuned.items()})
    ent = json.load(open(cache))
    assert len(ent) == 1
    stored = next(iter(ent.values()))
    assert set(stored) == set(tuned[0]) and all(len(v[0]) == 4 for v in stored.values())
    assert tuned[0] == tuned[1]
    algos = {v[0][3] for k, v in tuned[1].items() if k.startswith("awr_conv_wgrad:")}
    assert algos and algos <= {0, 1, 2, 3} and (algos & {1, 3})


@study_only
@pytest.mark.parametrize("net,streams", [("resnet_18", 2), ("resnet_18", 0), ("hourglass_1", 2)])
def test_half_batch_batchnorm_backward_wavefront(amd, dev, net, streams, monkeypatch):
    """Round 5: where half a batch still fills the chip the BatchNorm-backward apply is written half by half -- half A on the
    issuing chain, half B on the stream the layer's weight gradient takes -- and the data gradient runs as two half-batch parts, the first beside
    half B's pass (csrc/awr_net.hip: bn_bwd / conv_bwd, OP_HALFWAIT; awr_conv_gemm_part).  Forced on for every eligible layer of the batch-4 plans
    here and compared, in deterministic mode, with the plan that applies and differentiates the whole batch at once: same losses, same gradients
    (the fp64 statistics slots are summed in another order: not bit for bit), with and without side streams."""
    from awr_amd.trainer import TrainEngine
    J = 14
    ks = 1.0 if net.startswith("resnet") else 0.4
    img, jt_gt = O.synth_batch(4, 128, J, seed=83)
    man = O.manifest_for(net, J)
    amd.set_deterministic(True)
    try:
        res = []
        for half_min in ("0", "1"):
            monkeypatch.setenv("AWR_HALF_BNB_MIN_ROWS", half_min)
            m = make_net(amd, net, J, O.procedural_state(man, seed=8))
            eng = TrainEngine(m, 4, 128, ks, coord_weight=1.0, use_graph=False, autotune=False, wgrad_streams=streams)
            names = eng.plan.op_names("bwd")
            n_wait = sum(1 for n in names if n == "__halfwait__")
            n_half = sum(1 for n in names if n.startswith("awr_conv_dgrad:") and n.endswith("/b"))
            assert (n_wait > 0 and n_half == n_wait) if half_min == "1" else (n_wait == 0 and n_half == 0), (half_min, n_wait, n_half)
            for it in range(2):
                losses, jt = eng.step(img.to(dev), jt_gt.to(dev))
                torch.cuda.synchronize()
                if it == 0:
                    g0 = m.flat_grads()[:m.n_active].clone()
            res.append((losses.clone(), g0, jt.clone(), n_wait))
        (l0, g_a, j0, _), (l1, g_b, j1, n_wait) = res
        d = float((g_a - g_b).double().norm() / g_a.double().norm())
        assert d < 2e-6, d
        assert float((l0 - l1).abs().max()) <= 1e-5 * float(l0.abs().max()) and float((j0 - j1).abs().max()) < 1e-5
        report("%s/half_batch_wavefront/streams%d/layers" % (net, streams), n_wait)
        report("%s/half_batch_wavefront/streams%d/grad_rel_diff" % (net, streams), d)
    finally:
        amd.set_deterministic(False)
