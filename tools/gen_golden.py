#!/usr/bin/env python3
"""DEV-ONLY: pin oracle/awr_oracle.py against the real reference and emit golden vectors.

Runs only where /root/reference exists (the build container).  It imports the reference's
Python modules in-process (cv2 stubbed: util/feature_tool.py:7 imports it but never uses it),
asserts that every oracle function reproduces the reference on seeded inputs, and writes small
array-only fixtures to tests/golden/.  Nothing from the reference (source, bytecode, pickles)
is written -- fixtures hold inputs and expected outputs only.

    python tools/gen_golden.py            # check + (re)write fixtures
"""
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(REPO, "oracle"))
import awr_oracle as O  # noqa: E402


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted at %s -- golden vectors can only be regenerated in the dev container" % REF)
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("Agg")
    from model.resnet_deconv import get_deconv_net
    from model.hourglass import PoseNet
    from model.loss import My_SmoothL1Loss
    from util.feature_tool import FeatureModule
    from util.eval_tool import EvalUtil
    return get_deconv_net, PoseNet, My_SmoothL1Loss, FeatureModule, EvalUtil


def maxdiff(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


CHECKS = {}      # name -> [worst max|restatement - reference| over all calls, tolerance, number of comparisons]
SELF = {}        # same bookkeeping for comparisons in which the reference was routed through the build's OWN cv2 stand-ins:
                 # they show that the surrounding logic agrees, they do NOT pin the resamplers (build compared with itself)


def check(name, a, b, tol, pinned=True):
    d = maxdiff(a, b)
    print("  %-46s max|oracle-ref| = %.3e  (tol %.1e)%s" % (name, d, tol, "" if pinned else "   [UNPINNED: resampler compared with itself]"))
    assert d <= tol, name
    ent = (CHECKS if pinned else SELF).setdefault(name, [0.0, tol, 0])
    ent[0], ent[2] = max(ent[0], float(d)), ent[2] + 1
    return d


def sample_idx(n, k, stream):
    """k deterministic indices into a flat array of n elements."""
    u = O._hash_uniform(k, 1000 + stream, 7).astype(np.float64) + 0.5
    return np.minimum((u * n).astype(np.int64), n - 1)


def hashed(shape, stream, scale):
    return torch.from_numpy((O._hash_uniform(int(np.prod(shape)), stream, 99) * np.float32(2 * scale)).reshape(shape).copy())


def build_ref_net(net, J, get_deconv_net, PoseNet):
    return get_deconv_net(int(net.split("_")[1]), J, 2) if net.startswith("resnet") else PoseNet(net, J)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    get_deconv_net, PoseNet, RefLoss, RefFM, RefEval = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    fm, crit = RefFM(), RefLoss()
    report = {}

    # ---- (b) checkpoint layout --------------------------------------------------------------
    print("[manifest]")
    man = {}
    for net, J in [("resnet_18", 14), ("resnet_50", 14), ("resnet_101", 14), ("hourglass_1", 14), ("hourglass_2", 21)]:
        ref_sd = build_ref_net(net, J, get_deconv_net, PoseNet).state_dict()
        ours = O.manifest_for(net, J)
        assert [k for k, _, _ in ours] == list(ref_sd.keys()), net
        for (k, shp, kind), (rk, rv) in zip(ours, ref_sd.items()):
            assert tuple(rv.shape) == tuple(shp), (k, rv.shape, shp)
            assert (rv.dtype == torch.int64) == (kind == "counter"), k
        man["%s_J%d" % (net, J)] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in ref_sd.items()]
        print("  %-14s J=%d: %d keys, %d params -- key order, shapes, dtypes identical" %
              (net, J, len(ref_sd), sum(v.numel() for k, v in ref_sd.items() if v.dtype != torch.int64 and "running" not in k)))
    json.dump(man, open(os.path.join(GOLD, "statedict_manifest.json"), "w"))

    # ---- a4/a5 head --------------------------------------------------------------------------
    print("[head a4/a5]")
    for tag, (J, H, ks) in {"j14_ks04": (14, 128, 0.4), "j14_ks10": (14, 128, 1.0), "j21_h256": (21, 256, 0.4)}.items():
        F = H // 2
        img, _ = O.synth_batch(2, H, J, seed=11)
        off = hashed((2, 4 * J, F, F), 3, 0.6)
        off.requires_grad_(True)
        jt_ref = fm.offset2joint_softmax(off, img, ks)
        g_jt = hashed((2, J, 3), 4, 1.0)
        (g_ref,) = torch.autograd.grad((jt_ref * g_jt).sum(), off)
        jt_o = O.offset2joint_softmax(off.detach(), img, ks)
        g_o = O.head_backward(off.detach(), img, ks, g_jt)
        report["head_" + tag] = [check("offset2joint_softmax " + tag, jt_o, jt_ref.detach(), 2e-6),
                                 check("head backward (closed form) " + tag, g_o, g_ref, 2e-6 * float(g_ref.abs().max()) + 1e-9)]
        idx = sample_idx(g_ref.numel(), 4096, 5)
        np.savez_compressed(os.path.join(GOLD, "head_%s.npz" % tag), img=img.numpy(), ks=np.float32(ks), J=J,
                            offset_stream=3, offset_scale=np.float32(0.6), g_jt=g_jt.numpy(), jt=jt_ref.detach().numpy(),
                            g_idx=idx, g_val=g_ref.reshape(-1).numpy()[idx],
                            g_l2=np.float64(g_ref.double().norm()), g_sum=np.float64(g_ref.double().sum()))

    # ---- a6 GT map + round trip ---------------------------------------------------------------
    print("[joint2offset a6]")
    for tag, (J, H, ks) in {"j14_ks04": (14, 128, 0.4), "j14_ks10": (14, 128, 1.0), "j21_h256": (21, 256, 0.4)}.items():
        F = H // 2
        img, jt = O.synth_batch(2, H, J, seed=12)
        ref = fm.joint2offset(jt, img, ks, F)
        ours = O.joint2offset(jt, img, ks, F)
        check("joint2offset " + tag, ours, ref, 1e-6)
        rt_ref = fm.offset2joint_softmax(ref, img, ks)
        check("round trip " + tag, O.offset2joint_softmax(ours, img, ks), rt_ref, 2e-6)
        flat = ref.reshape(-1).numpy()
        nz = np.flatnonzero(flat)
        np.savez_compressed(os.path.join(GOLD, "j2o_%s.npz" % tag), img=img.numpy(), jt=jt.numpy(), ks=np.float32(ks), F=F,
                            nz_idx=nz.astype(np.int64), nz_val=flat[nz], shape=np.array(ref.shape), roundtrip=rt_ref.numpy())
        print("    nnz = %.2f %%" % (100.0 * len(nz) / flat.size))

    # ---- a7 Huber -----------------------------------------------------------------------------
    print("[huber a7]")
    x = hashed((3, 7, 11), 8, 0.03)
    y = hashed((3, 7, 11), 9, 0.03)
    x.view(-1)[:4] = torch.tensor([0.01, -0.01, 0.0, 0.02])
    y.view(-1)[:4] = 0.0
    x.requires_grad_(True)
    l_ref = crit(x, y)
    (gx_ref,) = torch.autograd.grad(l_ref, x)
    check("huber loss", O.huber(x.detach(), y), l_ref.detach(), 1e-9)
    np.savez_compressed(os.path.join(GOLD, "huber.npz"), x=x.detach().numpy(), y=y.numpy(), loss=l_ref.detach().numpy(), gx=gx_ref.numpy())

    # ---- a1-a3 backbones (procedural weights) --------------------------------------------------
    print("[backbones a1-a3]")
    for net, J, H, B in [("resnet_18", 14, 128, 2), ("resnet_50", 14, 128, 2), ("hourglass_1", 14, 128, 2), ("hourglass_2", 21, 128, 1)]:
        man_ = O.manifest_for(net, J)
        img, _ = O.synth_batch(B, H, J, seed=13)
        ks = 1.0 if net.startswith("resnet") else 0.4
        out = {"img": img.numpy(), "J": J, "ks": np.float32(ks)}
        for mode in ("eval", "train"):
            sd = O.procedural_state(man_, seed=0)
            ref = build_ref_net(net, J, get_deconv_net, PoseNet)
            ref.load_state_dict(sd, strict=True)
            ref.train(mode == "train")
            with torch.no_grad():
                r = ref(img)
            r = r if isinstance(r, list) else [r]
            o = O.backbone_forward(net, sd, img, training=(mode == "train"))
            for s, (a, b) in enumerate(zip(o, r)):
                check("%s %s stage %d dense map" % (net, mode, s), a, b, 2e-5 * max(1.0, float(b.abs().max())))
                idx = sample_idx(b.numel(), 8192, 20 + s)
                out["%s_s%d_idx" % (mode, s)] = idx
                out["%s_s%d_val" % (mode, s)] = b.reshape(-1).numpy()[idx]
                out["%s_s%d_l2" % (mode, s)] = np.float64(b.double().norm())
                out["%s_s%d_jt" % (mode, s)] = fm.offset2joint_softmax(b, img, ks).numpy()
            if mode == "train":      # BN running stats after one training forward
                rsd = ref.state_dict()
                bn_keys = [k for k, _, kind in man_ if kind in ("bn_mean", "bn_var")]
                worst = max(maxdiff(sd[k], rsd[k]) / max(1.0, float(rsd[k].abs().max())) for k in bn_keys)
                print("  %-46s max rel diff     = %.3e" % (net + " BN running stats after train fwd", worst))
                assert worst < 1e-5
                pick = [bn_keys[0], bn_keys[1], bn_keys[len(bn_keys) // 2], bn_keys[-2], bn_keys[-1]]
                out["bn_keys"] = np.array(pick)
                for i, k in enumerate(pick):
                    out["bn_%d" % i] = rsd[k].numpy()
        np.savez_compressed(os.path.join(GOLD, "%s_fwd.npz" % net), **out)

    # ---- a8/a9 full train step ------------------------------------------------------------------
    print("[train step a8/a9]")
    for net, J, B in [("resnet_18", 14, 2), ("resnet_50", 14, 2), ("hourglass_1", 14, 2), ("hourglass_2", 14, 1)]:
        man_ = O.manifest_for(net, J)
        pkeys = O.params_of(None, man_)
        img, jt_gt = O.synth_batch(B, 128, J, seed=14)
        ks = 1.0 if net.startswith("resnet") else 0.4
        out = {"img": img.numpy(), "jt_gt": jt_gt.numpy(), "J": J, "ks": np.float32(ks), "pkeys": np.array(pkeys)}
        for cw, dw in [(0.0, 1.0), (1.0, 1.0)]:
            tag = "c%d" % int(cw)
            sd0 = O.procedural_state(man_, seed=1)
            ref = build_ref_net(net, J, get_deconv_net, PoseNet)
            ref.load_state_dict(sd0, strict=True)
            ref.train()
            opt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=0)
            sd = O.procedural_state(man_, seed=1)
            ostate = {"step": 0, "m": {}, "v": {}}
            stacks = 1 if net.startswith("resnet") else int(net.split("_")[-1])
            for it in range(2):
                # the reference step, train.py:107-131, driven with the reference's own objects
                gt = fm.joint2offset(jt_gt, img, ks, 64)
                for stage in range(stacks):
                    pred = ref(img)
                    pred = pred[stage] if isinstance(pred, list) else pred
                    jt = fm.offset2joint_softmax(pred, img, ks)
                    l_coord = cw * crit(jt, jt_gt)
                    l_dense = dw * crit(pred, gt)
                    loss = l_coord + l_dense
                opt.zero_grad()
                loss.backward()
                named = dict(ref.named_parameters())
                if it == 0:
                    gl2 = np.array([float(named[k].grad.double().norm()) if named[k].grad is not None else -1.0 for k in pkeys])
                    gsmp = np.array([float(named[k].grad.reshape(-1)[sample_idx(named[k].numel(), 1, 40 + i)[0]])
                                     if named[k].grad is not None else 0.0 for i, k in enumerate(pkeys)], dtype=np.float32)
                opt.step()
                lo, lco, ldo, grads, jt_o = O.train_step(net, sd, ostate, img, jt_gt, ks, cw, dw, lr=1e-3)
                check("%s %s it%d loss" % (net, tag, it), lo, loss.detach(), 1e-6 * max(1.0, float(loss)))
                if it == 0:
                    nograd_o = sorted(k for k, g in grads.items() if g is None)
                    nograd_r = sorted(k for k in pkeys if named[k].grad is None)
                    assert nograd_o == nograd_r
                    worst = max(maxdiff(grads[k], named[k].grad) / (float(named[k].grad.abs().max()) + 1e-12)
                                for k in pkeys if grads[k] is not None)
                    print("  %-46s max rel grad diff = %.3e ; %d params without grad" % (net + " " + tag, worst, len(nograd_r)))
                    assert worst < 5e-3
                    out[tag + "_loss0"] = np.float32(loss.detach())
                    out[tag + "_lcoord0"] = np.float32(torch.as_tensor(l_coord).detach())
                    out[tag + "_ldense0"] = np.float32(torch.as_tensor(l_dense).detach())
                    out[tag + "_jt0"] = jt.detach().numpy()
                    out[tag + "_grad_l2"] = gl2
                    out[tag + "_grad_smp"] = gsmp
                    out["nograd"] = np.array(nograd_r)
                else:
                    out[tag + "_loss1"] = np.float32(loss.detach())
                psmp = np.array([float(named[k].detach().reshape(-1)[sample_idx(named[k].numel(), 1, 40 + i)[0]])
                                 for i, k in enumerate(pkeys)], dtype=np.float32)
                osmp = np.array([float(sd[k].reshape(-1)[sample_idx(sd[k].numel(), 1, 40 + i)[0]]) for i, k in enumerate(pkeys)], dtype=np.float32)
                # Adam's first steps move every weight by ~lr regardless of gradient scale, so params match to ~1e-6 abs
                check("%s %s params after step %d" % (net, tag, it + 1), osmp, psmp, 2e-4)
                out[tag + "_param_smp%d" % (it + 1)] = psmp
        np.savez_compressed(os.path.join(GOLD, "%s_train.npz" % net), **out)

    # ---- a well-conditioned training-mode fixture: batch 8, weights with the reference's init_weights distributions, -----
    # seeded inputs; held to the PLAIN north_star bar (1e-3 mm mean, 5e-3 max) in every accumulation mode.  KB-sized: the inputs are
    # regenerated from their seeds by the test (a checksum guards the generators), outputs are sampled.
    print("[resnet_18 training mode, batch 8, reference init distributions]")
    net, J, B, ks = "resnet_18", 14, 8, 1.0
    img, jt_gt = O.synth_batch(B, 128, J, seed=31)
    man_ = O.manifest_for(net, J)
    pkeys = O.params_of(None, man_)
    out = {"J": J, "ks": np.float32(ks), "B": B, "img_seed": 31, "w_seed": 21, "img_sum": np.float64(img.double().sum()),
           "w_sum": np.float64(sum(float(v.double().abs().sum()) for v in O.reference_init_state(net, J, seed=21).values())), "pkeys": np.array(pkeys)}
    for cw, dw in [(0.0, 1.0), (1.0, 1.0)]:
        tag = "c%d" % int(cw)
        ref = build_ref_net(net, J, get_deconv_net, PoseNet)
        ref.load_state_dict(O.reference_init_state(net, J, seed=21), strict=True)
        ref.train()
        gt = fm.joint2offset(jt_gt, img, ks, 64)
        pred = ref(img)
        jt = fm.offset2joint_softmax(pred, img, ks)
        l_coord, l_dense = cw * crit(jt, jt_gt), dw * crit(pred, gt)
        loss = l_coord + l_dense
        loss.backward()
        named = dict(ref.named_parameters())
        sd = O.reference_init_state(net, J, seed=21)
        lo, lco, ldo, grads, jt_o = O.train_step(net, sd, {"step": 0, "m": {}, "v": {}}, img, jt_gt, ks, cw, dw, lr=1e-3)
        check("%s b8 %s loss" % (net, tag), lo, loss.detach(), 1e-6 * max(1.0, float(loss)))
        check("%s b8 %s joints" % (net, tag), jt_o, jt.detach(), 2e-6)
        idx = sample_idx(pred.numel(), 4096, 77)
        out[tag + "_loss0"] = np.float32(loss.detach())
        out[tag + "_lcoord0"] = np.float32(torch.as_tensor(l_coord).detach())
        out[tag + "_jt0"] = jt.detach().numpy()
        out["pred_idx"] = idx
        out[tag + "_pred_val"] = pred.detach().reshape(-1).numpy()[idx]
        out[tag + "_grad_l2"] = np.array([float(named[k].grad.double().norm()) for k in pkeys])
        out[tag + "_grad_smp"] = np.array([float(named[k].grad.reshape(-1)[sample_idx(named[k].numel(), 1, 40 + i)[0]]) for i, k in enumerate(pkeys)], dtype=np.float32)
    rsd = ref.state_dict()
    bn_keys = [k for k in rsd if k.endswith("running_mean") or k.endswith("running_var")]
    out["bn_keys"] = np.array(bn_keys)
    out["bn_smp"] = np.array([float(rsd[k].reshape(-1)[sample_idx(rsd[k].numel(), 1, 90 + i)[0]]) for i, k in enumerate(bn_keys)], dtype=np.float32)
    np.savez_compressed(os.path.join(GOLD, "resnet_18_train_b8.npz"), **out)

    # ---- BASELINE config 5 shape: Hourglass-2, J = 21, 256x256 (train.py:116-121 semantics at the stress shape) ----------
    print("[config 5: hourglass_2, J=21, 256x256]")
    net, J, H, B, ks, cw, dw = "hourglass_2", 21, 256, 2, 0.4, 1.0, 1.0
    man_ = O.manifest_for(net, J)
    pkeys = O.params_of(None, man_)
    img, jt_gt = O.synth_batch(B, H, J, seed=15)
    out = {"img": img.numpy(), "jt_gt": jt_gt.numpy(), "J": J, "ks": np.float32(ks), "pkeys": np.array(pkeys)}
    sd_e = O.procedural_state(man_, seed=3)
    ref = build_ref_net(net, J, get_deconv_net, PoseNet)
    ref.load_state_dict(sd_e, strict=True)
    ref.eval()
    with torch.no_grad():
        r = ref(img)
    o = O.backbone_forward(net, sd_e, img, training=False)
    for s_, (a, b) in enumerate(zip(o, r)):
        check("c5 eval stage %d dense map" % s_, a, b, 2e-5 * max(1.0, float(b.abs().max())))
        idx = sample_idx(b.numel(), 8192, 60 + s_)
        out["eval_s%d_idx" % s_], out["eval_s%d_val" % s_] = idx, b.reshape(-1).numpy()[idx]
        out["eval_s%d_jt" % s_] = fm.offset2joint_softmax(b, img, ks).numpy()
    ref.train()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=0)
    sd = O.procedural_state(man_, seed=3)
    ostate = {"step": 0, "m": {}, "v": {}}
    for it in range(2):
        gt = fm.joint2offset(jt_gt, img, ks, H // 2)
        for stage in range(2):                                  # train.py:116-121: two forwards, the last stage's loss survives
            pred = ref(img)[stage]
            jt = fm.offset2joint_softmax(pred, img, ks)
            l_coord = cw * crit(jt, jt_gt)
            l_dense = dw * crit(pred, gt)
            loss = l_coord + l_dense
        opt.zero_grad()
        loss.backward()
        named = dict(ref.named_parameters())
        if it == 0:
            out["grad_l2"] = np.array([float(named[k].grad.double().norm()) if named[k].grad is not None else -1.0 for k in pkeys])
            out["pred_idx"] = sample_idx(pred.numel(), 8192, 70)
            out["pred_val"] = pred.detach().reshape(-1).numpy()[out["pred_idx"]]
        opt.step()
        lo, lco, ldo, grads, jt_o = O.train_step(net, sd, ostate, img, jt_gt, ks, cw, dw, lr=1e-3)
        check("c5 it%d loss" % it, lo, loss.detach(), 1e-6 * max(1.0, float(loss)))
        if it == 0:
            check("c5 joints (last stage, train mode)", jt_o, jt.detach(), 2e-5)
            worst = max(maxdiff(grads[k], named[k].grad) / (float(named[k].grad.abs().max()) + 1e-12) for k in pkeys if grads[k] is not None)
            print("  %-46s max rel grad diff = %.3e" % ("c5 gradients", worst))
            assert worst < 5e-3
            out["loss0"], out["lcoord0"], out["ldense0"] = np.float32(loss.detach()), np.float32(l_coord.detach()), np.float32(l_dense.detach())
            out["jt0"] = jt.detach().numpy().copy()
            rsd = ref.state_dict()
            bn_keys = [k for k, _, kind in man_ if kind in ("bn_mean", "bn_var")]
            pick = [bn_keys[0], bn_keys[1], bn_keys[len(bn_keys) // 2], bn_keys[-2], bn_keys[-1]]
            out["bn_keys"] = np.array(pick)
            for i, k in enumerate(pick):                        # running stats after TWO momentum updates (one per stage forward)
                check("c5 BN running stat " + k, sd[k], rsd[k], 1e-5 * max(1.0, float(rsd[k].abs().max())))
                out["bn_%d" % i] = rsd[k].numpy().copy()        # state_dict() hands out the LIVE buffers: the next iteration updates them in place
            cnt = [k for k, _, kind in man_ if kind == "counter"]
            assert all(int(rsd[k]) == 2 for k in cnt)
            out["num_batches_tracked"] = np.int64(2)
            psmp = np.array([float(named[k].detach().reshape(-1)[sample_idx(named[k].numel(), 1, 40 + i)[0]]) for i, k in enumerate(pkeys)], dtype=np.float32)
            out["param_smp1"] = psmp
        else:
            out["loss1"] = np.float32(loss.detach())
    np.savez_compressed(os.path.join(GOLD, "hourglass_2_c5.npz"), **out)

    # ---- f1 evaluator -----------------------------------------------------------------------------
    print("[evaluator f1]")
    rng = np.random.RandomState(5)
    N, J = 16, 14
    jt_uvd = rng.uniform(-0.8, 0.8, (N, J, 3)).astype(np.float32)
    jt_xyz_gt = rng.uniform(-0.8, 0.8, (N, J, 3)).astype(np.float32)
    center = np.stack([rng.uniform(-100, 100, N), rng.uniform(-100, 100, N), rng.uniform(600, 900, N)], 1).astype(np.float32)
    scale = rng.uniform(0.3, 0.6, N)
    Ms = np.stack([np.array([[s, 0, tx], [0, s, ty], [0, 0, 1]], np.float32) for s, tx, ty in
                   zip(scale, rng.uniform(-120, -40, N), rng.uniform(-100, -20, N))])
    cube = np.tile(np.array([300, 300, 300], np.float32), (N, 1))
    cube[N // 2:] *= 5.0 / 6.0
    ev = RefEval(128, O.NYU_PARAS, O.NYU_FLIP, J)
    for i in range(N):
        ev.feed(jt_uvd[i].copy(), jt_xyz_gt[i], center[i], Ms[i], cube[i])
    mpe, med, auc, pck, th = ev.get_measures()
    errs, uvds = O.joint_errors_mm(jt_uvd, jt_xyz_gt, center, Ms, cube)
    o_mpe, o_med, o_auc, o_pck, _ = O.measures(errs)
    check("EvalUtil MPE", o_mpe, mpe, 1e-4)
    check("EvalUtil AUC", o_auc, auc, 1e-6)
    check("EvalUtil uvd", uvds, np.array(ev.jt_uvd_pred), 1e-3)
    np.savez_compressed(os.path.join(GOLD, "eval_feed.npz"), jt_uvd=jt_uvd, jt_xyz_gt=jt_xyz_gt, center=center, M=Ms, cube=cube,
                        mpe=np.float64(mpe), med=np.float64(med), auc=np.float64(auc), pck=pck,
                        uvd=np.array(ev.jt_uvd_pred, dtype=np.float32))

    # ---- f2 loader helpers that need no cv2 (dataloader/loader.py:88-101, :181-260) -----------------------------
    print("[loader helpers f2]")
    from dataloader.loader import Loader
    from util.util import xyz2uvd as ref_xyz2uvd
    sys.path.insert(0, REPO)
    import awr_amd  # noqa: F401
    from awr_amd import nyu_data as ND
    ld = Loader.__new__(Loader)
    ld.paras, ld.flip, ld.img_size = np.array(ND.PARAS), -1, 128
    rng = np.random.RandomState(7)
    out = {}
    centers_xyz = np.stack([rng.uniform(-200, 200, 6), rng.uniform(-150, 150, 6), rng.uniform(500, 1100, 6)], 1)
    cube = np.array([300.0, 300.0, 300.0])
    depth = rng.uniform(400, 1300, (480, 640)).astype(np.float32)
    depth[rng.rand(480, 640) < 0.3] = 0
    b_ref, m_ref, c_ref, n_ref, j_ref, u_ref = [], [], [], [], [], []
    for c in centers_xyz:
        cuvd = ref_xyz2uvd(c, ld.paras, ld.flip).astype(np.float64)
        check("xyz2uvd", ND.xyz2uvd(c, ND.PARAS, -1), ref_xyz2uvd(c, ld.paras, ld.flip), 1e-4)
        b = ld.center2bounds(cuvd, cube)
        assert ND.center2bounds(cuvd, cube)[:4] == b[:4]
        check("center2bounds z", np.array(ND.center2bounds(cuvd, cube)[4:]), np.array(b[4:]), 1e-9)
        cr = ld.bounds2crop(depth.copy(), *b)
        check("bounds2crop", ND.bounds2crop(depth.copy(), *b), cr, 0.0)
        M = ld.center2transmat(cuvd, cube, np.array([128, 128]))
        check("center2transmat", ND.center2transmat(cuvd, cube, np.array([128, 128])), M, 0.0)
        small = cr[:96, :96].astype(np.float32).copy()
        nr = ld.normalize(small.max(), small.copy(), c, cube)
        check("normalize", ND.normalize(small.max(), small.copy(), c, cube), nr, 0.0)
        jt = rng.uniform(100, 500, (14, 3))
        check("transform_jt_uvd", ND.transform_jt_uvd(jt, M), ld.transform_jt_uvd(jt, M), 1e-4)
        u_ref.append(cuvd); b_ref.append(np.array(b)); m_ref.append(M); c_ref.append(cr.shape); n_ref.append(nr); j_ref.append(ld.transform_jt_uvd(jt, M))
    np.savez_compressed(os.path.join(GOLD, "loader_fns.npz"), centers_xyz=centers_xyz, center_uvd=np.array(u_ref), bounds=np.array(b_ref),
                        M=np.array(m_ref), crop_shape=np.array(c_ref), norm=np.array(n_ref), seed=7)

    # ---- f2 augmentation (dataloader/loader.py:53-179): random stream + label / matrix / cube arithmetic --------------------
    # The reference calls cv2.warpPerspective / warpAffine / getRotationMatrix2D; cv2 is not installable here, so the stub module
    # forwards those three to the numpy restatements (nyu_data.py).  What this pins: the RandomState(23455) stream, the choice
    # logic, centre / joint / cube / matrix arithmetic and the post-warp clean-up.  What it cannot pin: the resamplers themselves.
    print("[loader augmentation f2]")
    cv2 = sys.modules["cv2"]
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT, cv2.INTER_NEAREST = 1, 0, 0
    cv2.resize = lambda img, size, interpolation=None: ND.resize_nearest(img, size)
    cv2.warpPerspective = lambda img, M, dsize, flags=None, borderMode=None, borderValue=0: ND.warp_perspective(img, M, dsize, border=borderValue)
    cv2.warpAffine = lambda img, M, dsize, flags=None, borderMode=None, borderValue=0: ND.warp_affine(img, M, dsize, border=borderValue)
    cv2.getRotationMatrix2D = lambda c, a, sc: ND.rotation_matrix_2d(c, a, sc)
    import dataloader.loader as RL
    RL.cv2 = cv2
    ld2 = Loader("unused", "train", 128, "nyu")                # sets RandomState(23455) and aug_ops (loader.py:8-17)
    ld2.paras, ld2.flip = np.array(ND.PARAS), -1
    mine = ND.Augmenter(ND.PARAS, -1)
    draws = []
    rng = np.random.RandomState(11)
    yy, xx = np.mgrid[0:480, 0:640]
    aug_out = {}
    n_ops = {"trans": 0, "scale": 0, "rot": 0, None: 0}
    for i in range(12):
        r = ld2.random_aug(10, 0.1, 180)
        m = mine.random_aug(10, 0.1, 180)
        assert r[0] == m[0]
        check("random_aug", np.concatenate([m[1], [m[2], m[3]]]), np.concatenate([r[1], [r[2], r[3]]]), 0.0)
        draws.append(np.concatenate([[ld2.aug_ops.index(r[0])], r[1], [r[2], r[3]]]))
        c_xyz = np.array([rng.uniform(-120, 120), rng.uniform(-90, 90), rng.uniform(600, 900)])
        c_uvd = ref_xyz2uvd(c_xyz, ld2.paras, ld2.flip).astype(np.float64)
        depth = np.full((480, 640), 1400.0, np.float32)
        hand = (xx - c_uvd[0]) ** 2 + (yy - c_uvd[1]) ** 2 < (60 * 750.0 / c_xyz[2]) ** 2
        depth[hand] = (c_xyz[2] + 0.25 * (xx[hand] - c_uvd[0]) - 0.15 * (yy[hand] - c_uvd[1])).astype(np.float32)
        cube = np.array([300.0, 300.0, 300.0])
        jt = rng.uniform(-100, 100, (14, 3))
        img_r, M_r = ld2.crop(depth.copy(), c_uvd, cube, np.array([128, 128]))
        img_m, M_m = ND.crop(depth.copy(), c_uvd, cube, np.array([128, 128]))
        check("crop (resize rule shared)", img_m, img_r, 0.0, pinned=False)
        check("crop M", M_m, M_r, 0.0)
        out_r = ld2.augment(img_m.copy(), jt.copy(), c_uvd.copy(), cube.copy(), M_m.copy(), *r)
        out_m = mine.augment(img_m.copy(), jt.copy(), c_uvd.copy(), cube.copy(), M_m.copy(), *m)
        for name, a_, b_ in zip(("img", "jt_xyz", "cube", "center", "M"), out_m, out_r):
            # the warped IMAGE went through nyu_data's own resamplers on both sides; labels / cube / centre / matrix did not
            check("augment/%s/%s" % (r[0], name), np.asarray(a_, np.float64), np.asarray(b_, np.float64), 0.0, pinned=(name != "img"))
        n_ops[r[0]] += 1
        aug_out["case%d" % i] = np.concatenate([np.asarray(out_r[1], np.float64).ravel(), np.asarray(out_r[2], np.float64).ravel(),
                                                np.asarray(out_r[3], np.float64).ravel(), np.asarray(out_r[4], np.float64).ravel()])
        aug_out["imgsum%d" % i] = np.float64(np.asarray(out_r[0], np.float64).sum())
    assert all(v > 0 for v in n_ops.values()), n_ops
    np.savez_compressed(os.path.join(GOLD, "loader_aug.npz"), draws=np.array(draws), seed=11, **aug_out)

    report["every_comparison_restatement_vs_reference"] = {k: {"max_abs_diff": v[0], "tolerance": v[1], "comparisons": v[2]} for k, v in sorted(CHECKS.items())}
    report["unpinned"] = {
        "what": "cv2.resize(INTER_NEAREST) / cv2.warpAffine / cv2.warpPerspective (dataloader/loader.py:19-51, :53-179) are restated in "
                "awr_amd/nyu_data.py from OpenCV's documented fixed-point sampling rule; cv2 cannot be installed here, so NO cv2-produced "
                "vector exists and these three resamplers are UNPINNED.  The entries below ran the reference's augmentation with its cv2 calls "
                "forwarded to those restatements, i.e. they compare the build with itself (they pin the surrounding logic only).  "
                "tests/test_nyu_data_cpu.py holds analytic known-answer cases (integer shifts, 90/180 degree turns, exact x2 scale, identity) "
                "that any correct nearest / bilinear implementation must reproduce bit for bit.  Round 3: CROSS-CHECKED against independent "
                "implementations -- resize_nearest against a literal scalar restatement of OpenCV's resizeNN index rule (fx = dst/src, ifx = 1/fx, "
                "floor(i*ifx): the division order matters for ~5 % of size pairs), warp_affine / warp_perspective against "
                "scipy.ndimage.map_coordinates(order=1, mode='grid-constant') at the exact inverse-mapped coordinates within the 1/32-pixel "
                "fixed-point bound L*(1/64+1/64) (tests/test_nyu_data_cpu.py::test_bilinear_warps_cross_checked_against_scipy).  "
                "Status: cross-checked, cv2's own rounding unverified.",
        "cross_check_bound": "max |delta| <= L * (2/64 + 2/1024) (affine) / L * 2/64 (perspective), L = largest jump between neighbouring samples incl. the border value",
        "functions": ["nyu_data.resize_nearest", "nyu_data.warp_affine", "nyu_data.warp_perspective"],
        "self_comparisons": {k: {"max_abs_diff": v[0], "tolerance": v[1], "comparisons": v[2]} for k, v in sorted(SELF.items())},
    }
    json.dump(report, open(os.path.join(GOLD, "pin_report.json"), "w"), indent=1)
    print("golden vectors written to", GOLD)
    os.system("du -sh %s" % GOLD)


if __name__ == "__main__":
    main()
