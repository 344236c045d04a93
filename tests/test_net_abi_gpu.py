"""GPU: one full ResNet18-deconv optimisation step driven through the NETWORK-LEVEL C ABI alone (include/awr_hip.h: awr_net_*,
awr_plan_*, head / loss / optimiser entry points) -- ctypes on raw device pointers, torch only as the allocator -- the way a
non-Python host would use libawr_hip.so.  Checked against the oracle and against the golden vectors."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import awr_oracle as O

pytestmark = pytest.mark.gpu


def test_resnet18_train_step_through_the_c_abi_only(golden_dir):
    import awr_amd  # noqa: F401
    from awr_amd import _lib as L
    lib = L.lib
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "resnet_18_train.npz"))
    img_h, jt_h = torch.from_numpy(g["img"]), torch.from_numpy(g["jt_gt"])
    B, J, H, F, ks = img_h.shape[0], int(g["J"]), 128, 64, float(g["ks"])

    def ok(rc):
        assert rc == 0, L.last_error()

    net = C.c_void_p()
    ok(lib.awr_net_create(0, 1, J, 2, C.byref(net)))
    nt, npar, nact, nbuf, ncnt, nst = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
    ok(lib.awr_net_sizes(net, C.byref(nt), C.byref(npar), C.byref(nact), C.byref(nbuf), C.byref(ncnt), C.byref(nst)))
    # arenas: parameters / gradients / BatchNorm buffers, filled from a state_dict through awr_net_tensor_info (the checkpoint layout)
    params, grads, bufs = torch.zeros(npar.value, device=dev), torch.zeros(npar.value, device=dev), torch.zeros(nbuf.value, device=dev)
    sd = O.procedural_state(O.manifest_for("resnet_18", J), seed=1)
    key, kd, nd, off, un = C.c_char_p(), C.c_int(), C.c_int(), C.c_int64(), C.c_int()
    shape = (C.c_int64 * 4)()
    where = {}
    for i in range(nt.value):
        ok(lib.awr_net_tensor_info(net, i, C.byref(key), C.byref(kd), C.byref(nd), shape, C.byref(off), C.byref(un)))
        k = key.value.decode()
        if kd.value == 7:
            continue
        arena = params if kd.value <= 4 else bufs
        n = sd[k].numel()
        arena[off.value:off.value + n].copy_(sd[k].reshape(-1))
        where[k] = (kd.value, off.value, n)
    ok(lib.awr_net_bind(net, params.data_ptr(), grads.data_ptr(), bufs.data_ptr()))
    # one training plan: boundary tensors are the caller's
    img, out, gout = img_h.to(dev), torch.zeros(B, 4 * J, F, F, device=dev), torch.zeros(B, 4 * J, F, F, device=dev)
    outs, gouts = (C.c_void_p * 1)(out.data_ptr()), (C.c_void_p * 1)(gout.data_ptr())
    plan = C.c_void_p()
    ok(lib.awr_plan_create(net, B, H, 1, 1, 1, 1, img.data_ptr(), outs, gouts, C.byref(plan)))
    s = torch.cuda.current_stream().cuda_stream
    jt_gt, jt, stat, g_jt = jt_h.to(dev), torch.zeros(B, J, 3, device=dev), torch.zeros(B, J, 2, device=dev), torch.zeros(B, J, 3, device=dev)
    acc, losses = torch.zeros(2, device=dev, dtype=torch.float64), torch.zeros(3, device=dev)
    m, v = torch.zeros(nact.value, device=dev), torch.zeros(nact.value, device=dev)
    ok(lib.awr_plan_set_streams(plan, 2, 0))
    for step in (1, 2):
        # train.py:107-131 as ABI calls: repack, forward, head, losses (coord + dense), their gradients, backward, Adam
        ok(lib.awr_plan_refresh_weights(plan, s))
        ok(lib.awr_plan_forward(plan, s))
        ok(lib.awr_head_forward(out.data_ptr(), img.data_ptr(), B, J, F, H, ks, jt.data_ptr(), stat.data_ptr(), s))
        ok(lib.awr_zero_f64(acc.data_ptr(), 2, s))
        ok(lib.awr_dense_loss(out.data_ptr(), jt_gt.data_ptr(), img.data_ptr(), B, J, F, H, ks, 0.01, 1.0, acc.data_ptr() + 8, gout.data_ptr(), 0, s))
        ok(lib.awr_huber(jt.data_ptr(), jt_gt.data_ptr(), B * J * 3, 0.01, 1.0, acc.data_ptr(), g_jt.data_ptr(), 0, s))
        ok(lib.awr_head_backward(out.data_ptr(), img.data_ptr(), jt.data_ptr(), stat.data_ptr(), g_jt.data_ptr(), B, J, F, H, ks, gout.data_ptr(), 1, s))
        ok(lib.awr_loss_finalize(acc.data_ptr(), 2, losses.data_ptr(), s))
        ok(lib.awr_plan_backward(plan, s))
        if step == 1:
            torch.cuda.synchronize()
            loss0, jt0, grads0 = float(losses[2]), jt.cpu().numpy().copy(), grads.clone()
        ok(lib.awr_adam_step(params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), nact.value, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, 1.0, s))
    torch.cuda.synchronize()
    # golden (reference autograd, tag c1 = coord_weight 1)
    assert abs(loss0 - float(g["c1_loss0"])) <= 2e-4 * abs(float(g["c1_loss0"]))
    assert float(np.abs(jt0 - g["c1_jt0"]).max()) * 150 <= 5e-3
    pkeys = [str(k) for k in g["pkeys"]]
    gmax = float(np.max(g["c1_grad_l2"]))
    for i, k in enumerate(pkeys):
        kd_, o_, n_ = where[k]
        got = float(grads0[o_:o_ + n_].double().norm())
        assert abs(got - float(g["c1_grad_l2"][i])) <= 5e-3 * (float(g["c1_grad_l2"][i]) + 1e-3 * gmax), k
    assert abs(float(losses[2]) - float(g["c1_loss1"])) <= 2e-2 * abs(float(g["c1_loss1"]))
    # BatchNorm running statistics moved (momentum 0.1, twice) and match the oracle after two steps
    sdo, ost = O.procedural_state(O.manifest_for("resnet_18", J), seed=1), {"step": 0, "m": {}, "v": {}}
    for _ in range(2):
        O.train_step("resnet_18", sdo, ost, img_h, jt_h, ks, 1.0, 1.0)
    for k in ("pre.1.running_mean", "layer4.1.bn2.running_var", "deconv_layers.7.running_mean"):
        kd_, o_, n_ = where[k]
        np.testing.assert_allclose(bufs[o_:o_ + n_].cpu().numpy(), sdo[k].numpy(), rtol=3e-3, atol=3e-4)
    # timed replay + autotune entry points
    nbytes, det, nf, nb, nbk, ng, nbn = C.c_int64(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ok(lib.awr_plan_info(plan, C.byref(nbytes), C.byref(det), C.byref(nf), C.byref(nb), C.byref(nbk), C.byref(ng), C.byref(nbn)))
    assert nbytes.value > 1 << 20 and nbn.value == 23 and nbk.value == 1 and ng.value > 60
    ms = (C.c_float * nf.value)()
    ok(lib.awr_plan_run_timed(plan, 0, s, ms))
    assert sum(ms) > 0
    # every launch carries an event pair (fills / copies / markers report 0): GEMM family AND the element-wise launches between them
    name, macs, flags = C.c_char_p(), C.c_double(), C.c_int()
    timed = {}
    for i in range(nf.value):
        ok(lib.awr_plan_op(plan, 0, i, C.byref(name), C.byref(macs), C.byref(flags)))
        if ms[i] > 0:
            timed.setdefault(name.value.decode().split(":")[0], []).append(ms[i])
    assert len(timed["awr_conv_gemm"]) >= 20 and len(timed["awr_bn_finalize"]) >= 20 and len(timed["awr_bn_apply"]) >= 8 and "awr_stem_pool" in timed
    ok(lib.awr_plan_autotune(plan, 1, s))
    # what a tuning cache stores per GEMM launch: tile, split-K target AND (weight gradients) the algorithm; presetting them is accepted,
    # an algorithm the launch cannot run is refused
    nm, tm, tn, tb, us, tuned, algo = C.c_char_p(), C.c_int(), C.c_int(), C.c_int(), C.c_float(), C.c_int(), C.c_int()
    seen = set()
    for i in range(ng.value):
        ok(lib.awr_plan_gemm(plan, i, C.byref(nm), C.byref(tm), C.byref(tn), C.byref(tb), C.byref(us), C.byref(tuned)))
        ok(lib.awr_plan_gemm_algo(plan, i, C.byref(algo)))
        kind = nm.value.decode().split(":")[0]
        if tuned.value:
            assert tm.value in (1, 2) and tn.value in (1, 2) and us.value > 0
            ok(lib.awr_plan_set_gemm(plan, i, tm.value, tn.value, tb.value, us.value))
            ok(lib.awr_plan_set_gemm_algo(plan, i, algo.value))
        if kind == "awr_conv_wgrad":
            assert algo.value in (0, 1, 2, 3)
            seen.add(algo.value)
            if nm.value.decode().endswith("layer2.0.conv1"):      # 3x3 stride 2: the kernel-row kernel does not serve it
                assert lib.awr_plan_set_gemm_algo(plan, i, 3) != 0 and b"algo 3" in lib.awr_last_error()
        else:
            assert algo.value == 0 and lib.awr_plan_set_gemm_algo(plan, i, 1) != 0
    assert 3 in seen or 1 in seen
    ok(lib.awr_plan_destroy(plan))
    ok(lib.awr_net_destroy(net))


def test_hourglass2_inference_through_the_c_abi_only(golden_dir):
    """Stacked hourglass (kind 1, two stages, J = 21) in eval mode through awr_net_* / awr_plan_* on raw pointers: both stages' dense
    maps against the reference-generated golden samples and the joints against the golden joints."""
    import awr_amd  # noqa: F401
    from awr_amd import _lib as L
    lib = L.lib
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "hourglass_2_fwd.npz"))
    img_h = torch.from_numpy(g["img"])
    B, J, H, F, ks = img_h.shape[0], int(g["J"]), 128, 64, float(g["ks"])

    def ok(rc):
        assert rc == 0, L.last_error()

    net = C.c_void_p()
    ok(lib.awr_net_create(1, 2, J, 2, C.byref(net)))
    nt, npar, nact, nbuf, ncnt, nst = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
    ok(lib.awr_net_sizes(net, C.byref(nt), C.byref(npar), C.byref(nact), C.byref(nbuf), C.byref(ncnt), C.byref(nst)))
    assert nst.value == 2 and nact.value < npar.value             # the never-used skip_layer convs sit behind n_active
    params, grads, bufs = torch.zeros(npar.value, device=dev), torch.zeros(npar.value, device=dev), torch.zeros(nbuf.value, device=dev)
    sd = O.procedural_state(O.manifest_for("hourglass_2", J), seed=0)
    key, kd, nd, off, un = C.c_char_p(), C.c_int(), C.c_int(), C.c_int64(), C.c_int()
    shape = (C.c_int64 * 4)()
    seen = 0
    for i in range(nt.value):
        ok(lib.awr_net_tensor_info(net, i, C.byref(key), C.byref(kd), C.byref(nd), shape, C.byref(off), C.byref(un)))
        k = key.value.decode()
        if kd.value == 7:
            continue
        assert tuple(shape[:nd.value]) == tuple(sd[k].shape), k
        (params if kd.value <= 4 else bufs)[off.value:off.value + sd[k].numel()].copy_(sd[k].reshape(-1))
        seen += 1
    assert seen == sum(1 for k in sd if not k.endswith("num_batches_tracked"))
    ok(lib.awr_net_bind(net, params.data_ptr(), grads.data_ptr(), bufs.data_ptr()))
    img = img_h.to(dev)
    out = [torch.zeros(B, 4 * J, F, F, device=dev) for _ in range(2)]
    outs = (C.c_void_p * 2)(out[0].data_ptr(), out[1].data_ptr())
    plan = C.c_void_p()
    ok(lib.awr_plan_create(net, B, H, 0, 3, 1, 1, img.data_ptr(), outs, None, C.byref(plan)))
    s = torch.cuda.current_stream().cuda_stream
    ok(lib.awr_plan_set_streams(plan, 2, 0))
    ok(lib.awr_plan_refresh_weights(plan, s))
    ok(lib.awr_plan_forward(plan, s))
    jt = torch.zeros(B, J, 3, device=dev)
    for st in range(2):
        ok(lib.awr_head_forward(out[st].data_ptr(), img.data_ptr(), B, J, F, H, ks, jt.data_ptr(), None, s))
        torch.cuda.synchronize()
        ref = g["eval_s%d_val" % st]
        got = out[st].cpu().reshape(-1).numpy()[g["eval_s%d_idx" % st]]
        assert float(np.abs(got - ref).max()) <= 2e-4 * max(1.0, float(np.abs(ref).max()))
        d = np.linalg.norm(jt.cpu().numpy().astype(np.float64) - g["eval_s%d_jt" % st], axis=-1) * 150.0
        assert float(d.mean()) <= 2e-2, (st, float(d.mean()))      # (stage 1 of these procedural weights is ill-conditioned: test_nets_gpu.py holds the yardstick)
    ok(lib.awr_plan_destroy(plan))
    ok(lib.awr_net_destroy(net))
