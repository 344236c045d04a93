#!/usr/bin/env python3
"""Layer-by-layer parity localiser (GPU): where does the HIP path leave the float64 evaluation of the oracle's formulas?

    python tools/diag_parity.py --net hourglass_1 --cw 0 [--seed 23] [--streams 2] [--top 12] [--out gpurun_out/diag.txt]

Runs ONE fused train step (TrainEngine, batch 2, reference-initialised weights -- the set-up of
tests/test_nets_gpu.py::test_gradients_elementwise_against_the_fp64_yardstick), reads every activation / gradient buffer of the
static plan through awr_plan_tensor (engine.Plan.tensors) and prints, in forward order, the relative L2 distance from float64 of
(a) the HIP tensors, (b) the fp32 oracle's tensors: raw conv outputs, their gradients, and the parameter gradients.  The float64
run also locates ReLU kinks (elements whose pre-activation is within 1e-6 of zero, with the share of the gradient norm they carry), and
the last table repeats the parameter gradients against float64 evaluated with each implementation's own ReLU decisions (tests/yardstick.py).
Bisecting hooks: AWR_NO_DUAL=1 (conv3 + skip_layer as two launches), --streams 0 (no side / branch streams), --det.
The oracle is test infrastructure: nothing here is imported by the product path.
"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import awr_oracle as O      # noqa: E402
import yardstick as Y       # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="hourglass_1")
    ap.add_argument("--cw", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=23)
    ap.add_argument("--wseed", type=int, default=9)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--det", action="store_true")
    ap.add_argument("--top", type=int, default=0, help="only print the N worst rows of each table (0 = all, forward order)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import awr_amd
    from awr_amd.trainer import TrainEngine
    torch.set_num_threads(max(1, os.cpu_count() // 2))
    if args.det:
        awr_amd.set_deterministic(True)
    net, J, B = args.net, 14, args.batch
    ks = 1.0 if net.startswith("resnet") else 0.4
    img, jt_gt = O.synth_batch(B, 128, J, seed=args.seed)
    sd = O.reference_init_state(net, J, seed=args.wseed)
    ref = Y.trace(net, sd, img, jt_gt, ks, args.cw, True)
    f32 = Y.trace(net, sd, img, jt_gt, ks, args.cw, False)
    r64, g64, loss64, r32, g32, loss32 = ref["acts"], ref["grads"], ref["loss"], f32["acts"], f32["grads"], f32["loss"]
    m = awr_amd.get_deconv_net(18, J, 2) if net.startswith("resnet") else awr_amd.PoseNet(net, J)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    eng = TrainEngine(m, B, 128, ks, coord_weight=args.cw, dense_weight=1.0, lr=1e-3, autotune=False, wgrad_streams=args.streams)
    eng.step(img.cuda(), jt_gt.cuda())
    torch.cuda.synchronize()
    hip = eng.plan.tensors()
    # the float64 yardstick with the ReLU decisions of each implementation (tests/yardstick.py)
    fl32, pl32 = Y.decisions_from_trace(ref, f32)
    flh, plh, rep_h = Y.decisions_from_plan(ref, eng.plan.tensors(lazy=True))
    ref_f32 = Y.trace(net, sd, img, jt_gt, ks, args.cw, True, flips=fl32, pools=pl32)
    ref_hip = Y.trace(net, sd, img, jt_gt, ks, args.cw, True, flips=flh, pools=plh)
    lines = []
    P = lines.append
    P("# %s cw=%g seed=%d wseed=%d B=%d streams=%d det=%d NO_DUAL=%s | loss hip %.9g  f32 %.9g  f64 %.9g" % (
        net, args.cw, args.seed, args.wseed, B, args.streams, int(args.det), os.environ.get("AWR_NO_DUAL", "0"), float(eng.losses[2]), loss32, loss64))
    P("# ReLU kinks of the float64 run (|pre-activation| < 1e-6): (ReLU behind this BatchNorm, elements, share of the gradient norm behind it)")
    for k in Y.kink_table(ref):
        P("#   %-36s n=%d  grad share %.2e" % k)
    P("# ReLU decisions that differ from float64's: HIP %s" % (rep_h,))
    P("#                                    fp32 oracle %s + max-pools %s" % ([(t, int(v.sum()), float(ref["relus"][t][0][v].abs().max())) for t, v in fl32.items() if bool(v.any())], sorted(pl32)))

    def oracle_name(hn):
        """HIP tensor name -> (oracle record name(s) summed, channel count)"""
        base = hn[:-4]      # strip ".out"
        if base in ("final",) or base.startswith("outs."):
            return None
        if base.endswith(".conv3+skip_layer"):
            return [base[:-len(".conv3+skip_layer")] + ".resout"]
        if base.endswith(".conv3.conv"):
            return [base[:-len(".conv3.conv")] + ".resout"]
        return [base + ".out"]

    rows_a, rows_g = [], []
    for hn, (val, grad) in hip.items():
        if not hn.endswith(".out"):
            continue
        on = oracle_name(hn)
        if hn[:-4] == "final" or hn[:-4].startswith("outs."):
            ref64, ref32 = r64["pred"], r32["pred"]
        else:
            if on is None or on[0] not in r64:
                continue
            ref64, ref32 = r64[on[0]], r32[on[0]]
        C = ref64.shape[1]
        hv = val.permute(0, 3, 1, 2)[:, :C].cpu()
        rows_a.append((hn, rel(hv, ref64.detach()), rel(ref32.detach(), ref64.detach()), tuple(ref64.shape)))
        if grad is not None and ref64.grad is not None:
            hg = grad.permute(0, 3, 1, 2)[:, :C].cpu()
            rows_g.append((hn, rel(hg, ref64.grad), rel(ref32.grad, ref64.grad), tuple(ref64.shape)))
    rows_p, rows_q = [], []
    gmax = max(float(g.norm()) for g in g64.values() if g is not None)
    for k, g in g64.items():
        if g is None:
            continue
        rows_p.append((k, rel(m.grad_view(k).cpu(), g), rel(g32[k], g), tuple(g.shape)))
        rows_q.append((k, Y.rel_l2(m.grad_view(k).cpu(), ref_hip["grads"][k], 1e-3 * gmax), Y.rel_l2(g32[k], ref_f32["grads"][k], 1e-3 * gmax), tuple(g.shape)))

    def table(title, rows):
        P("")
        P("## %s  (relative L2 distance from float64: HIP | fp32 oracle | ratio)" % title)
        sel = sorted(rows, key=lambda r: -r[1])[:args.top] if args.top else rows
        for n, eh, eo, shp in sel:
            P("%-44s %9.2e %9.2e %7.2f   %s" % (n, eh, eo, eh / max(eo, 1e-12), "x".join(map(str, shp))))
        if rows:
            P("median ratio %.2f   max HIP %.2e   max fp32-oracle %.2e" % (float(np.median([r[1] / max(r[2], 1e-12) for r in rows])), max(r[1] for r in rows), max(r[2] for r in rows)))
    table("activations (raw conv outputs, forward order)", rows_a)
    table("gradients w.r.t. those tensors (forward order; the backward runs bottom-up)", rows_g)
    table("parameter gradients (state_dict order)", rows_p)
    table("parameter gradients against float64 WITH EACH IMPLEMENTATION'S OWN ReLU DECISIONS (denominator floored at 1e-3 of the largest tensor)", rows_q)
    text = "\n".join(lines)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "a") as f:
            f.write(text + "\n\n")


if __name__ == "__main__":
    main()
