"""float64 yardstick for gradient parity (test infrastructure; imported by tests/test_nets_gpu.py and tools/diag_parity.py only).

`trace` evaluates the oracle's formulas (oracle/awr_oracle.py) for one train-step loss in float64 or float32 with every conv output
and every ReLU recorded.  ReLU is the only non-differentiable point of the loss that a rounding-level perturbation can cross: an
element whose pre-activation is ~1e-6 gets derivative 0 in one fp32 implementation and 1 in another, and everything upstream of it in
the backward inherits the difference.  Instead of widening tolerances by a "kink noise floor", the yardstick is evaluated WITH THE
ReLU DECISIONS OF THE IMPLEMENTATION UNDER TEST: `flips` marks the elements whose derivative is to be taken from the other side
(the forward value is untouched), so `trace(..., flips=decisions_of(hip))` is the float64 gradient of exactly the piecewise-linear
branch the HIP step differentiated.  What remains is rounding error proper, which can be held to a multiple of the fp32 oracle's own.
"""
import torch

import awr_oracle as O


def trace(net, sd, img, jt_gt, ks, cw, f64=True, flips=None):
    """One forward + backward of the oracle.  Returns dict(acts, grads, relus, loss):
    acts  {name: tensor}    raw conv outputs "<layer>.out", hourglass residual outputs "<block>.resout", "pred" (.grad retained)
    grads {param key: grad}
    relus {tag: (pre-activation (detached), gradient w.r.t. the ReLU output)}, tag = the BatchNorm prefix feeding the ReLU
          ("layer1.0.bn1", "pre.1.bn2", ...; "<bn2 prefix>+res" for a ResNet block output relu(bn2(.) + identity))
    flips {tag: bool mask}  elements whose ReLU derivative is flipped (value unchanged)."""
    TF = O.TF
    acts, relus, tags, alive = {}, {}, {}, []
    last_bn = [None]
    o_conv, o_convt, o_relu, o_res, o_bn = TF.conv2d, TF.conv_transpose2d, TF.relu, O._hg_res, O._bn
    dt = torch.float64 if f64 else torch.float32
    work = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    pkeys = O.params_of(work, O.manifest_for(net, jt_gt.shape[1]))
    leaves = {k: work[k].detach().clone().requires_grad_(True) for k in pkeys}
    for k in pkeys:
        work[k] = leaves[k]
    names = {id(v): k for k, v in work.items()}

    def keep(name, y):
        if y.requires_grad:
            y.retain_grad()
        acts[name] = y
        return y

    def conv(x, w, *a, **k):
        return keep(names[id(w)][:-len(".weight")] + ".out", o_conv(x, w, *a, **k))

    def convt(x, w, *a, **k):
        return keep(names[id(w)][:-len(".weight")] + ".out", o_convt(x, w, *a, **k))

    def bn(sd_, p, x, training):
        y = o_bn(sd_, p, x, training)
        tags[id(y)] = p
        alive.append(y)                          # ids must not be recycled while the trace runs
        if "downsample" not in p:
            last_bn[0] = p
        return y

    def relu(x, *a, **k):
        tag = tags.get(id(x)) or (last_bn[0] + "+res")
        y = o_relu(x)
        if flips is not None and tag in flips and bool(flips[tag].any()):
            m = flips[tag]
            sign = torch.where(x.detach() > 0, -torch.ones_like(x), torch.ones_like(x))      # derivative 1 -> 0 / 0 -> 1
            y = y + torch.where(m, (x - x.detach()) * sign, torch.zeros_like(x))
        if x.requires_grad:
            y.retain_grad()
            relus[tag] = (x, y)
        return y

    def hg_res(sd_, p, x, training):
        return keep(p + ".resout", o_res(sd_, p, x, training))

    TF.conv2d, TF.conv_transpose2d, TF.relu, O._hg_res, O._bn = conv, convt, relu, hg_res, bn
    O.HIGH_PRECISION = bool(f64)
    try:
        im, jg = img.to(dt), jt_gt.to(dt)
        gt = O.joint2offset(jg, im, ks, img.shape[-1] // 2)
        pred = keep("pred", O.backbone_forward(net, work, im, True)[-1])
        loss = cw * O.huber(O.offset2joint_softmax(pred, im, ks), jg) + O.huber(pred, gt)
        loss.backward()
    finally:
        TF.conv2d, TF.conv_transpose2d, TF.relu, O._hg_res, O._bn = o_conv, o_convt, o_relu, o_res, o_bn
        O.HIGH_PRECISION = False
    return {"acts": acts, "grads": {k: leaves[k].grad for k in pkeys}, "loss": float(loss.detach()),
            "relus": {t: (x.detach(), y.grad) for t, (x, y) in relus.items()}}


def kink_table(ref, tol=1e-6):
    """[(tag, n elements with |pre-activation| < tol, share of the gradient norm behind that ReLU they carry)] of a float64 trace."""
    out = []
    for tag, (x, g) in ref["relus"].items():
        near = x.abs() < tol
        if g is not None and bool(near.any()):
            out.append((tag, int(near.sum()), float((g * near).norm() / (g.norm() + 1e-300))))
    return out


def decisions_from_trace(ref64, other):
    """flips that make the float64 trace differentiate the branch `other` (a float32 trace of the same inputs) took."""
    flips = {}
    for tag, (x64, _) in ref64["relus"].items():
        x = other["relus"][tag][0]
        flips[tag] = (x > 0) != (x64 > 0)
    return flips


def decisions_from_plan(ref64, plan_tensors):
    """flips that make the float64 trace differentiate the branch the HIP step took.  plan_tensors = engine.Plan.tensors(lazy=True):
    "<bn>.act" holds a materialised [relu](bn(.) [+ res]) (its sign is the decision), "<bn>.act(lazy)" the raw conv output plus the
    per-channel (scale, shift) the loaders apply.  ReLUs the plan never materialises in any form (the fused ResNet stem) keep the
    float64 decisions.  Returns (flips, report) with report = [(tag, n flipped, max |float64 pre-activation| among them)]."""
    flips, report = {}, []
    for tag, (x64, _) in ref64["relus"].items():
        base = tag[:-4] if tag.endswith("+res") else tag
        cand = [base + ".act", base + ".act(lazy)"]
        if base.endswith(".bn") and base[:-3] + ".conv.act" in plan_tensors:      # hourglass stem: conv + bias -> bn -> relu, fused kernel
            cand = [base[:-3] + ".conv.act"]
        ent = next((plan_tensors[c] for c in cand if c in plan_tensors), None)
        if ent is None:
            continue
        if len(ent) == 2:                      # materialised, post-ReLU
            on = ent[0] > 0
        else:                                  # lazy: value * scale + shift (what the consumers' loaders compute)
            val, _, sc, sh, _relu = ent
            on = torch.addcmul(sh, val, sc) > 0
        on = on.permute(0, 3, 1, 2).cpu()
        m = on != (x64 > 0)
        flips[tag] = m
        if bool(m.any()):
            report.append((tag, int(m.sum()), float(x64[m].abs().max())))
    return flips, report


def rel_l2(a, b, floor=0.0):
    return float((a.double().reshape(-1) - b.double().reshape(-1)).norm() / (b.double().norm() + floor + 1e-300))
