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


def trace(net, sd, img, jt_gt, ks, cw, f64=True, flips=None, pools=None):
    """One forward + backward of the oracle.  Returns dict(acts, grads, relus, loss):
    acts  {name: tensor}    raw conv outputs "<layer>.out", hourglass residual outputs "<block>.resout", "pred" (.grad retained)
    grads {param key: grad}
    relus {tag: (pre-activation (detached), gradient w.r.t. the ReLU output)}, tag = the BatchNorm prefix feeding the ReLU
          ("layer1.0.bn1", "pre.1.bn2", ...; "<bn2 prefix>+res" for a ResNet block output relu(bn2(.) + identity))
    pools {tag: (input (detached), argmax indices)} of every max-pool, tag = name of the pooled tensor ("pre.1.resout", ...)
    flips {tag: bool mask}  elements whose ReLU derivative is flipped (value unchanged).
    pools (argument) {tag: indices}: max-pools whose gradient is routed to THESE window elements (the other decision a rounding-level
          perturbation can change: two window elements within 1e-6 of each other; the forward value moves by that much, which is noise)."""
    TF = O.TF
    acts, relus, tags, alive, pooled = {}, {}, {}, [], {}
    last_bn = [None]
    o_conv, o_convt, o_relu, o_res, o_bn, o_pool = TF.conv2d, TF.conv_transpose2d, TF.relu, O._hg_res, O._bn, TF.max_pool2d
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

    def pool(x, *a, **k):
        tag = next((n for n, t in acts.items() if t is x), "pool%d" % len(pooled))
        y, idx = o_pool(x, *a, return_indices=True, **k)
        if pools is not None and tag in pools:
            idx = pools[tag]
            y = torch.gather(x.flatten(2), 2, idx.flatten(2)).view(idx.shape)
        if y.requires_grad:
            y.retain_grad()
        pooled[tag] = (x, idx, y, a, k)
        return y

    TF.conv2d, TF.conv_transpose2d, TF.relu, O._hg_res, O._bn, TF.max_pool2d = conv, convt, relu, hg_res, bn, pool
    O.HIGH_PRECISION = bool(f64)
    try:
        im, jg = img.to(dt), jt_gt.to(dt)
        gt = O.joint2offset(jg, im, ks, img.shape[-1] // 2)
        pred = keep("pred", O.backbone_forward(net, work, im, True)[-1])
        loss = cw * O.huber(O.offset2joint_softmax(pred, im, ks), jg) + O.huber(pred, gt)
        loss.backward()
    finally:
        TF.conv2d, TF.conv_transpose2d, TF.relu, O._hg_res, O._bn, TF.max_pool2d = o_conv, o_convt, o_relu, o_res, o_bn, o_pool
        O.HIGH_PRECISION = False
    return {"acts": acts, "grads": {k: leaves[k].grad for k in pkeys}, "loss": float(loss.detach()),
            "relus": {t: (x.detach(), y.grad) for t, (x, y) in relus.items()},
            "pools": {t: (x.detach(), idx) for t, (x, idx, y, a, k) in pooled.items()},
            "pool_near_ties": {t: _near_tie_share(x.detach(), y.grad, a, k) for t, (x, idx, y, a, k) in pooled.items() if not t.endswith(".resout")}}


def _near_tie_share(x, gy, args, kwargs, tol=1e-6):
    """share of the gradient norm behind a max-pool that sits on windows whose two largest DISTINCT-position elements are within
    `tol` (relative to the tensor's magnitude) of each other while not being equal -- equal elements (ReLU zeros, constant background)
    are resolved by position, identically everywhere"""
    if gy is None:
        return 0.0
    ksz = args[0] if args else kwargs.get("kernel_size")
    st = (args[1] if len(args) > 1 else kwargs.get("stride", ksz)) or ksz
    pd = args[2] if len(args) > 2 else kwargs.get("padding", 0)
    cols = TFunfold(x, ksz, st, pd)                     # (B, C, k*k, L)
    top = cols.topk(2, dim=2).values
    gap = top[:, :, 0] - top[:, :, 1]
    near = ((gap > 0) & (gap < tol * float(x.abs().max()))).view(gy.shape)
    return float((gy * near).norm() / (gy.norm() + 1e-300))


def TFunfold(x, k, s, p):
    B, C, H, W = x.shape
    xp = torch.nn.functional.pad(x, (p, p, p, p), value=float("-inf"))
    u = xp.unfold(2, k, s).unfold(3, k, s)                 # (B, C, Ho, Wo, k, k)
    return u.reshape(B, C, u.shape[2] * u.shape[3], k * k).permute(0, 1, 3, 2)


def stem_allowance(ref, tol=1e-6):
    """ResNet: the fused stem (conv -> BN -> ReLU -> max-pool, csrc/awr_stem.hip) materialises neither its ReLU nor its pooling
    decisions, so they stay float64's in the yardstick; what they may cost is bounded by the share of the gradient norm that sits on
    stem ReLU inputs within `tol` of zero and on pooling windows with a near tie (root sum of squares)."""
    r = [k[2] for k in kink_table(ref, tol) if k[0] == "pre.1"]
    p = [v for t, v in ref["pool_near_ties"].items() if not t.endswith(".resout")]
    return float(sum(v * v for v in r + p) ** 0.5)


def kink_table(ref, tol=1e-6):
    """[(tag, n elements with |pre-activation| < tol, share of the gradient norm behind that ReLU they carry)] of a float64 trace."""
    out = []
    for tag, (x, g) in ref["relus"].items():
        near = x.abs() < tol
        if g is not None and bool(near.any()):
            out.append((tag, int(near.sum()), float((g * near).norm() / (g.norm() + 1e-300))))
    return out


def decisions_from_trace(ref64, other):
    """(flips, pools) that make the float64 trace differentiate the branch `other` (a float32 trace of the same inputs) took."""
    flips = {}
    for tag, (x64, _) in ref64["relus"].items():
        x = other["relus"][tag][0]
        flips[tag] = (x > 0) != (x64 > 0)
    pools = {tag: other["pools"][tag][1] for tag in ref64["pools"] if not torch.equal(other["pools"][tag][1], ref64["pools"][tag][1])}
    return flips, pools


def _pool_gap(x64, idx_a, idx_b):
    """largest |x[a] - x[b]| over the windows whose argmax differs, relative to the tensor's largest magnitude"""
    m = idx_a != idx_b
    if not bool(m.any()):
        return 0, 0.0
    xa = torch.gather(x64.flatten(2), 2, idx_a.flatten(2)).view(idx_a.shape)
    xb = torch.gather(x64.flatten(2), 2, idx_b.flatten(2)).view(idx_b.shape)
    return int(m.sum()), float((xa - xb)[m].abs().max() / x64.abs().max())


def decisions_from_plan(ref64, plan_tensors):
    """flips that make the float64 trace differentiate the branch the HIP step took.  plan_tensors = engine.Plan.tensors(lazy=True):
    "<bn>.act" holds a materialised [relu](bn(.) [+ res]) (its sign is the decision), "<bn>.act(lazy)" the raw conv output plus the
    per-channel (scale, shift) the loaders apply.  ReLUs the plan never materialises in any form (the fused ResNet stem) keep the
    float64 decisions.  Max-pools: the plan's argmax is re-derived from ITS pooled tensor ("<block>.conv3.conv.out" /
    "<block>.conv3+skip_layer.out", first maximum wins like awr_maxpool_fwd and ATen).  Returns (flips, pools, report) with report =
    [(tag, n decisions that differ from float64's, max |float64 pre-activation| among them relative to 1 + the tensor's largest magnitude
    -- for pools: the largest gap between the two window elements relative to the tensor's magnitude)]."""
    flips, pools, report = {}, {}, []
    for tag, (x64, idx64) in ref64["pools"].items():
        if not tag.endswith(".resout"):
            continue
        base = tag[:-len(".resout")]
        ent = next((plan_tensors[c] for c in (base + ".conv3+skip_layer.out", base + ".conv3.conv.out") if c in plan_tensors), None)
        if ent is None or len(ent) != 2:
            continue
        B, C, H, W = x64.shape
        k = H // idx64.shape[2]
        _, idx = O.TF.max_pool2d(ent[0].permute(0, 3, 1, 2)[:, :C].contiguous(), k, k, return_indices=True)
        idx = idx.cpu()
        n, gap = _pool_gap(x64, idx, idx64)
        if n:
            pools[tag] = idx
            report.append((tag + " (max-pool)", n, gap))
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
            # the kernels' masks are ONE fused multiply-add (`y * scale + shift > 0`, contracted): its sign is the sign of the exact
            # value, which float64 holds (24 + 24 product bits); torch.addcmul rounds the product first and differs on elements
            # within an ulp of the kink (seen once the Winograd mode perturbed the forward by 2e-7)
            on = (val.double() * sc.double() + sh.double()) > 0
        on = on.permute(0, 3, 1, 2).cpu()
        m = on != (x64 > 0)
        flips[tag] = m
        if bool(m.any()):      # relative to the tensor's scale: residual sums grow with depth, and so does the absolute rounding error
            report.append((tag, int(m.sum()), float(x64[m].abs().max() / (1.0 + x64.abs().max()))))
    return flips, pools, report


def rel_l2(a, b, floor=0.0):
    return float((a.double().reshape(-1) - b.double().reshape(-1)).norm() / (b.double().norm() + floor + 1e-300))
