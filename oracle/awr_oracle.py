"""CPU oracle for the AWR hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional (state-dict driven) restatement in
torch-CPU fp32 of the algorithm the reference implements in

    model/resnet_deconv.py   (ResNet18-deconv backbone)
    model/hourglass.py       (stacked hourglass backbone)
    util/feature_tool.py     (AWR head + GT dense-map synthesis)
    model/loss.py            (Huber, delta = 0.01)
    train.py:107-131         (one optimisation step)
    util/eval_tool.py, util/util.py (mm metric)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package never imports it: the HIP path fails
loudly when its extension is missing.

Pinning: ``tools/gen_golden.py`` imports the real reference from
``/root/reference`` (dev container only), asserts every function below equals
the reference on seeded inputs, and writes the golden vectors committed under
``tests/golden/``.  The reference itself ships no tests or known-answer vectors
for this path, so those golden vectors are the pin.

All arithmetic is fp32.  The conv/BN numerics live in PyTorch (ATen/oneDNN),
which the reference pins only as ``torch==1.1.0`` / "Pytorch 1.4.0"
(requirements.txt:1, README.md:11); here it is torch 2.10 CPU.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as TF

BN_EPS = 1e-5          # torch BatchNorm2d default (resnet_deconv.py:33, hourglass.py:16)
BN_MOMENTUM = 0.1      # resnet_deconv.py:6 ; hourglass uses the torch default 0.1
HUBER_DELTA = 0.01     # loss.py:12-13
SOFTMAX_BETA = 30.0    # feature_tool.py:60
DEPTH_BG = 0.99        # feature_tool.py:35, :57

# Test-only switch: evaluate the same formulas without the reference's `.float()` casts so that float64
# inputs stay float64 end to end.  Used to measure how ill-conditioned a gradient is (fp32-vs-fp64 gap
# of the oracle itself) before judging the HIP path's deviation from the fp32 oracle.
HIGH_PRECISION = False


def _fl(t):
    return t if HIGH_PRECISION else t.float()


# --------------------------------------------------------------------------
# state-dict manifests (checkpoint layout)
# --------------------------------------------------------------------------
def _bn_entries(prefix, c):
    return [
        (prefix + ".weight", (c,), "bn_w"),
        (prefix + ".bias", (c,), "bn_b"),
        (prefix + ".running_mean", (c,), "bn_mean"),
        (prefix + ".running_var", (c,), "bn_var"),
        (prefix + ".num_batches_tracked", (), "counter"),
    ]


RESNET_BLOCKS = {18: [2, 2, 2, 2], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}      # resnet_deconv.py:9-13


def resnet_manifest(depth=18, J=14, downsample=2):
    """Ordered (key, shape, kind) list of get_deconv_net(depth, J, downsample); resnet_deconv.py:8-16, :19-53 -- BasicBlock (:145-174)
    for depth 18, Bottleneck with expansion 4 (:177-215) for 50 / 101 / 152; a block registers conv1, bn1, conv2, bn2, (conv3, bn3,)
    downsample in that order."""
    bott = depth != 18
    exp = 4 if bott else 1
    m = [("pre.0.weight", (64, 1, 5, 5), "conv_w")] + _bn_entries("pre.1", 64)
    cin = 64
    for li, (planes, stride) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)], start=1):
        for bi in range(RESNET_BLOCKS[depth][li - 1]):
            p = "layer%d.%d" % (li, bi)
            s = stride if bi == 0 else 1
            if not bott:
                m.append((p + ".conv1.weight", (planes, cin, 3, 3), "conv_w"))
                m += _bn_entries(p + ".bn1", planes)
                m.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv_w"))
                m += _bn_entries(p + ".bn2", planes)
            else:
                m.append((p + ".conv1.weight", (planes, cin, 1, 1), "conv_w"))
                m += _bn_entries(p + ".bn1", planes)
                m.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv_w"))
                m += _bn_entries(p + ".bn2", planes)
                m.append((p + ".conv3.weight", (planes * exp, planes, 1, 1), "conv_w"))
                m += _bn_entries(p + ".bn3", planes * exp)
            if bi == 0 and (s != 1 or cin != planes * exp):
                m.append((p + ".downsample.0.weight", (planes * exp, cin, 1, 1), "conv_w"))
                m += _bn_entries(p + ".downsample.1", planes * exp)
            cin = planes * exp
    ndeconv = 4 - int(math.log2(downsample))
    for i in range(ndeconv):
        m.append(("deconv_layers.%d.weight" % (3 * i), (cin, 256, 4, 4), "deconv_w"))
        m += _bn_entries("deconv_layers.%d" % (3 * i + 1), 256)
        cin = 256
    m += [("final1.weight", (3 * J, 256, 1, 1), "conv_w"), ("final1.bias", (3 * J,), "conv_b"),
          ("final2.weight", (J, 256, 1, 1), "conv_w"), ("final2.bias", (J,), "conv_b")]
    return m


def resnet18_manifest(J=14, downsample=2):
    return resnet_manifest(18, J, downsample)


def _hg_conv(prefix, cin, cout, k, bn):
    e = [(prefix + ".conv.weight", (cout, cin, k, k), "conv_w"), (prefix + ".conv.bias", (cout,), "conv_b")]
    if bn:
        e += _bn_entries(prefix + ".bn", cout)
    return e


def _hg_residual(prefix, cin, cout):
    """hourglass.py:28-42 -- registration order bn1, conv1, bn2, conv2, bn3, conv3, skip_layer."""
    h = cout // 2
    e = _bn_entries(prefix + ".bn1", cin) + _hg_conv(prefix + ".conv1", cin, h, 1, False)
    e += _bn_entries(prefix + ".bn2", h) + _hg_conv(prefix + ".conv2", h, h, 3, False)
    e += _bn_entries(prefix + ".bn3", h) + _hg_conv(prefix + ".conv3", h, cout, 1, False)
    e += _hg_conv(prefix + ".skip_layer", cin, cout, 1, False)
    return e


def _hg_hourglass(prefix, n, f):
    """hourglass.py:62-78 -- up1, low1, low2 (recursive), low3."""
    e = _hg_residual(prefix + ".up1", f, f) + _hg_residual(prefix + ".low1", f, f)
    if n > 1:
        e += _hg_hourglass(prefix + ".low2", n - 1, f)
    else:
        e += _hg_residual(prefix + ".low2", f, f)
    e += _hg_residual(prefix + ".low3", f, f)
    return e


def hourglass_manifest(nstack=1, J=14, f=256):
    """Ordered (key, shape, kind) list of PoseNet('hourglass_<nstack>', J); hourglass.py:105-142."""
    m = _hg_conv("pre.0", 1, 64, 5, True) + _hg_residual("pre.1", 64, 128)
    m += _hg_residual("pre.3", 128, 256) + _hg_residual("pre.4", 256, f)
    for i in range(nstack):
        m += _hg_hourglass("hgs.%d.0" % i, 4, f)
    for i in range(nstack):
        m += _hg_residual("features.%d.0" % i, f, f) + _hg_conv("features.%d.1" % i, f, f, 1, True)
    for i in range(nstack):
        m += [("outs_1.%d.weight" % i, (3 * J, f, 1, 1), "conv_w"), ("outs_1.%d.bias" % i, (3 * J,), "conv_b")]
    for i in range(nstack):
        m += [("outs_2.%d.weight" % i, (J, f, 1, 1), "conv_w"), ("outs_2.%d.bias" % i, (J,), "conv_b")]
    for i in range(nstack - 1):
        m += _hg_conv("merge_features.%d.conv" % i, f, f, 1, False)
    for i in range(nstack - 1):
        m += _hg_conv("merge_preds.%d.conv" % i, 4 * J, f, 1, False)
    return m


def manifest_for(net, J):
    """net: 'resnet_<18|50|101|152>' | 'hourglass_<n>' (train.py:51-57)."""
    if net.startswith("resnet"):
        return resnet_manifest(int(net.split("_")[-1]), J, 2)
    return hourglass_manifest(int(net.split("_")[-1]), J)


# --------------------------------------------------------------------------
# procedural weights: bit-identical on every box from integer hashing, so the
# golden fixtures do not have to ship 61 MB of weights.
# --------------------------------------------------------------------------
def _hash_uniform(n, stream, seed):
    """n fp32 values in [-0.5, 0.5) from an exact 64-bit integer hash (numpy uint64 wraparound)."""
    i = np.arange(n, dtype=np.uint64)
    x = i * np.uint64(0x9E3779B97F4A7C15) + np.uint64((stream * 0x632BE59BD9B4E019 + seed * 0xD1B54A32D192ED03 + 0x1234567) & 0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 24) - 0.5).astype(np.float32)


PARAM_KINDS = ("conv_w", "deconv_w", "conv_b", "bn_w", "bn_b")


def procedural_state(manifest, seed=0):
    """Fill a manifest deterministically.  Conv weights ~ U(-.5,.5)*sqrt(12)*sqrt(2/fan_in) (unit-gain
    He scale, keeps activations O(1) through 20+ layers); BN gamma / running_var in [0.75,1.25],
    conv bias / BN beta / running_mean in [-0.1,0.1]."""
    sd = OrderedDict()
    with np.errstate(over="ignore"):
        for k, (key, shape, kind) in enumerate(manifest):
            if kind == "counter":
                sd[key] = torch.zeros((), dtype=torch.int64)
                continue
            u = _hash_uniform(int(np.prod(shape)), k, seed)
            if kind == "deconv_w":               # ConvTranspose2d weight (Cin,Cout,kh,kw); 4 of 16 taps hit a pixel
                v = u * np.float32(math.sqrt(12.0) * math.sqrt(2.0 / (shape[0] * 4)))
            elif kind == "conv_w":
                gain = 0.1 if key.startswith(("final", "outs_")) else (0.5 if ".conv3." in key else 1.0)
                v = u * np.float32(math.sqrt(12.0) * math.sqrt(gain / (shape[1] * shape[2] * shape[3])))
            elif kind in ("bn_w", "bn_var"):
                v = np.float32(1.0) + u * np.float32(0.5)
            else:                                # conv_b / bn_b / bn_mean
                v = u * np.float32(0.2)
            sd[key] = torch.from_numpy(v.reshape(shape).copy())
    return sd


def reference_init_state(net, J, seed=0):
    """Random init with the reference's distributions (resnet_deconv.py:93-115; torch defaults for
    hourglass, hourglass.py:10).  Values differ from the reference's RNG stream; distributions match."""
    g = torch.Generator().manual_seed(seed)
    resnet = net.startswith("resnet")
    sd = OrderedDict()
    last_w = None
    for key, shape, kind in manifest_for(net, J):
        if kind == "counter":
            sd[key] = torch.zeros((), dtype=torch.int64)
        elif kind in ("bn_w", "bn_var"):
            sd[key] = torch.ones(shape)
        elif kind in ("bn_b", "bn_mean"):
            sd[key] = torch.zeros(shape)
        elif kind in ("conv_w", "deconv_w"):
            if resnet:
                if kind == "deconv_w" or key.startswith("final"):
                    std = 0.001                                              # :103-104, :108-115
                else:
                    std = math.sqrt(2.0 / (shape[2] * shape[3] * shape[0]))  # :95-97
                sd[key] = torch.randn(shape, generator=g) * std
            else:                                                            # kaiming_uniform(a=sqrt(5))
                bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
                sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
            last_w = sd[key]
        else:                                                                # conv_b
            if resnet:
                sd[key] = torch.zeros(shape)                                 # :110, :114
            else:
                bound = 1.0 / math.sqrt(last_w.shape[1] * last_w.shape[2] * last_w.shape[3])
                sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return sd


# --------------------------------------------------------------------------
# backbone forward passes (functional)
# --------------------------------------------------------------------------
def _bn(sd, p, x, training):
    """BatchNorm2d forward; training updates running stats in-place in `sd`."""
    return TF.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                         training, BN_MOMENTUM, BN_EPS) if not training else _bn_train(sd, p, x)


def _bn_train(sd, p, x):
    y = TF.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                      True, BN_MOMENTUM, BN_EPS)
    if p + ".num_batches_tracked" in sd:
        sd[p + ".num_batches_tracked"] += 1
    return y


def resnet_forward(sd, x, training=False, downsample=2):
    """(B,1,H,H) -> (B,4J,H/ds,H/ds); resnet_deconv.py:118-136 with BasicBlock :158-174 / Bottleneck :194-215 (told apart by the
    presence of a third conv in the block)."""
    c = TF.conv2d(x, sd["pre.0.weight"], None, 1, 2)
    c = TF.relu(_bn(sd, "pre.1", c, training))
    c = TF.max_pool2d(c, 3, 2, 1)
    for li, stride in enumerate([1, 2, 2, 2], start=1):
        bi = 0
        while "layer%d.%d.conv1.weight" % (li, bi) in sd:
            p = "layer%d.%d" % (li, bi)
            s = stride if bi == 0 else 1
            if p + ".conv3.weight" in sd:          # Bottleneck: 1x1 -> 3x3 (stride) -> 1x1
                o = TF.conv2d(c, sd[p + ".conv1.weight"], None, 1, 0)
                o = TF.relu(_bn(sd, p + ".bn1", o, training))
                o = TF.conv2d(o, sd[p + ".conv2.weight"], None, s, 1)
                o = TF.relu(_bn(sd, p + ".bn2", o, training))
                o = TF.conv2d(o, sd[p + ".conv3.weight"], None, 1, 0)
                o = _bn(sd, p + ".bn3", o, training)
            else:
                o = TF.conv2d(c, sd[p + ".conv1.weight"], None, s, 1)
                o = TF.relu(_bn(sd, p + ".bn1", o, training))
                o = TF.conv2d(o, sd[p + ".conv2.weight"], None, 1, 1)
                o = _bn(sd, p + ".bn2", o, training)
            if p + ".downsample.0.weight" in sd:
                r = TF.conv2d(c, sd[p + ".downsample.0.weight"], None, s, 0)
                r = _bn(sd, p + ".downsample.1", r, training)
            else:
                r = c
            c = TF.relu(o + r)
            bi += 1
    for i in range(4 - int(math.log2(downsample))):
        c = TF.conv_transpose2d(c, sd["deconv_layers.%d.weight" % (3 * i)], None, 2, 1)
        c = TF.relu(_bn(sd, "deconv_layers.%d" % (3 * i + 1), c, training))
    vec = TF.conv2d(c, sd["final1.weight"], sd["final1.bias"])
    ht = TF.conv2d(c, sd["final2.weight"], sd["final2.bias"])
    return torch.cat([vec, ht], 1)


resnet18_forward = resnet_forward


def _hg_res(sd, p, x, training):
    """Pre-activation bottleneck; hourglass.py:44-59."""
    cin, cout = sd[p + ".conv1.conv.weight"].shape[1], sd[p + ".conv3.conv.weight"].shape[0]
    r = TF.conv2d(x, sd[p + ".skip_layer.conv.weight"], sd[p + ".skip_layer.conv.bias"]) if cin != cout else x
    o = TF.relu(_bn(sd, p + ".bn1", x, training))
    o = TF.conv2d(o, sd[p + ".conv1.conv.weight"], sd[p + ".conv1.conv.bias"])
    o = TF.relu(_bn(sd, p + ".bn2", o, training))
    o = TF.conv2d(o, sd[p + ".conv2.conv.weight"], sd[p + ".conv2.conv.bias"], 1, 1)
    o = TF.relu(_bn(sd, p + ".bn3", o, training))
    o = TF.conv2d(o, sd[p + ".conv3.conv.weight"], sd[p + ".conv3.conv.bias"])
    return o + r


def _hg_module(sd, p, n, x, training):
    """hourglass.py:80-88."""
    up1 = _hg_res(sd, p + ".up1", x, training)
    low = TF.max_pool2d(x, 2, 2)
    low = _hg_res(sd, p + ".low1", low, training)
    low = _hg_module(sd, p + ".low2", n - 1, low, training) if n > 1 else _hg_res(sd, p + ".low2", low, training)
    low = _hg_res(sd, p + ".low3", low, training)
    return up1 + TF.interpolate(low, scale_factor=2, mode="nearest")


def hourglass_forward(sd, x, nstack, training=False):
    """(B,1,H,H) -> list over stacks of (B,4J,H/2,H/2); hourglass.py:144-165."""
    c = TF.conv2d(x, sd["pre.0.conv.weight"], sd["pre.0.conv.bias"], 1, 2)
    c = TF.relu(_bn(sd, "pre.0.bn", c, training))
    c = _hg_res(sd, "pre.1", c, training)
    c = TF.max_pool2d(c, 2, 2)
    c = _hg_res(sd, "pre.3", c, training)
    c = _hg_res(sd, "pre.4", c, training)
    outs = []
    for i in range(nstack):
        hg = _hg_module(sd, "hgs.%d.0" % i, 4, c, training)
        ft = _hg_res(sd, "features.%d.0" % i, hg, training)
        ft = TF.conv2d(ft, sd["features.%d.1.conv.weight" % i], sd["features.%d.1.conv.bias" % i])
        ft = TF.relu(_bn(sd, "features.%d.1.bn" % i, ft, training))
        pred = torch.cat([TF.conv2d(ft, sd["outs_1.%d.weight" % i], sd["outs_1.%d.bias" % i]),
                          TF.conv2d(ft, sd["outs_2.%d.weight" % i], sd["outs_2.%d.bias" % i])], 1)
        outs.append(pred)
        if i < nstack - 1:
            c = c + TF.conv2d(pred, sd["merge_preds.%d.conv.conv.weight" % i], sd["merge_preds.%d.conv.conv.bias" % i]) \
                  + TF.conv2d(ft, sd["merge_features.%d.conv.conv.weight" % i], sd["merge_features.%d.conv.conv.bias" % i])
    return outs


def backbone_forward(net, sd, x, training=False):
    """Returns a list of per-stage dense maps (length 1 for resnet)."""
    if net.startswith("resnet"):
        return [resnet_forward(sd, x, training)]
    return hourglass_forward(sd, x, int(net.split("_")[-1]), training)


# --------------------------------------------------------------------------
# AWR head, GT dense map, loss
# --------------------------------------------------------------------------
def _down_depth(img, F):
    """F.interpolate(img, size=[F,F]) (nearest) picks source pixel floor(i*H/F); feature_tool.py:20,:44."""
    H = img.shape[-1]
    assert H % F == 0
    r = H // F
    return img[:, :, ::r, ::r]


def _grid(F, device=None):
    """Pixel-centre grid in [-1,1]; feature_tool.py:23-24, :50-51."""
    a = 2.0 * (torch.arange(F, device=device).to(torch.float64 if HIGH_PRECISION else torch.float32) + 0.5) / F - 1.0
    return a.view(1, F).expand(F, F), a.view(F, 1).expand(F, F)       # (x varies along W, y along H)


def joint2offset(jt_uvd, img, ks, F):
    """GT dense map (B,4J,F,F) from joints (B,J,3) and depth (B,1,H,H); feature_tool.py:12-39."""
    B, J, _ = jt_uvd.shape
    d = _down_depth(img, F)[:, 0]                                      # (B,F,F)
    gx, gy = _grid(F, img.device)
    coord = torch.stack([gx.expand(B, F, F), gy.expand(B, F, F), d], 1)     # (B,3,F,F)
    off = jt_uvd.view(B, J, 3, 1, 1) - coord.view(B, 1, 3, F, F)       # :29
    dist = torch.sqrt((off * off).sum(2) + 1e-8)                       # :31
    unit = off / dist.unsqueeze(2)                                     # :33
    hm = (ks - dist) / ks                                              # :34
    mask = (hm >= 0).to(hm.dtype) * (d < DEPTH_BG).to(hm.dtype).unsqueeze(1)     # :35
    return _fl(torch.cat([(unit * mask.unsqueeze(2)).reshape(B, 3 * J, F, F), hm * mask], 1))


def joint2offset_ieee(jt_uvd, img, ks, F):
    """joint2offset restated with numpy float32 scalars-and-arrays only: every operation (subtract, multiply, add, sqrt, divide) is a
    single correctly rounded IEEE-754 binary32 operation in the order of util/feature_tool.py:29-39.  torch's CPU sqrt is NOT
    correctly rounded (MKL VML: ~0.6 % of arguments come out 1 ulp off on this host, more on others), so `joint2offset` above --
    bit-identical to the reference on the SAME host -- is not a host-independent answer; this one is, and it is what the HIP
    kernel (correctly rounded v_sqrt / v_div sequences, no FMA contraction) reproduces bit for bit."""
    f32 = np.float32
    jt = jt_uvd.numpy().astype(np.float32)
    B, J, _ = jt.shape
    r = img.shape[-1] // F
    d = img.numpy()[:, 0, ::r, ::r].astype(np.float32)
    a = (f32(2.0) * (np.arange(F, dtype=np.float32) + f32(0.5)) / f32(F) - f32(1.0)).astype(np.float32)
    o0 = jt[:, :, 0, None, None] - a[None, None, None, :]
    o1 = jt[:, :, 1, None, None] - a[None, None, :, None]
    o2 = jt[:, :, 2, None, None] - d[:, None]
    o0, o1 = np.broadcast_to(o0, o2.shape), np.broadcast_to(o1, o2.shape)
    dist = np.sqrt(((o0 * o0 + o1 * o1) + o2 * o2) + f32(1e-8))
    hm = (f32(ks) - dist) / f32(ks)
    mk = (hm >= 0).astype(np.float32) * (d[:, None] < f32(DEPTH_BG)).astype(np.float32)
    unit = np.stack([o0 / dist * mk, o1 / dist * mk, o2 / dist * mk], 2).reshape(B, 3 * J, F, F)
    return np.concatenate([unit, hm * mk], 1).astype(np.float32)


def offset2joint_softmax(offset, img, ks):
    """Dense map (B,4J,F,F) + depth -> joints (B,J,3); feature_tool.py:41-65."""
    B, C4, F, _ = offset.shape
    J = C4 // 4
    d = _down_depth(img, F)[:, 0].reshape(B, 1, F * F)                 # (B,1,P)
    gx, gy = _grid(F, offset.device)
    coord = torch.stack([gx.reshape(1, -1).expand(B, -1), gy.reshape(1, -1).expand(B, -1), d[:, 0]], 1)  # (B,3,P)
    m = (d < DEPTH_BG).to(offset.dtype)                                # :57
    vec = offset[:, :3 * J].reshape(B, J, 3, F * F) * m.unsqueeze(1)   # :58
    h = offset[:, 3 * J:].reshape(B, J, F * F) * m                     # :59
    w = torch.softmax(h * SOFTMAX_BETA, -1)                            # :60  (masked pixels keep logit 0)
    dis = ks - h * ks                                                  # :61
    val = vec * dis.unsqueeze(2) + coord.unsqueeze(1)                  # :63
    return _fl((val * w.unsqueeze(2)).sum(-1))


def head_backward(offset, img, ks, g_jt):
    """Closed-form gradient of offset2joint_softmax w.r.t. `offset`; used to
    cross-check autograd and as documentation of what the HIP backward kernel computes."""
    B, C4, F, _ = offset.shape
    J = C4 // 4
    P = F * F
    d = _down_depth(img, F)[:, 0].reshape(B, 1, P)
    gx, gy = _grid(F, offset.device)
    coord = torch.stack([gx.reshape(1, -1).expand(B, -1), gy.reshape(1, -1).expand(B, -1), d[:, 0]], 1).unsqueeze(1)
    m = (d < DEPTH_BG).to(offset.dtype)
    vec = offset[:, :3 * J].reshape(B, J, 3, P)
    h = offset[:, 3 * J:].reshape(B, J, P) * m
    w = torch.softmax(h * SOFTMAX_BETA, -1)
    dis = ks - h * ks
    val = vec * m.unsqueeze(1) * dis.unsqueeze(2) + coord
    out = (val * w.unsqueeze(2)).sum(-1)                                # (B,J,3)
    g = g_jt.view(B, J, 3, 1)
    g_vec = g * (w * dis * m).unsqueeze(2)
    g_h = (g * (-ks * w.unsqueeze(2) * vec * m.unsqueeze(1)
                + SOFTMAX_BETA * w.unsqueeze(2) * (val - out.unsqueeze(-1)))).sum(2) * m
    return torch.cat([g_vec.reshape(B, 3 * J, F, F), g_h.reshape(B, J, F, F)], 1)


def huber(x, y, delta=HUBER_DELTA):
    """My_SmoothL1Loss.forward; loss.py:8-25.  Mean over ALL elements of
    0.5 z^2 (|z|<delta) / delta(|z|-delta/2) otherwise."""
    assert x.shape == y.shape
    z = _fl(x - y)
    a = z.abs()
    small = (a < delta).to(z.dtype)
    return (0.5 * z * z * small).mean() + (delta * (a - 0.5 * delta) * (1.0 - small)).mean()


# --------------------------------------------------------------------------
# one optimisation step (train.py:107-131) and Adam (torch.optim.Adam defaults)
# --------------------------------------------------------------------------
def params_of(sd, manifest):
    return [k for k, _, kind in manifest if kind in PARAM_KINDS]


def loss_and_grads(net, sd, img, jt_gt, ks, coord_w, dense_w, J=None):
    """Training-mode forward + backward.  Returns (loss, loss_coord, loss_dense, {key: grad|None},
    jt_pred).  Hourglass quirk (train.py:116-121): the net is run once per stage and only the LAST
    stage's loss survives; BN running stats therefore update `stacks` times."""
    J = J or jt_gt.shape[1]
    man = manifest_for(net, J)
    pkeys = params_of(sd, man)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in pkeys}
    work = OrderedDict((k, leaves[k] if k in leaves else sd[k]) for k in sd)
    F = img.shape[-1] // 2
    gt = joint2offset(jt_gt, img, ks, F)
    nstage = 1 if net.startswith("resnet") else int(net.split("_")[-1])
    for stage in range(nstage):
        pred = backbone_forward(net, work, img, True)[stage]
        jt = offset2joint_softmax(pred, img, ks)
        l_coord = coord_w * huber(jt, jt_gt)
        l_dense = dense_w * huber(pred, gt)
        loss = l_coord + l_dense
    grads = torch.autograd.grad(loss, [leaves[k] for k in pkeys], allow_unused=True)
    for k in sd:                                     # propagate BN buffer updates
        if k not in leaves:
            sd[k] = work[k]
    return (loss.detach(), torch.as_tensor(l_coord).detach(), torch.as_tensor(l_dense).detach(),
            dict(zip(pkeys, grads)), jt.detach())


def adam_update(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.Adam single-tensor rule (train.py:66-67 uses the defaults); in-place."""
    if wd != 0.0:
        g = g + wd * p
    m.lerp_(g, 1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def train_step(net, sd, opt_state, img, jt_gt, ks, coord_w, dense_w, lr=1e-3, wd=0.0):
    """One full reference step: GT map, forward, head, loss, backward, Adam.  `opt_state` is
    {'step': int, 'm': {key: tensor}, 'v': {key: tensor}}; params without grad are skipped exactly
    like torch.optim.Adam skips `p.grad is None`."""
    loss, lc, ld, grads, jt = loss_and_grads(net, sd, img, jt_gt, ks, coord_w, dense_w)
    opt_state["step"] += 1
    for k, g in grads.items():
        if g is None:
            continue
        if k not in opt_state["m"]:
            opt_state["m"][k] = torch.zeros_like(sd[k])
            opt_state["v"][k] = torch.zeros_like(sd[k])
        adam_update(sd[k], g, opt_state["m"][k], opt_state["v"][k], opt_state["step"], lr=lr, wd=wd)
    return loss, lc, ld, grads, jt


# --------------------------------------------------------------------------
# evaluator: normalised uvd -> mm error; eval_tool.py:20-56, util.py:13-20
# --------------------------------------------------------------------------
NYU_PARAS = (588.03, 587.07, 320.0, 240.0)   # nyu_loader.py:23
NYU_FLIP = -1                                # nyu_loader.py:33


def uvd2xyz(pts, paras=NYU_PARAS, flip=NYU_FLIP):
    """util.py:13-20."""
    p = np.array(pts, dtype=np.float32).reshape(-1, 3).copy()
    p[:, :2] = (p[:, :2] - np.asarray(paras[2:], np.float32)) * p[:, 2:] / np.asarray(paras[:2], np.float32)
    p[:, 1] *= flip
    return p.reshape(np.shape(pts)).astype(np.float32)


def joint_errors_mm(jt_uvd_pred, jt_xyz_gt, center_xyz, M, cube, img_size=128, paras=NYU_PARAS, flip=NYU_FLIP):
    """Batched EvalUtil.feed (eval_tool.py:20-46): returns (B,J) Euclidean errors in mm and the
    predictions in original-image uvd (what test.py:105-108 writes to results/*.txt)."""
    jt = np.array(jt_uvd_pred, dtype=np.float32).copy()
    B = jt.shape[0]
    errs, uvds = [], []
    for i in range(B):
        Mi = np.asarray(M[i], np.float32)
        Minv = np.linalg.inv(Mi)
        p = jt[i]
        p[:, :2] = (p[:, :2] + 1) * img_size / 2.0
        p[:, 2] = p[:, 2] * np.float32(cube[i][2]) / 2.0 + np.float32(center_xyz[i][2])
        hom = np.hstack([p[:, :2], np.ones((p.shape[0], 1))])
        p[:, :2] = np.dot(Minv, hom.T).T[:, :2]
        uvds.append(p.copy())
        xyz = uvd2xyz(p, paras, flip)
        gt = np.asarray(jt_xyz_gt[i], np.float32) * (np.asarray(cube[i], np.float32) / 2.0) + np.asarray(center_xyz[i], np.float32)
        errs.append(np.sqrt(np.sum(np.square(gt - xyz), axis=1)))
    return np.stack(errs), np.stack(uvds)


_trapz = getattr(np, "trapezoid", None) or np.trapz


def measures(errs_mm):
    """EvalUtil.get_measures (eval_tool.py:80-122) on a (N,J) error matrix: MPE, median, AUC, PCK."""
    th = np.linspace(0, 50, 100)
    norm = _trapz(np.ones_like(th), th)
    e = np.asarray(errs_mm, dtype=np.float64)
    mean = np.mean([np.mean(e[:, j]) for j in range(e.shape[1])])
    med = np.mean([np.median(e[:, j]) for j in range(e.shape[1])])
    pck = np.stack([[np.mean((e[:, j] <= t).astype("float")) for t in th] for j in range(e.shape[1])])
    auc = np.mean([_trapz(pck[j], th) / norm for j in range(e.shape[1])])
    return mean, med, auc, pck.mean(0), th


# --------------------------------------------------------------------------
# synthetic inputs shared by tests, bench and the golden generator
# --------------------------------------------------------------------------
def synth_batch(B, H=128, J=14, seed=1234):
    """Depth crops: background exactly 1.0, a ~30%-area disk of hand-like depth; joints U(-0.6,0.6)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(H).float(), indexing="ij")
    R = 0.31 * H
    c = H / 2 + (torch.rand(B, 2, generator=g) * 2 - 1) * (H / 16)
    r2 = (xx.view(1, H, H) - c[:, 0].view(B, 1, 1)) ** 2 + (yy.view(1, H, H) - c[:, 1].view(B, 1, 1)) ** 2
    fg = 0.5 * r2 / (R * R) - 0.3 + 0.02 * torch.randn(B, H, H, generator=g)
    img = torch.where(r2 < R * R, fg.clamp(-1.0, 0.98), torch.ones(()))
    jt = (torch.rand(B, J, 3, generator=g) * 2 - 1) * 0.6
    return img.view(B, 1, H, H).float().contiguous(), jt.float().contiguous()
