"""Backbones of the AWR hot path as nn.Module-compatible objects backed by libawr_hip.so.

`AwrBackbone` owns ONE flat fp32 parameter arena, one flat gradient arena and one flat buffer
arena in HBM; every entry of the reference's `state_dict()` (same keys, order, shapes, dtypes --
SURVEY.md 8b) is a view into them, so `load_state_dict(torch.load(...)['model'])`, `parameters()`,
`torch.optim.Adam(net.parameters())`, `.cuda()`, `.train()/.eval()` behave like the reference's
modules while a gradient all-reduce / fused Adam can treat the whole network as one buffer.
`forward()` replays a static `engine.Plan` of hand-written HIP kernels; there is no PyTorch
implementation of the network anywhere in this package.
"""
import math

import torch
import torch.nn as nn

from . import _lib as L
from .engine import BNLayer, ConvLayer, HeadLayer, Plan
from .ops import ConvSpec, round_up

PARAM_KINDS = ("conv_w", "deconv_w", "conv_b", "bn_w", "bn_b")
_LAZY_ALL = __import__("os").environ.get("AWR_LAZY_ALL") == "1"       # study hook: never materialise a BN+ReLU output that a single GEMM consumes
_LAZY_MAXC = int(__import__("os").environ.get("AWR_LAZY_MAXC", "128"))  # study hook: widest conv1 output that stays un-materialised


# ---- checkpoint layout ------------------------------------------------------------------------------
def _bn_keys(prefix, c):
    return [(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"), (prefix + ".running_mean", (c,), "bn_mean"),
            (prefix + ".running_var", (c,), "bn_var"), (prefix + ".num_batches_tracked", (), "counter")]


def resnet18_layout(J, downsample=2):
    """Key/shape list of get_deconv_net(18, J, downsample).state_dict() (resnet_deconv.py:31-53)."""
    keys = [("pre.0.weight", (64, 1, 5, 5), "conv_w")] + _bn_keys("pre.1", 64)
    cin = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = "layer%d.%d" % (li, bi)
            keys.append((p + ".conv1.weight", (planes, cin, 3, 3), "conv_w"))
            keys += _bn_keys(p + ".bn1", planes)
            keys.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv_w"))
            keys += _bn_keys(p + ".bn2", planes)
            if bi == 0 and cin != planes:
                keys.append((p + ".downsample.0.weight", (planes, cin, 1, 1), "conv_w"))
                keys += _bn_keys(p + ".downsample.1", planes)
            cin = planes
    for i in range(4 - int(math.log2(downsample))):
        keys.append(("deconv_layers.%d.weight" % (3 * i), (cin, 256, 4, 4), "deconv_w"))
        keys += _bn_keys("deconv_layers.%d" % (3 * i + 1), 256)
        cin = 256
    for name, n in (("final1", 3 * J), ("final2", J)):
        keys += [(name + ".weight", (n, 256, 1, 1), "conv_w"), (name + ".bias", (n,), "conv_b")]
    return keys


def _hgconv_keys(prefix, cin, cout, k, bn=False):
    keys = [(prefix + ".conv.weight", (cout, cin, k, k), "conv_w"), (prefix + ".conv.bias", (cout,), "conv_b")]
    return keys + (_bn_keys(prefix + ".bn", cout) if bn else [])


def _residual_keys(prefix, cin, cout):
    h = cout // 2
    keys = []
    for bn, conv, a, b, k in (("bn1", "conv1", cin, h, 1), ("bn2", "conv2", h, h, 3), ("bn3", "conv3", h, cout, 1)):
        keys += _bn_keys("%s.%s" % (prefix, bn), a) + _hgconv_keys("%s.%s" % (prefix, conv), a, b, k)
    return keys + _hgconv_keys(prefix + ".skip_layer", cin, cout, 1)     # present even when unused (hourglass.py:38)


def _hourglass_keys(prefix, depth, f):
    keys = _residual_keys(prefix + ".up1", f, f) + _residual_keys(prefix + ".low1", f, f)
    keys += _hourglass_keys(prefix + ".low2", depth - 1, f) if depth > 1 else _residual_keys(prefix + ".low2", f, f)
    return keys + _residual_keys(prefix + ".low3", f, f)


def hourglass_layout(nstack, J, f=256):
    """Key/shape list of PoseNet('hourglass_<nstack>', J).state_dict() (hourglass.py:105-142)."""
    keys = _hgconv_keys("pre.0", 1, 64, 5, bn=True) + _residual_keys("pre.1", 64, 128)
    keys += _residual_keys("pre.3", 128, 256) + _residual_keys("pre.4", 256, f)
    for i in range(nstack):
        keys += _hourglass_keys("hgs.%d.0" % i, 4, f)
    for i in range(nstack):
        keys += _residual_keys("features.%d.0" % i, f, f) + _hgconv_keys("features.%d.1" % i, f, f, 1, bn=True)
    for name, n in (("outs_1", 3 * J), ("outs_2", J)):
        for i in range(nstack):
            keys += [("%s.%d.weight" % (name, i), (n, f, 1, 1), "conv_w"), ("%s.%d.bias" % (name, i), (n,), "conv_b")]
    for name, cin in (("merge_features", f), ("merge_preds", 4 * J)):
        for i in range(nstack - 1):
            keys += _hgconv_keys("%s.%d.conv" % (name, i), cin, f, 1)
    return keys


class _Node(nn.Module):
    """Anonymous container: the module tree exists only to reproduce the reference's state_dict keys."""


class AwrBackbone(nn.Module):
    nstage = 1

    def __init__(self, layout, J, unused_prefixes=()):
        super().__init__()
        self.J = J
        self._layout = layout
        # parameters that never receive a gradient go to the tail of the arena so the optimiser /
        # all-reduce can skip them exactly like torch skips `p.grad is None`
        def unused(key):
            return any(key.startswith(u) for u in unused_prefixes)
        plist = [(k, s) for k, s, kind in layout if kind in PARAM_KINDS]
        order = [e for e in plist if not unused(e[0])] + [e for e in plist if unused(e[0])]
        self._poff, off = {}, 0
        for k, s in order:
            n = int(torch.Size(s).numel())
            self._poff[k] = (off, n, s)
            off = round_up(off + n, 4)                        # keep every view 16-byte aligned
            if not unused(k):
                self.n_active = off
        self.n_params = off
        self._boff, boff = {}, 0
        for k, s, kind in layout:
            if kind in ("bn_mean", "bn_var"):
                self._boff[k] = (boff, s[0])
                boff += round_up(s[0], 4)
        nbn = sum(1 for _, _, kind in layout if kind == "counter")
        self._arena = torch.zeros(self.n_params)
        self._garena = torch.zeros(self.n_params)
        self._barena = torch.zeros(boff)
        self._counters = torch.zeros(nbn, dtype=torch.int64)      # num_batches_tracked, kept on the host
        self._unused = set(k for k, _ in plist if unused(k))
        self._plans = {}
        self._packed_sig = {}
        self._stats_version = 0          # bumped by every training forward (kernels update running stats behind torch's back)
        ci = 0
        for key, shape, kind in layout:
            node, leaf = self._node_for(key)
            if kind in PARAM_KINDS:
                node.register_parameter(leaf, nn.Parameter(self._view(self._arena, key)))
            elif kind == "counter":
                node.register_buffer(leaf, self._counters[ci])
                ci += 1
            else:
                o, n = self._boff[key]
                node.register_buffer(leaf, self._barena[o:o + n])
        self.reset_parameters()

    # ---- arena plumbing ------------------------------------------------------------------------------
    def _node_for(self, key):
        parts = key.split(".")
        node = self
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        return node, parts[-1]

    def _view(self, arena, key):
        o, n, s = self._poff[key]
        return arena[o:o + n].view(s)

    def _rebind(self):
        ci = 0
        for key, shape, kind in self._layout:
            node, leaf = self._node_for(key)
            if kind in PARAM_KINDS:
                node._parameters[leaf].data = self._view(self._arena, key)
                node._parameters[leaf].grad = None
            elif kind == "counter":
                node._buffers[leaf] = self._counters[ci]
                ci += 1
            else:
                o, n = self._boff[key]
                node._buffers[leaf] = self._barena[o:o + n]
        self._plans.clear()
        self._packed_sig.clear()

    def _apply(self, fn, recurse=True):
        """.cuda()/.to(device)/.cpu(): move the arenas as whole buffers and rebind every view."""
        probe = fn(torch.zeros(1))
        if probe.dtype != torch.float32:
            raise L.AwrError("the AWR HIP path is fp32 (1e-3 mm parity); dtype conversion is not supported")
        self._arena, self._garena, self._barena = fn(self._arena), fn(self._garena), fn(self._barena)
        self._rebind()
        return self

    @property
    def device(self):
        return self._arena.device

    def grad_view(self, key):
        return self._view(self._garena, key)

    def param_view(self, key):
        return self._view(self._arena, key)

    def flat_params(self):
        return self._arena

    def flat_grads(self):
        return self._garena

    def reset_parameters(self, seed=None):
        """Reference initialisation (resnet_deconv.py:93-115 / torch defaults for hourglass)."""
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        last_w = None
        with torch.no_grad():
            for key, shape, kind in self._layout:
                if kind == "bn_w":
                    self.param_view(key).fill_(1.0)
                elif kind == "bn_b":
                    self.param_view(key).zero_()
                elif kind == "bn_var":
                    o, n = self._boff[key]
                    self._barena[o:o + n].fill_(1.0)
                elif kind == "bn_mean":
                    o, n = self._boff[key]
                    self._barena[o:o + n].zero_()
                elif kind in ("conv_w", "deconv_w"):
                    v = self._init_conv(key, shape, kind, g)
                    self.param_view(key).copy_(v)
                    last_w = shape
                elif kind == "conv_b":
                    self.param_view(key).copy_(self._init_bias(key, shape, last_w, g))
            self._counters.zero_()

    # ---- execution ----------------------------------------------------------------------------------
    def _layers(self):
        """name -> ConvLayer/BNLayer/HeadLayer bound to the current arenas (cached per device)."""
        if getattr(self, "_layer_cache_dev", None) != self._arena.data_ptr():
            self._layer_cache = self._make_layers()
            self._layer_cache_dev = self._arena.data_ptr()
        return self._layer_cache

    def _conv_layer(self, key_w, spec, key_b=None):
        return ConvLayer(spec, self.param_view(key_w), self.grad_view(key_w), self.param_view(key_b) if key_b else None,
                         self.grad_view(key_b) if key_b else None, name=key_w.rsplit(".", 1)[0])

    def _bn_layer(self, prefix, idx):
        c = self._poff[prefix + ".weight"][1]
        om, _ = self._boff[prefix + ".running_mean"]
        ov, _ = self._boff[prefix + ".running_var"]
        return BNLayer(c, self.param_view(prefix + ".weight"), self.param_view(prefix + ".bias"), self.grad_view(prefix + ".weight"),
                       self.grad_view(prefix + ".bias"), self._barena[om:om + c], self._barena[ov:ov + c], self._counters[idx], name=prefix)

    def get_plan(self, B, H, training, supervised="all", bn_repeat=1, n_buckets=1):
        if not self._arena.is_cuda:
            raise L.AwrError("the AWR backbone runs on the MI355X only: call .cuda() first (there is no CPU path)")
        key = (B, H, bool(training), supervised if isinstance(supervised, str) else tuple(supervised), bn_repeat, n_buckets)
        plan = self._plans.get(key)
        if plan is None:
            plan = Plan(B, self.device, training, bn_repeat=bn_repeat)
            plan.img = plan.alloc(B, 1, H, H)
            plan.gen = 0
            plan.garena, plan.n_active, plan.n_buckets = self._garena, self.n_active, n_buckets
            self.build(plan, plan.img, H)
            if training:
                plan.build_backward(range(self.nstage) if supervised == "all" else supervised)
            self._plans[key] = plan
        return plan

    def _sig(self):
        # Version counters of the Parameter / buffer OBJECTS, not of the arenas: `_rebind()` re-points `param.data` at arena
        # views, which keeps each parameter's own counter, so an in-place update through the parameters (stock
        # torch.optim step, nn.init, load_state_dict) never bumps `self._arena._version`.
        pv = sum(p._version for p in self.parameters())
        bv = sum(b._version for b in self.buffers())
        return (pv, bv, self._arena._version, self._barena._version, self._stats_version, self._arena.data_ptr())

    def weights_changed(self):
        """Tell the module that kernels modified parameters / running stats through raw pointers."""
        self._stats_version += 1

    def sync_weights(self, plan, force=False):
        sig = self._sig()
        if force or self._packed_sig.get(id(plan)) != sig:
            plan.refresh_weights()
            self._packed_sig[id(plan)] = self._sig()

    def mark_packed(self, plan):
        self._packed_sig[id(plan)] = self._sig()

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != x.shape[3]:
            raise L.AwrError("expected a (B,1,H,H) depth batch, got %s" % (tuple(x.shape),))
        if not x.is_cuda:
            raise L.AwrError("input must live on the GPU (no CPU fallback)")
        plan = self.get_plan(x.shape[0], x.shape[2], self.training)
        params = [p for p in self.parameters()]
        outs = _BackboneFn.apply(self, plan, x, *params)
        return self._wrap_outputs(list(outs))

    def _wrap_outputs(self, outs):
        return outs[0]


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, plan, x, *params):
        # training forwards always re-pack (one batched launch): the weights normally changed since the last step, and a
        # missed update (e.g. through `p.data`) would silently train on stale packed copies
        net.sync_weights(plan, force=plan.training)
        plan.img.copy_(x.detach().float())
        plan.forward()
        if plan.training:
            net.weights_changed()            # running stats moved (weights did not):
            net.mark_packed(plan)            # ... stale for inference plans, still fresh for this one
        plan.gen += 1
        ctx.net, ctx.plan, ctx.gen = net, plan, plan.gen
        return tuple(o.clone() for o in plan.outputs)

    @staticmethod
    def backward(ctx, *gouts):
        net, plan = ctx.net, ctx.plan
        if not plan.training:
            raise L.AwrError("backward through an eval-mode forward: call net.train() (the inference plan keeps no activations)")
        if plan.gen != ctx.gen:
            raise L.AwrError("backward after a newer forward of the same (batch, size): saved activations were overwritten")
        for buf, g in zip(plan.grad_outs, gouts):
            if g is None:
                buf.zero_()
            else:
                buf.copy_(g)
        plan.backward()
        grads = []
        for (key, shape, kind) in ((k, s, kd) for k, s, kd in net._layout if kd in PARAM_KINDS):
            grads.append(None if key in net._unused else net.grad_view(key).clone())
        return (None, None, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------------
class ResNet18Deconv(AwrBackbone):
    """get_deconv_net(18, J, downsample): resnet_deconv.py:8-16, :19-136, BasicBlock :145-174."""

    def __init__(self, J, downsample=2):
        self.downsample = downsample
        self.ndeconv = 4 - int(math.log2(downsample))
        super().__init__(resnet18_layout(J, downsample), J)

    def _init_conv(self, key, shape, kind, g):
        if kind == "deconv_w" or key.startswith("final"):
            return torch.randn(shape, generator=g) * 0.001                           # :103-104, :108-115
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[2] * shape[3] * shape[0]))   # :95-97

    def _init_bias(self, key, shape, wshape, g):
        return torch.zeros(shape)                                                    # :110, :114

    def _make_layers(self):
        Lr, bn_idx = {}, [0]

        def bn(prefix):
            Lr[prefix] = self._bn_layer(prefix, bn_idx[0])
            bn_idx[0] += 1
        Lr["pre.0"] = self._conv_layer("pre.0.weight", ConvSpec("conv", 25, 64, 1, 1, 0, cin_pad=32))
        bn("pre.1")
        cin = 64
        for li, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
            for bi in range(2):
                p = "layer%d.%d" % (li, bi)
                s = stride if bi == 0 else 1
                Lr[p + ".conv1"] = self._conv_layer(p + ".conv1.weight", ConvSpec("conv", cin, planes, 3, s, 1))
                bn(p + ".bn1")
                Lr[p + ".conv2"] = self._conv_layer(p + ".conv2.weight", ConvSpec("conv", planes, planes, 3, 1, 1))
                bn(p + ".bn2")
                if bi == 0 and cin != planes:
                    Lr[p + ".downsample.0"] = self._conv_layer(p + ".downsample.0.weight", ConvSpec("conv", cin, planes, 1, s, 0))
                    bn(p + ".downsample.1")
                cin = planes
        for i in range(self.ndeconv):
            Lr["deconv_layers.%d" % (3 * i)] = self._conv_layer("deconv_layers.%d.weight" % (3 * i), ConvSpec("deconv", cin, 256, 4, 2, 1))
            bn("deconv_layers.%d" % (3 * i + 1))
            cin = 256
        Lr["head"] = HeadLayer(256, self.J, self.param_view("final1.weight"), self.grad_view("final1.weight"), self.param_view("final1.bias"),
                               self.grad_view("final1.bias"), self.param_view("final2.weight"), self.grad_view("final2.weight"),
                               self.param_view("final2.bias"), self.grad_view("final2.bias"), name="final")
        return Lr

    @staticmethod
    def _cbr(P, Lr, x, conv, bn, relu, res=None, lazy=False):
        """conv -> BatchNorm [-> +res] [-> ReLU]; fused into one GEMM epilogue in inference.  lazy (training): the
        BN+ReLU output is not materialised -- legal when every consumer is a conv / max-pool loader."""
        if P.training:
            y = P.conv(x, Lr[conv], want_stats=True, use_bias=Lr[conv].bias is not None)
            return P.bn_act(y, Lr[bn], relu, res, lazy=lazy and res is None)
        return P.conv(x, Lr[conv], out_affine=P.fold_bn(Lr[bn]), res=res, relu_out=relu, use_bias=Lr[conv].bias is not None)

    def build(self, P, img, H):
        Lr = self._layers()
        c = P.stem_pool(img, Lr["pre.0"], Lr["pre.1"], H, H)      # conv 5x5 -> BN -> ReLU -> MaxPool(3,2,1), one fused kernel family
        for li in range(1, 5):
            for bi in range(2):
                p = "layer%d.%d" % (li, bi)
                # the downsample branch is emitted first so that, in the reversed (backward) order, conv1's
                # full-coverage data gradient initialises d(block input) before the strided 1x1 accumulates
                if (p + ".downsample.0") in Lr:      # 1x1 stride-2 projection + its BatchNorm: independent of conv1 / conv2 until the residual add
                    P.fork()
                    r = self._cbr(P, Lr, c, p + ".downsample.0", p + ".downsample.1", False)
                    P.end_fork(r)
                else:
                    r = c
                # bn1 + ReLU feeds conv2 only.  Un-materialised (applied by conv2's loaders) while that is cheaper than one write + read
                # of the tensor: the loader arithmetic costs 8-15 % of a GEMM whose K grows with the channel count (measured per layer,
                # profiles/r02_summary.md), the tensor pass does not -- beyond 128 channels the activation is written out
                o = self._cbr(P, Lr, c, p + ".conv1", p + ".bn1", True, lazy=_LAZY_ALL or Lr[p + ".conv1"].spec.cout <= _LAZY_MAXC)
                c = self._cbr(P, Lr, o, p + ".conv2", p + ".bn2", True, res=r)
        for i in range(self.ndeconv):
            # feeds the next transposed conv (K = 4 x 256 per phase: materialised, as above) or the 1x1 head GEMM (K = 256: lazy)
            c = self._cbr(P, Lr, c, "deconv_layers.%d" % (3 * i), "deconv_layers.%d" % (3 * i + 1), True, lazy=_LAZY_ALL or (i == self.ndeconv - 1))
        pred = P.conv(c, Lr["head"])
        P.head_out(pred, self.J)


class HourglassNet(AwrBackbone):
    """PoseNet('hourglass_<n>', J): hourglass.py:105-165 (Conv :6-25, Residual :28-59, Hourglass :62-88)."""

    def __init__(self, nstack, J, f=256):
        self.nstack, self.f = nstack, f
        self.nstage = nstack
        layout = hourglass_layout(nstack, J, f)
        # Residual.skip_layer exists in every block but only runs when inp_dim != out_dim (hourglass.py:38-47)
        unused = []
        for k, s, kind in layout:
            if kind == "conv_w" and ".skip_layer." in k and s[0] == s[1]:
                unused.append(k.rsplit(".conv.", 1)[0] + ".")
        super().__init__(layout, J, unused_prefixes=tuple(unused))

    def _init_conv(self, key, shape, kind, g):
        bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])          # kaiming_uniform_(a=sqrt(5))
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def _init_bias(self, key, shape, wshape, g):
        bound = 1.0 / math.sqrt(wshape[1] * wshape[2] * wshape[3])
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def _make_layers(self):
        Lr, idx = {}, 0
        cp = round_up(4 * self.J, 32)
        for key, shape, kind in self._layout:
            if kind == "conv_w" and key.endswith(".conv.weight"):
                pfx = key[:-len(".conv.weight")]
                if (pfx + ".") in [u for u in self._unused_prefix_list()]:
                    continue
                cout, cin, k, _ = shape
                if pfx == "pre.0":
                    spec = ConvSpec("conv", 25, 64, 1, 1, 0, cin_pad=32)
                elif pfx.startswith("merge_preds"):
                    spec = ConvSpec("conv", cin, cout, 1, 1, 0, cin_pad=cp)
                else:
                    spec = ConvSpec("conv", cin, cout, k, 1, (k - 1) // 2)
                Lr[pfx] = self._conv_layer(key, spec, pfx + ".conv.bias")
            elif kind == "bn_w":
                pfx = key[:-len(".weight")]
                Lr[pfx] = self._bn_layer(pfx, idx)
                idx += 1
        for i in range(self.nstack):
            a, b = "outs_1.%d" % i, "outs_2.%d" % i
            Lr["head.%d" % i] = HeadLayer(self.f, self.J, self.param_view(a + ".weight"), self.grad_view(a + ".weight"),
                                          self.param_view(a + ".bias"), self.grad_view(a + ".bias"), self.param_view(b + ".weight"),
                                          self.grad_view(b + ".weight"), self.param_view(b + ".bias"), self.grad_view(b + ".bias"),
                                          name="outs.%d" % i)
        return Lr

    def _unused_prefix_list(self):
        return set(k.rsplit(".conv.", 1)[0] + "." for k in self._unused if k.endswith(".conv.weight"))

    def _residual(self, P, Lr, x, p):
        skip = Lr.get(p + ".skip_layer")
        if P.training:
            # the three pre-activations feed exactly one conv each: never written to HBM
            a = P.bn_act(x, Lr[p + ".bn1"], True, lazy=True)
            a = P.bn_act(P.conv(a, Lr[p + ".conv1"], want_stats=True), Lr[p + ".bn2"], True, lazy=True)
            a = P.bn_act(P.conv(a, Lr[p + ".conv2"], want_stats=True), Lr[p + ".bn3"], True, lazy=True)
            r = P.conv(x, skip) if skip is not None else x
            return P.conv(a, Lr[p + ".conv3"], res=r, want_stats=True)
        # inference: bn1 stays a loader affine (x also feeds the skip path un-normalised); bn2 / bn3 + ReLU normalise tensors that
        # only conv2 / conv3 read, so they fold into the EPILOGUE of the conv that produces them -- applied once per element
        # instead of once per (element, tap, column tile) in the 3x3 conv's loader, which cost 8-15 % of that GEMM
        y = P.conv(x, Lr[p + ".conv1"], in_affine=P.fold_bn(Lr[p + ".bn1"]), relu_in=True, out_affine=P.fold_bn(Lr[p + ".bn2"]), relu_out=True)
        y = P.conv(y, Lr[p + ".conv2"], out_affine=P.fold_bn(Lr[p + ".bn3"]), relu_out=True)
        r = P.conv(x, skip) if skip is not None else x
        return P.conv(y, Lr[p + ".conv3"], res=r)

    def _hg(self, P, Lr, x, p, depth):
        # the skip branch of a level only meets the low-resolution path again at the up-sampling add: issued on its own side
        # stream, its full-resolution GEMMs fill the chip while the main stream walks the small (<= 16x16) levels
        P.fork(depth)
        up1 = self._residual(P, Lr, x, p + ".up1")
        P.end_fork(up1)
        low = self._residual(P, Lr, P.maxpool(x, 2, 2, 0), p + ".low1")
        low = self._hg(P, Lr, low, p + ".low2", depth - 1) if depth > 1 else self._residual(P, Lr, low, p + ".low2")
        low = self._residual(P, Lr, low, p + ".low3")
        return P.upsample_add(up1, low)

    def build(self, P, img, H):
        Lr = self._layers()
        c = ResNet18Deconv._cbr(P, Lr, P.im2col5(img, H, H), "pre.0", "pre.0.bn", True)
        c = self._residual(P, Lr, c, "pre.1")
        c = P.maxpool(c, 2, 2, 0)
        c = self._residual(P, Lr, c, "pre.3")
        c = self._residual(P, Lr, c, "pre.4")
        for i in range(self.nstack):
            hg = self._hg(P, Lr, c, "hgs.%d.0" % i, 4)
            ft = self._residual(P, Lr, hg, "features.%d.0" % i)
            ft = ResNet18Deconv._cbr(P, Lr, ft, "features.%d.1" % i, "features.%d.1.bn" % i, True, lazy=True)      # head / merge GEMMs
            pred = P.conv(ft, Lr["head.%d" % i])
            P.head_out(pred, self.J)
            if i < self.nstack - 1:
                m = P.conv(pred, Lr["merge_preds.%d.conv" % i], res=c)
                c = P.conv(ft, Lr["merge_features.%d.conv" % i], res=m, want_stats=True)

    def _wrap_outputs(self, outs):
        return outs                                     # list over stacks, like hourglass.py:165
