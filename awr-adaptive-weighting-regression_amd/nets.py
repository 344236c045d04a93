"""Backbones of the AWR hot path as nn.Module-compatible objects backed by libawr_hip.so.

`AwrBackbone` owns ONE flat fp32 parameter arena, one flat gradient arena and one flat buffer
arena in HBM; every entry of the reference's `state_dict()` (same keys, order, shapes, dtypes --
SURVEY.md 8b) is a view into them, so `load_state_dict(torch.load(...)['model'])`, `parameters()`,
`torch.optim.Adam(net.parameters())`, `.cuda()`, `.train()/.eval()` behave like the reference's
modules while a gradient all-reduce / fused Adam can treat the whole network as one buffer.
The checkpoint layout, the execution plans and their replay live in the library (csrc/awr_net.hip,
include/awr_hip.h "Network-level API"); this module is the nn.Module shell around those handles:
there is no PyTorch implementation of the network anywhere in this package.
"""
import math

import torch
import torch.nn as nn

from . import _lib as L
from .engine import PARAM_KINDS, NetHandle, Plan
from . import _winograd_code


def _auto_rule():
    k, d = L.C.c_int(0), L.C.c_int(0)
    L.call("awr_get_gemm_accum_auto", L.C.byref(k), L.C.byref(d))
    return (k.value, d.value)


class _Node(nn.Module):
    """Anonymous container: the module tree exists only to reproduce the reference's state_dict keys."""


class AwrBackbone(nn.Module):
    nstage = 1
    downsample = 2

    def __init__(self, kind, nstack, J, downsample=2):
        super().__init__()
        self.J = J
        self._handle = NetHandle(kind, nstack, J, downsample)        # checkpoint layout (keys, shapes, arena offsets) from the library
        self._layout = [(k, s, kd) for k, s, kd, _, _ in self._handle.layout]
        self.nstage = self._handle.nstage
        self.n_params, self.n_active = self._handle.n_params, self._handle.n_active
        # parameters that never receive a gradient sit at the tail of the arena so the optimiser / all-reduce can skip them
        # exactly like torch skips `p.grad is None`
        self._poff = {k: (off, int(torch.Size(s).numel()), s) for k, s, kd, off, _ in self._handle.layout if kd in PARAM_KINDS}
        self._boff = {k: (off, s[0]) for k, s, kd, off, _ in self._handle.layout if kd in ("bn_mean", "bn_var")}
        self._unused = set(k for k, s, kd, off, un in self._handle.layout if un)
        self._arena = torch.zeros(self.n_params)
        self._garena = torch.zeros(self.n_params)
        self._barena = torch.zeros(self._handle.n_buffers)
        self._counters = torch.zeros(self._handle.n_counters, dtype=torch.int64)      # num_batches_tracked, kept on the host
        self._plans = {}
        self._packed_sig = {}
        self._stats_version = 0          # bumped by every training forward (kernels update running stats behind torch's back)
        ci = 0
        for key, shape, kind in self._layout:
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
        for plan in self._plans.values():      # awr_net_bind destroys the net's native plans
            plan.h = None
        self._plans.clear()
        self._packed_sig.clear()
        if self._arena.is_cuda:
            self._handle.bind(self._arena, self._garena, self._barena)

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
    def get_plan(self, B, H, training, supervised="all", bn_repeat=1, n_buckets=1, accum=None, winograd=None):
        """accum: None = the process-wide mode (awr_amd.set_gemm_accum), "ordered" / "blocked" / "auto" = this plan's own (awr_conv_args.accum;
        "auto" blocks the forward launches whose K extent reaches awr_get_gemm_accum_auto()'s threshold, include/awr_hip.h)."""
        if not self._arena.is_cuda:
            raise L.AwrError("the AWR backbone runs on the MI355X only: call .cuda() first (there is no CPU path)")
        acc = int(L.lib.awr_get_gemm_accum()) if accum is None else {"ordered": 0, "blocked": 1, "auto": 2}[accum]
        if acc == 2 and not training:
            acc = 0          # auto never blocks an evaluation plan's launches (include/awr_hip.h): the same plan as "ordered"
        if acc and (int(L.lib.awr_get_gemm_products()) != 1 or int(L.lib.awr_get_gemm_staging()) == 0):
            # the blocked kernel exists for the FP32-MFMA mode with LDS-DMA staging only (include/awr_hip.h: awr_conv_args.accum): fail here, at
            # build time and by name, rather than at the first launch -- "auto" never asks for what cannot run, so it degrades to ordered
            if acc == 1:
                raise L.AwrError("blocked accumulation (the parity mode) needs gemm_products = 1 and LDS-DMA staging; this process runs "
                                 "products = %d, staging = %d" % (L.lib.awr_get_gemm_products(), L.lib.awr_get_gemm_staging()))
        wino = int(L.lib.awr_get_conv_winograd()) if winograd is None else _winograd_code(winograd)      # captured when the plan is built, like accum
        key = (B, H, bool(training), supervised if isinstance(supervised, str) else tuple(supervised), bn_repeat, n_buckets, L.lib.awr_get_deterministic(), acc,
               _auto_rule() if acc == 2 else 0, wino)
        plan = self._plans.get(key)
        if plan is None:
            was, was_w = int(L.lib.awr_get_gemm_accum()), int(L.lib.awr_get_conv_winograd())
            L.call("awr_set_gemm_accum", acc)          # plans capture the modes when they are built
            L.call("awr_set_conv_winograd", wino)
            try:
                plan = Plan(self, B, H, H // getattr(self, "downsample", 2), self.J, training, supervised, bn_repeat, n_buckets)
            finally:
                L.call("awr_set_gemm_accum", was)
                L.call("awr_set_conv_winograd", was_w)
            nw, wm = L.C.c_int(0), L.C.c_double(0)
            L.call("awr_plan_winograd", plan.h, L.C.byref(nw), L.C.byref(wm))
            plan.n_winograd, plan.winograd_macs = nw.value, wm.value       # launches that run as Winograd F(2x2, 3x3) (forward, data / weight gradients); their algorithmic MACs
            plan.accum = acc          # 0 = ordered, 1 = blocked, 2 = auto (blocked per launch by K extent): what the plan's GEMM launches captured
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
        if plan.nhwc:                        # an engine shared this plan and left it on the NHWC boundary: the module returns NCHW maps
            plan.set_nhwc_boundary(False)
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
        if plan.nhwc:
            plan.set_nhwc_boundary(False)
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
class ResNetDeconv(AwrBackbone):
    """get_deconv_net(depth, J, downsample): resnet_deconv.py:8-16, :19-136; BasicBlock (:145-174) for depth 18, Bottleneck (:177-215)
    for 50 / 101 / 152."""

    def __init__(self, J, downsample=2, depth=18):
        self.downsample, self.depth = downsample, depth
        self.ndeconv = 4 - int(math.log2(downsample))
        super().__init__(0, depth, J, downsample)           # (kind 0 takes the depth in the `nstack` slot of awr_net_create)

    def _init_conv(self, key, shape, kind, g):
        if kind == "deconv_w" or key.startswith("final"):
            return torch.randn(shape, generator=g) * 0.001                           # :103-104, :108-115
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[2] * shape[3] * shape[0]))   # :95-97

    def _init_bias(self, key, shape, wshape, g):
        return torch.zeros(shape)                                                    # :110, :114


class ResNet18Deconv(ResNetDeconv):
    def __init__(self, J, downsample=2):
        super().__init__(J, downsample, 18)


class HourglassNet(AwrBackbone):
    """PoseNet('hourglass_<n>', J): hourglass.py:105-165 (Conv :6-25, Residual :28-59, Hourglass :62-88)."""
    downsample = 2                  # dense maps at half the input resolution (hourglass.py:113-118)

    def __init__(self, nstack, J, f=256):
        assert f == 256, "the reference's hourglass is 256 features wide (hourglass.py:106)"
        self.nstack, self.f = nstack, f
        super().__init__(1, nstack, J, 2)

    def _init_conv(self, key, shape, kind, g):
        bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])          # kaiming_uniform_(a=sqrt(5))
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def _init_bias(self, key, shape, wshape, g):
        bound = 1.0 / math.sqrt(wshape[1] * wshape[2] * wshape[3])
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def _wrap_outputs(self, outs):
        return outs                                     # list over stacks, like hourglass.py:165
