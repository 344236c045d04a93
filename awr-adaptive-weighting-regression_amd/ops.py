"""Host-side geometry + thin tensor wrappers over the backbone entry points of libawr_hip.so.

Activations are NHWC fp32 CUDA(HIP) tensors.  A `ConvSpec` describes one reference layer
(nn.Conv2d or nn.ConvTranspose2d); from it this module derives the three GEMM problems the layer
needs -- forward, data gradient, weight gradient -- as `awr_conv_args` / `awr_wgrad_args`
geometries plus the weight-packing recipe for each (see include/awr_hip.h).
"""
import ctypes as C

import torch

from . import _lib as L

N_ALIGN = 128     # packed weight rows are padded to a multiple of the widest GEMM tile
K_ALIGN = 32      # GEMM K-slice


def round_up(x, m):
    return (x + m - 1) // m * m


class ConvSpec:
    """kind: 'conv' (nn.Conv2d, weight (Cout,Cin,k,k)) or 'deconv' (nn.ConvTranspose2d k4 s2 p1,
    weight (Cin,Cout,k,k)).  cin_pad/cout_pad: channel counts of the NHWC tensors (>= logical)."""

    def __init__(self, kind, cin, cout, k, stride, pad, cin_pad=None, cout_pad=None):
        assert kind in ("conv", "deconv")
        self.kind, self.cin, self.cout, self.k, self.stride, self.pad = kind, cin, cout, k, stride, pad
        self.cin_pad = cin_pad or cin
        self.cout_pad = cout_pad or cout
        self.T = k * k
        assert self.cin_pad % K_ALIGN == 0, "GEMM K extent (Cin=%d) must be a multiple of %d" % (self.cin_pad, K_ALIGN)
        if kind == "deconv":
            assert stride == 2

    # ---- shapes -----------------------------------------------------------------------------
    def out_hw(self, h, w):
        if self.kind == "conv":
            return (h + 2 * self.pad - self.k) // self.stride + 1, (w + 2 * self.pad - self.k) // self.stride + 1
        return (h - 1) * self.stride - 2 * self.pad + self.k, (w - 1) * self.stride - 2 * self.pad + self.k

    # ---- tap tables ---------------------------------------------------------------------------
    def _gather_taps(self):
        """conv-type gather: one phase, taps (ky-p, kx-p, ky*k+kx)."""
        return [(0, 0, [(ky - self.pad, kx - self.pad, ky * self.k + kx) for ky in range(self.k) for kx in range(self.k)])]

    def _mirror_taps(self):
        """stride-1 data gradient: dx[y] = sum_k dy[y + p - k] w[k]."""
        return [(0, 0, [(self.pad - ky, self.pad - kx, ky * self.k + kx) for ky in range(self.k) for kx in range(self.k)])]

    def _scatter_phases(self):
        """transposed-type (output stride 2): phase (py,px) takes the taps with (py+p-ky) even."""
        s, p, k = self.stride, self.pad, self.k
        phases = []
        for py in range(s):
            for px in range(s):
                taps = [((py + p - ky) // s, (px + p - kx) // s, ky * k + kx)
                        for ky in range(k) if (py + p - ky) % s == 0
                        for kx in range(k) if (px + p - kx) % s == 0]
                if taps:
                    phases.append((py, px, taps))
        return phases

    # ---- the three GEMM problems ------------------------------------------------------------------
    def fwd_problem(self, hin, win):
        """-> dict(Hin,Win,Cin,Hout,Wout,N,Hq,Wq,so,si,phases,full) for the forward pass."""
        hout, wout = self.out_hw(hin, win)
        if self.kind == "conv":
            return dict(Hin=hin, Win=win, Cin=self.cin_pad, Hout=hout, Wout=wout, N=self.cout_pad, Hq=hout, Wq=wout, so=1,
                        si=self.stride, phases=self._gather_taps(), full=True)
        ph = self._scatter_phases()
        return dict(Hin=hin, Win=win, Cin=self.cin_pad, Hout=hout, Wout=wout, N=self.cout_pad, Hq=hout // 2, Wq=wout // 2, so=2, si=1,
                    phases=ph, full=len(ph) == 4)

    def dgrad_problem(self, hin, win):
        """gradient w.r.t. the layer input: reads dY (B,Hout,Wout,cout_pad), writes dX (B,hin,win,cin_pad)."""
        hout, wout = self.out_hw(hin, win)
        base = dict(Hin=hout, Win=wout, Cin=self.cout_pad, Hout=hin, Wout=win, N=self.cin_pad)
        if self.kind == "deconv":      # adjoint of a transposed conv is a strided conv
            return dict(base, Hq=hin, Wq=win, so=1, si=self.stride, phases=self._gather_taps(), full=True)
        if self.stride == 1:
            return dict(base, Hq=hin, Wq=win, so=1, si=1, phases=self._mirror_taps(), full=True)
        assert hin % 2 == 0 and win % 2 == 0
        ph = self._scatter_phases()
        return dict(base, Hq=hin // 2, Wq=win // 2, so=2, si=1, phases=ph, full=len(ph) == 4)

    # weight packing recipes: (d0, d1, T, transpose, rows, ld) for awr_pack_weight
    def fwd_pack(self):
        if self.kind == "conv":
            return (self.cout, self.cin, self.T, 0, round_up(self.cout_pad, N_ALIGN), self.cin_pad)
        return (self.cin, self.cout, self.T, 1, round_up(self.cout_pad, N_ALIGN), self.cin_pad)

    def dgrad_pack(self):
        if self.kind == "conv":
            return (self.cout, self.cin, self.T, 1, round_up(self.cin_pad, N_ALIGN), self.cout_pad)
        return (self.cin, self.cout, self.T, 0, round_up(self.cin_pad, N_ALIGN), self.cout_pad)

    def wgrad_problem(self, hin, win):
        """R[d0][t][d1] in the weight's own (d0,d1) order.  conv: D=dY, G=X; deconv: D=X, G=dY."""
        hout, wout = self.out_hw(hin, win)
        taps = [(ky - self.pad, kx - self.pad) for ky in range(self.k) for kx in range(self.k)]
        if self.kind == "conv":
            return dict(D="dy", Hd=hout, Wd=wout, Cd=self.cout_pad, Hg=hin, Wg=win, Cg=self.cin_pad, sg=self.stride, taps=taps,
                        d0=self.cout, d1=self.cin)
        return dict(D="x", Hd=hin, Wd=win, Cd=self.cin_pad, Hg=hout, Wg=wout, Cg=self.cout_pad, sg=self.stride, taps=taps,
                    d0=self.cin, d1=self.cout)


def make_conv_args(prob, B, x, w, out, in_scale=None, in_shift=None, bias=None, out_scale=None, out_shift=None, res=None,
                   stats=None, relu_in=False, relu_out=False, T=None, partial=None, split_k=0, in_split=None):
    """partial: (split_max, *out.shape) scratch -> split-K (awr_hip.h: awr_conv_args.partial); split_k = 0 lets the library pick the depth.
    in_split: the pre-cut image of x (split_act) for the split-operand mode's LDS-DMA kernel."""
    a = L.ConvArgs()
    if in_split is not None:
        a.in_split = L.ptr(in_split)
        a._keep_split = in_split
    if partial is not None:
        a.partial, a.split_max, a.split_k = L.ptr(partial), partial.shape[0], split_k
        a._keep = partial
    a.in_, a.w, a.out = L.ptr(x), L.ptr(w), L.ptr(out)
    a.w_split = L.ptr(getattr(w, "split", None))
    a.in_scale, a.in_shift, a.bias = L.ptr(in_scale), L.ptr(in_shift), L.ptr(bias)
    a.out_scale, a.out_shift, a.res, a.stats = L.ptr(out_scale), L.ptr(out_shift), L.ptr(res), L.ptr(stats)
    a.B, a.Hin, a.Win, a.Cin = B, prob["Hin"], prob["Win"], prob["Cin"]
    a.Hq, a.Wq, a.Hout, a.Wout, a.N = prob["Hq"], prob["Wq"], prob["Hout"], prob["Wout"], prob["N"]
    a.so, a.si, a.T = prob["so"], prob["si"], T
    a.relu_in, a.relu_out = int(relu_in), int(relu_out)
    a.nphase = len(prob["phases"])
    for i, (py, px, taps) in enumerate(prob["phases"]):
        ph = a.ph[i]
        ph.py, ph.px, ph.ntaps = py, px, len(taps)
        for t, (dy, dx, wt) in enumerate(taps):
            ph.tap[t] = (dy & 0xff) | ((dx & 0xff) << 8) | (wt << 16)
    return a


def make_wgrad_args(prob, B, D, G, R, ld, d_affine=None, g_affine=None, d_colsum=None, algo=0):
    """d_affine / g_affine: (scale, shift, relu) applied to the operand while staging (un-materialised BN+ReLU)."""
    a = L.WgradArgs()
    a.D, a.G, a.R = L.ptr(D), L.ptr(G), L.ptr(R)
    a.d_colsum = L.ptr(d_colsum)
    a.algo = algo
    if d_affine is not None:
        a.d_scale, a.d_shift, a.d_relu = L.ptr(d_affine[0]), L.ptr(d_affine[1]), int(d_affine[2])
    if g_affine is not None:
        a.g_scale, a.g_shift, a.g_relu = L.ptr(g_affine[0]), L.ptr(g_affine[1]), int(g_affine[2])
    a.B, a.Hd, a.Wd, a.Cd = B, prob["Hd"], prob["Wd"], prob["Cd"]
    a.Hg, a.Wg, a.Cg, a.sg, a.T, a.ld = prob["Hg"], prob["Wg"], prob["Cg"], prob["sg"], len(prob["taps"]), ld
    for t, (dy, dx) in enumerate(prob["taps"]):
        a.dy[t], a.dx[t] = dy, dx
    return a


# ---------------------------------------------------------------------------------------------
# eager wrappers (allocate outputs; used by tests and by the drop-in modules' slow path)
# ---------------------------------------------------------------------------------------------
def alloc_packed(rows, T, ld, device):
    """Packed GEMM weight [rows][T][ld] fp32 plus, as attribute `.split`, its split image (include/awr_hip.h:
    awr_split_weight) for the 6- / 9-product modes of awr_conv_gemm."""
    p = torch.zeros(rows, T, ld, device=device, dtype=torch.float32)
    p.split = torch.zeros(rows * T * ld * 3, device=device, dtype=torch.int16)
    return p


def split_act(x, scale=None, shift=None, relu=False):
    """Split image of an NHWC activation tensor (include/awr_hip.h: awr_split_act): [relu](x * scale + shift) cut exactly into three bf16
    pieces, 6 bytes per element, the layout awr_conv_args.in_split reads."""
    C_ = x.shape[-1]
    out = torch.empty(x.numel() * 3, device=x.device, dtype=torch.int16)
    L.call("awr_split_act", L.ptr(x), L.ptr(scale), L.ptr(shift), int(bool(relu)), x.numel() // C_, C_, L.ptr(out), L.stream())
    return out


def split_packed(p):
    L.call("awr_split_weight", L.ptr(p), L.ptr(p.split), p.numel(), L.stream())


def pack_weight(w, recipe, out=None, row_offset=0, rows=None):
    d0, d1, T, tr, n_rows, ld = recipe
    if out is None:
        out = alloc_packed(n_rows, T, ld, w.device)
    rows = n_rows if rows is None else rows
    dst = out.view(-1)[row_offset * T * ld:]
    L.call("awr_pack_weight", L.ptr(w), d0, d1, T, tr, rows, ld, dst.data_ptr(), L.stream())
    if hasattr(out, "split"):
        split_packed(out)
    return out


def conv_forward(spec, x, wp, **kw):
    B, H, W, _ = x.shape
    prob = spec.fwd_problem(H, W)
    out = kw.pop("out", None)
    if out is None:
        out = (torch.empty if prob["full"] or kw.get("res") is not None else torch.zeros)(
            B, prob["Hout"], prob["Wout"], prob["N"], device=x.device, dtype=torch.float32)
    a = make_conv_args(prob, B, x, wp, out, T=spec.T, **kw)
    L.call("awr_conv_gemm", C.byref(a), L.stream())
    return out


def conv_dgrad(spec, dy, wp, hin, win, **kw):
    B = dy.shape[0]
    prob = spec.dgrad_problem(hin, win)
    out = kw.pop("out", None)
    if out is None:
        out = (torch.empty if prob["full"] else torch.zeros)(B, hin, win, prob["N"], device=dy.device, dtype=torch.float32)
        if not prob["full"] and kw.get("res") is None:
            kw["res"] = out
    a = make_conv_args(prob, B, dy, wp, out, T=spec.T, **kw)
    L.call("awr_conv_gemm", C.byref(a), L.stream())
    return out


def conv_wgrad(spec, x, dy, grad=None, accumulate=False, x_affine=None, bias_grad=None, algo=0):
    """Returns the gradient in checkpoint layout (same shape as the layer's weight).  x_affine: (scale, shift, relu)
    when the layer input is relu(x*scale+shift) of the tensor passed as `x`."""
    B, H, W, _ = x.shape
    prob = spec.wgrad_problem(H, W)
    ld = prob["Cg"]
    R = torch.zeros(prob["Cd"], len(prob["taps"]), ld, device=x.device, dtype=torch.float32)
    D, G = (dy, x) if prob["D"] == "dy" else (x, dy)
    bslots = None
    if bias_grad is not None:
        assert prob["D"] == "dy"
        bslots = torch.zeros(16, prob["Cd"], device=x.device, dtype=torch.float32)      # AWR_STAT_SLOTS copies, summed below
    a = make_wgrad_args(prob, B, D, G, R, ld, d_colsum=bslots, algo=algo, **({"g_affine" if prob["D"] == "dy" else "d_affine": x_affine} if x_affine else {}))
    L.call("awr_conv_wgrad", C.byref(a), L.stream())
    if grad is None:
        grad = torch.empty(prob["d0"], prob["d1"], spec.k, spec.k, device=x.device, dtype=torch.float32)
    L.call("awr_unpack_wgrad", L.ptr(R), prob["d0"], prob["d1"], spec.T, ld, L.ptr(grad), int(accumulate), L.stream())
    if bias_grad is not None:
        bias_grad.copy_(bslots.sum(0)[:bias_grad.numel()])
    return grad


def nhwc(x):
    """(B,C,H,W) -> contiguous (B,H,W,C) [test helper]"""
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()
