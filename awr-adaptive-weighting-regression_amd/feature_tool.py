"""FeatureModule: drop-in for the reference's util/feature_tool.py:10-65 on the HIP path.

Same class name, method names, positional arguments and return conventions
(`.float()` tensors), so train.py:113/:118 and test.py:72/:76 work unchanged:

    FM = FeatureModule()
    offset_gt = FM.joint2offset(jt_uvd_gt, img, kernel_size, feature_size)   # (B,4J,F,F)
    jt_uvd    = FM.offset2joint_softmax(offset_pred, img, kernel_size)       # (B,J,3)

offset2joint_softmax participates in autograd (gradient w.r.t. `offset` only -- the reference's
graph has no path to `img` parameters either).  Everything runs in hand-written HIP kernels
(csrc/awr_head.hip); there is no PyTorch fallback.
"""
import torch

from . import _lib as L


def _prep(t):
    return t.detach().contiguous().float()


class _Offset2Joint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, offset, img, ks):
        B, C4, F, F2 = offset.shape
        if C4 % 4 or F != F2:
            raise L.AwrError("offset must be (B,4J,F,F); got %s" % (tuple(offset.shape),))
        J, H = C4 // 4, img.shape[-1]
        off, im = _prep(offset), _prep(img)
        jt = torch.empty(B, J, 3, device=off.device, dtype=torch.float32)
        stat = torch.empty(B, J, 2, device=off.device, dtype=torch.float32)
        L.call("awr_head_forward", L.ptr(off), L.ptr(im), B, J, F, H, float(ks), L.ptr(jt), L.ptr(stat), L.stream())
        ctx.save_for_backward(off, im, jt, stat)
        ctx.ks = float(ks)
        return jt

    @staticmethod
    def backward(ctx, g_jt):
        off, im, jt, stat = ctx.saved_tensors
        B, C4, F, _ = off.shape
        g = torch.empty_like(off)
        L.call("awr_head_backward", L.ptr(off), L.ptr(im), L.ptr(jt), L.ptr(stat), L.ptr(_prep(g_jt)), B, C4 // 4, F, im.shape[-1],
               ctx.ks, L.ptr(g), 0, L.stream())
        return g, None, None


class FeatureModule:
    def joint2offset(self, jt_uvd, img, kernel_size, feature_size):
        """GT dense map (B,4J,F,F) from joints (B,J,3) + depth (B,1,H,H); feature_tool.py:12-39."""
        B, J, _ = jt_uvd.shape
        jt, im = _prep(jt_uvd), _prep(img)
        out = torch.empty(B, 4 * J, feature_size, feature_size, device=im.device, dtype=torch.float32)
        L.call("awr_joint2offset", L.ptr(jt), L.ptr(im), B, J, int(feature_size), im.shape[-1], float(kernel_size), L.ptr(out), L.stream())
        return out

    def offset2joint_softmax(self, offset, img, kernel_size):
        """Dense map (B,4J,F,F) + depth -> joints (B,J,3); feature_tool.py:41-65."""
        return _Offset2Joint.apply(offset, img, kernel_size)
