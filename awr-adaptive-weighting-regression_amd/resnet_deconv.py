"""Drop-in for the reference's model/resnet_deconv.py: `get_deconv_net(layers, num_classes, downsample)`
(resnet_deconv.py:8-16) returning an nn.Module-compatible object whose forward/backward run on the
hand-written HIP kernels.  Only ResNet18 is on the hot path named by BASELINE.json (configs 1,2,4);
the Bottleneck variants (50/101/152) are reported as unsupported rather than silently emulated."""
from . import _lib as L
from .nets import ResNet18Deconv


def get_deconv_net(layers, num_classes, downsample):
    if layers != 18:
        raise L.AwrError("get_deconv_net(%r, ...): only the ResNet18-deconv backbone is implemented on the MI355X path" % (layers,))
    if downsample not in (1, 2, 4, 8):
        raise L.AwrError("downsample must be one of 1,2,4,8")
    return ResNet18Deconv(num_classes, downsample)
