"""Drop-in for the reference's model/resnet_deconv.py: `get_deconv_net(layers, num_classes, downsample)`
(resnet_deconv.py:8-16) returning an nn.Module-compatible object whose forward/backward run on the
hand-written HIP kernels: ResNet18 (BasicBlock; BASELINE.json configs 1, 2, 4) and the Bottleneck variants 50 / 101 / 152
(resnet_deconv.py:10-12, :177-215; not in any BASELINE config, same kernels and plan builder)."""
from . import _lib as L
from .nets import ResNetDeconv


def get_deconv_net(layers, num_classes, downsample):
    if layers not in (18, 50, 101, 152):
        raise L.AwrError("get_deconv_net(%r, ...): the reference builds ResNet 18 / 50 / 101 / 152 (resnet_deconv.py:9-13)" % (layers,))
    if downsample not in (1, 2, 4, 8):
        raise L.AwrError("downsample must be one of 1,2,4,8")
    return ResNetDeconv(num_classes, downsample, layers)
