"""Drop-in for the reference's model/hourglass.py: `PoseNet(net, joint_num)` with net = 'hourglass_<n>'
(hourglass.py:105-106); forward returns a list with one (B,4J,F,F) dense map per stack (:165)."""
from . import _lib as L
from .nets import HourglassNet


def PoseNet(net, joint_num, inp_dim=256, bn=False, increase=0, **kwargs):
    if increase != 0:
        raise L.AwrError("PoseNet(increase != 0) is not used by the reference configs and is not implemented")
    try:
        nstack = int(str(net).split("_")[-1])
    except ValueError:
        raise L.AwrError("net must look like 'hourglass_<nstack>', got %r" % (net,))
    return HourglassNet(nstack, joint_num, inp_dim)
