"""awr_amd -- MI355X-native implementation of the AWR (Adaptive Weighting Regression) hot path.

Host side (Python, mirrors the reference's call surface) over the C ABI of libawr_hip.so
(include/awr_hip.h, hand-written HIP for gfx950).  Importing this package does not load the
library; the first use of any hot-path object does, and fails loudly if it is missing.
"""
__version__ = "0.1.0"


def __getattr__(name):
    import importlib
    table = {
        "FeatureModule": ("feature_tool", "FeatureModule"),
        "My_SmoothL1Loss": ("loss", "My_SmoothL1Loss"),
        "get_deconv_net": ("resnet_deconv", "get_deconv_net"),
        "PoseNet": ("hourglass", "PoseNet"),
        "Trainer": ("trainer", "Trainer"),
        "TrainEngine": ("trainer", "TrainEngine"),
        "opt": ("config", "opt"),
    }
    if name in table:
        mod, attr = table[name]
        return getattr(importlib.import_module("." + mod, __name__), attr)
    raise AttributeError(name)
