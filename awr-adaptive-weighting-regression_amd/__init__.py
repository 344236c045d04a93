"""awr_amd -- MI355X-native implementation of the AWR (Adaptive Weighting Regression) hot path.

Host side (Python, mirrors the reference's call surface) over the C ABI of libawr_hip.so
(include/awr_hip.h, hand-written HIP for gfx950).  Importing this package does not load the
library; the first use of any hot-path object does, and fails loudly if it is missing.
"""
__version__ = "0.1.0"


def set_gemm_products(n):
    """Process-wide choice of how the conv GEMMs form their fp32 products (include/awr_hip.h: awr_set_gemm_products):
    1 = FP32 MFMA (default), 6 = exact 3-way bf16 split of both operands, 6 bf16 MFMA partial products per fp32 product.
    Takes effect for kernels launched (or hipGraphs captured) afterwards; autotuned tile choices are kept per mode."""
    from . import _lib as L
    L.call("awr_set_gemm_products", int(n))


def get_gemm_products():
    from . import _lib as L
    return int(L.lib.awr_get_gemm_products())


def set_deterministic(on=True):
    """Process-wide deterministic mode (include/awr_hip.h: awr_set_deterministic): plans built afterwards give every producer
    workgroup its own accumulator copy and every split-K chunk its own copy of the weight gradient, summed in a fixed order --
    two runs on the same inputs are bitwise identical (losses, gradients, parameters).  Slower; the default ($AWR_DETERMINISTIC
    unset) combines partial sums with atomics."""
    from . import _lib as L
    L.call("awr_set_deterministic", int(bool(on)))


def get_deterministic():
    from . import _lib as L
    return bool(L.lib.awr_get_deterministic())


def set_gemm_accum(mode):
    """Process-wide accumulation order of the forward / data-gradient GEMMs of plans built afterwards (include/awr_hip.h:
    awr_set_gemm_accum): "ordered" / 0 = one k-ordered chain per output element (fastest), "blocked" / 1 = the chain restarts every 128 k into
    a second accumulator set -- a convolution's rounding error falls to torch-CPU's (the parity mode), "auto" / 2 (the default) = blocked only
    on the forward launches of TRAINING plans with a long K extent, everything else ordered."""
    from . import _lib as L
    L.call("awr_set_gemm_accum", {"ordered": 0, "blocked": 1, "auto": 2}.get(mode, mode))


def set_conv_winograd(on):
    """Process-wide: plans built afterwards run the FORWARD (True / "forward") or the forward, the data gradient and the weight gradient ("full") of
    every eligible stride-1 3x3 convolution as Winograd F(2x2, 3x3) on the FP32 matrix pipe (include/awr_hip.h: awr_set_conv_winograd;
    csrc/awr_wino.hip) -- 2.25x fewer multiplies, 0.3-1.4x the direct kernels' rounding error, not bit-compatible with them.  Inference plans (InferEngine)
    take the forward form in every non-zero mode."""
    from . import _lib as L
    L.call("awr_set_conv_winograd", _winograd_code(on))


def _winograd_code(on):
    """False / None -> 0, True / "forward" -> 1 (forward launches), "full" -> 2 (forward + data gradients + weight gradients), "forward+wgrad" -> 3 (the data gradients stay direct), "force" -> 6 (tests: "full" on every layer the
    kernel can run, whatever the launch size)"""
    return {"forward": 1, "full": 2, "forward+wgrad": 3, "force": 6}.get(on, 1 if on else 0) if not isinstance(on, int) or isinstance(on, bool) else int(on)


def get_conv_winograd():
    from . import _lib as L
    return int(L.lib.awr_get_conv_winograd())


def get_gemm_accum():
    from . import _lib as L
    return ("ordered", "blocked", "auto")[int(L.lib.awr_get_gemm_accum())]


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
