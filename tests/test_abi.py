"""CPU: the C-ABI shared library loads and exports every symbol include/awr_hip.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import awr_amd  # noqa: F401
    from awr_amd import build
    if not os.path.exists(build.LIB):
        build.build_lib(verbose=False)
    from awr_amd import _lib
    return _lib


def declared_symbols():
    text = open(os.path.join(REPO, "include", "awr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(awr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    assert not lib.MISSING, lib.MISSING
    for s in syms:
        assert hasattr(lib.lib, s), "libawr_hip.so does not export %s" % s
        assert s in lib.EXPORTS, "%s is declared in the header but not bound in _lib.py" % s
    for s in lib.EXPORTS:
        assert s in syms, "%s is bound but not declared in include/awr_hip.h" % s


def test_version_and_error_string(lib):
    assert lib.lib.awr_version() >= 100
    assert isinstance(lib.last_error(), str)


def test_argument_validation_without_gpu(lib):
    # argument checks run before any HIP call, so they are testable on a CPU-only box
    rc = lib.lib.awr_head_forward(None, None, 1, 14, 63, 128, 0.4, None, None, None)
    assert rc == -1 and "F % 4" in lib.last_error()
    rc = lib.lib.awr_adam_step(None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None)
    assert rc == -1
    rc = lib.lib.awr_debug_force_tile(3, 1)
    assert rc == -1
    assert lib.lib.awr_debug_force_tile(0, 0) == 0


def test_struct_layout_matches_header(lib):
    import ctypes as C
    # awr_phase: 3 ints + 16 packed taps = 76 bytes; awr_conv_args: 12 pointers + 17 ints + 4 phases + pad + w_split
    assert C.sizeof(lib.Phase) == 76
    assert C.sizeof(lib.ConvArgs) == 12 * 8 + 17 * 4 + 4 * 76 + 4 + 8 + 2 * 4   # 4 bytes of padding before the w_split pointer, two ints after it
    assert lib.ConvArgs.w_split.offset == 472 and lib.ConvArgs.stat_slots.offset == 480
    assert C.sizeof(lib.PackJob) == 3 * 8 + 6 * 4 + 8 and C.sizeof(lib.UnpackJob) == 2 * 8 + 4 * 4 + 8 + 2 * 4
    assert C.sizeof(lib.WgradArgs) == 8 * 8 + 16 * 4 + 32 + 8 + 8 and lib.WgradArgs.dy.offset == 8 * 8 + 16 * 4 and lib.WgradArgs.split_stride.offset == 160


def test_product_path_fails_loudly_on_cpu_tensors(lib):
    import torch
    import awr_amd
    with pytest.raises(lib.AwrError):
        awr_amd.FeatureModule().offset2joint_softmax(torch.zeros(1, 56, 64, 64), torch.zeros(1, 1, 128, 128), 0.4)
    net = awr_amd.get_deconv_net(18, 14, 2)
    with pytest.raises(lib.AwrError):
        net(torch.zeros(1, 1, 128, 128))
