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


def declared_symbols(study=False):
    """entry points include/awr_hip.h declares for the default build (study=True: those inside #ifdef AWR_STUDY blocks instead)"""
    text = open(os.path.join(REPO, "include", "awr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    blocks = re.findall(r"#ifdef AWR_STUDY(.*?)#endif", text, flags=re.S)
    text = "\n".join(blocks) if study else re.sub(r"#ifdef AWR_STUDY.*?#endif", "", text, flags=re.S)
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


def test_default_library_exports_no_study_entry_points(lib):
    """forms that were built, measured and NOT adopted (pre-cut split-operand GEMM, half-batch BatchNorm-backward
    wavefront) are compiled only with -DAWR_STUDY: the header declares them inside #ifdef AWR_STUDY, the binding lists them apart, and the
    library the product loads has neither their symbols nor their kernels."""
    import subprocess
    from awr_amd import build
    study = declared_symbols(study=True)
    assert sorted(study) == sorted(lib.STUDY_EXPORTS) and len(study) >= 2
    if lib.HAS_STUDY:
        pytest.skip("a study build (-DAWR_STUDY) is loaded")
    for s in study:
        assert not hasattr(lib.lib, s), "the default library exports the study entry point %s" % s
    dyn = subprocess.run(["nm", "-D", "--defined-only", build.LIB], capture_output=True, text=True).stdout
    for needle in ("sdma", "split_act"):
        assert needle not in dyn, "study kernel / entry point %r is in the default library" % needle


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


def test_data_parallel_entry_points_without_gpu(lib):
    """awr_dp_* open librccl.so with dlopen at first use: the probe needs no GPU (the ROCm image ships the library), argument checks run
    before any RCCL call, and nothing here creates a communicator."""
    import ctypes as C
    v, p = C.c_int(), C.c_char_p()
    rc = lib.lib.awr_dp_available(C.byref(v), C.byref(p))
    assert rc in (0, -3)
    if rc == 0:
        assert v.value > 0 and b"rccl" in p.value
    assert lib.lib.awr_dp_init(2, 2, C.create_string_buffer(128), C.byref(C.c_void_p())) == -1 and "rank" in lib.last_error()
    assert lib.lib.awr_dp_unique_id(None) == -1
    assert lib.lib.awr_dp_destroy(None) == 0
    assert lib.lib.awr_plan_set_dp(None, None) == -1


def test_struct_layout_matches_header(lib):
    import ctypes as C
    # awr_phase: 3 ints + 16 packed taps = 76 bytes; awr_conv_args: 12 pointers + 17 ints + 4 phases + pad + w_split
    assert C.sizeof(lib.Phase) == 76
    assert C.sizeof(lib.ConvArgs) == 12 * 8 + 17 * 4 + 4 * 76 + 4 + 8 + 2 * 4 + 8 + 8 + 8 + 2 * 4 + 8 + 3 * 8 + 2 * 8 + 2 * 4 + 2 * 8 + 2 * 4 + 8 + 8 + 8   # padding before w_split; stat_slots, stat_slot_base; in2; Cin1 + padding; partial; split_k, split_max; bnr_act; bnr2_y, bnr2_coef, stats2; w2, bias2; N1, N1x; in_bnb_y, in_bnb_coef; accum + padding; in_split; pool_out; out_nt + padding
    assert lib.ConvArgs.w_split.offset == 472 and lib.ConvArgs.stat_slots.offset == 480
    assert C.sizeof(lib.PackJob) == 3 * 8 + 6 * 4 + 8 + 2 * 4 and C.sizeof(lib.UnpackJob) == 2 * 8 + 4 * 4 + 8 + 2 * 4
    assert C.sizeof(lib.WgradArgs) == 8 * 8 + 16 * 4 + 32 + 8 + 8 and lib.WgradArgs.dy.offset == 8 * 8 + 16 * 4 and lib.WgradArgs.split_stride.offset == 160


def test_product_path_fails_loudly_on_cpu_tensors(lib):
    import torch
    import awr_amd
    with pytest.raises(lib.AwrError):
        awr_amd.FeatureModule().offset2joint_softmax(torch.zeros(1, 56, 64, 64), torch.zeros(1, 1, 128, 128), 0.4)
    net = awr_amd.get_deconv_net(18, 14, 2)
    with pytest.raises(lib.AwrError):
        net(torch.zeros(1, 1, 128, 128))


def test_network_level_layout_without_a_gpu(lib):
    """awr_net_create builds the checkpoint layout on the host only: keys, shapes and kinds of the three reference networks
    must equal the manifest generated from the reference's own modules (tests/golden/statedict_manifest.json), arena offsets
    must be 16-byte aligned and disjoint, never-trained hourglass skip_layers must sit behind n_active."""
    import ctypes as C
    import json
    man = json.load(open(os.path.join(REPO, "tests", "golden", "statedict_manifest.json")))
    L = lib
    for name, (kind, nstack, J) in {"resnet_18_J14": (0, 1, 14), "resnet_50_J14": (0, 50, 14), "resnet_101_J14": (0, 101, 14),
                                    "hourglass_1_J14": (1, 1, 14), "hourglass_2_J21": (1, 2, 21)}.items():
        h = C.c_void_p()
        assert L.lib.awr_net_create(kind, nstack, J, 2, C.byref(h)) == 0, L.last_error()
        nt, npar, nact, nbuf, ncnt, nst = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        assert L.lib.awr_net_sizes(h, C.byref(nt), C.byref(npar), C.byref(nact), C.byref(nbuf), C.byref(ncnt), C.byref(nst)) == 0
        ref = man[name]
        assert nt.value == len(ref) and nst.value == (nstack if kind == 1 else 1)
        key, kd, nd, off, un = C.c_char_p(), C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        shape = (C.c_int64 * 4)()
        spans, n_float, n_unused = [], 0, 0
        for i, (rkey, rshape, rdtype) in enumerate(ref):
            assert L.lib.awr_net_tensor_info(h, i, C.byref(key), C.byref(kd), C.byref(nd), shape, C.byref(off), C.byref(un)) == 0
            assert key.value.decode() == rkey and list(shape[:nd.value]) == rshape, (rkey, list(shape[:nd.value]), rshape)
            assert (kd.value == 7) == (rdtype == "int64")
            if kd.value <= 4:
                n = 1
                for d in rshape:
                    n *= d
                assert off.value % 4 == 0 and off.value + n <= npar.value
                assert (off.value >= nact.value) == bool(un.value)
                spans.append((off.value, off.value + n))
                n_float += n
                n_unused += n if un.value else 0
        spans.sort()
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))          # views never overlap
        assert (kind == 0) == (n_unused == 0) and nact.value <= npar.value
        assert L.lib.awr_net_tensor_info(h, nt.value, None, None, None, None, None, None) == -1      # index out of range
        assert L.lib.awr_plan_create(h, 2, 128, 0, 1, 1, 1, None, None, None, C.byref(C.c_void_p())) == -1      # not bound / null pointers
        assert L.lib.awr_net_destroy(h) == 0
    assert L.lib.awr_net_create(2, 1, 14, 2, C.byref(C.c_void_p())) == -1 and "kind" in L.last_error()
    assert L.lib.awr_net_create(0, 1, 14, 3, C.byref(C.c_void_p())) == -1
    assert L.lib.awr_net_create(0, 34, 14, 2, C.byref(C.c_void_p())) == -1 and "depth" in L.last_error()      # resnet_deconv.py:9-13 builds 18/50/101/152


def test_winograd_entry_points_without_gpu(lib):
    """Round 6: the Winograd entry points validate their arguments before any HIP call; the launch-size rules (awr_wino_eligible /
    awr_wino_wgrad_eligible) and the split-copy scratch size are host arithmetic; awr_wino_args carries the inference epilogue fields; the
    Python mode names map to the C codes."""
    import ctypes as C
    import awr_amd
    L = lib
    assert C.sizeof(L.WinoArgs) == 11 * 8 + 8 * 4 + 2 * 8 and L.WinoArgs.out_scale.offset == 11 * 8 + 8 * 4
    # mode codes
    assert [awr_amd._winograd_code(m) for m in (None, False, True, "forward", "full", "forward+wgrad", "force", 3)] == [0, 0, 1, 1, 2, 3, 6, 3]
    assert L.lib.awr_set_conv_winograd(-1) == -1 and L.lib.awr_set_conv_winograd(64) == -1
    was = L.lib.awr_get_conv_winograd()
    try:
        assert L.lib.awr_set_conv_winograd(0) == 0
        # forward / data gradient: power-of-two maps of 8 x 8 and up, C % 8, N % 32, at least 256 workgroups
        assert L.lib.awr_wino_eligible(64, 64, 64, 128, 128) == 1           # Hourglass 3x3 at full resolution
        assert L.lib.awr_wino_eligible(64, 8, 8, 512, 512) == 1             # ResNet18 layer4 at batch 64: 16 tiles x 16 channel tiles
        assert L.lib.awr_wino_eligible(64, 8, 8, 128, 128) == 0             # ... a Hourglass level of that size: 64 workgroups
        assert L.lib.awr_wino_eligible(64, 4, 4, 512, 512) == 0 and L.lib.awr_wino_eligible(64, 24, 24, 64, 64) == 0 and L.lib.awr_wino_eligible(64, 64, 64, 60, 64) == 0
        # weight gradient: C and N multiples of 64, at least 8 stages (2 x 4 patch blocks) per split
        assert L.lib.awr_wino_wgrad_eligible(64, 64, 64, 128, 128) == 1 and L.lib.awr_wino_wgrad_eligible(64, 32, 32, 64, 64) == 1
        assert L.lib.awr_wino_wgrad_eligible(2, 64, 64, 128, 128) == 0 and L.lib.awr_wino_wgrad_eligible(64, 64, 64, 96, 128) == 0
        assert L.lib.awr_wino_wgrad_eligible(64, 4, 4, 512, 512) == 0
        # one 16 x C x N tile copy per split (one split per CU's workgroup: 256 / tiles) + the bias column sums
        assert int(L.lib.awr_wino_wgrad_scratch(64, 64, 64, 128, 128)) == 64 * 16 * 128 * 128 + 64 * 128
        assert int(L.lib.awr_wino_wgrad_scratch(64, 8, 8, 512, 512)) == 4 * 16 * 512 * 512 + 4 * 512
        L.lib.awr_set_conv_winograd(4)                                       # tests' mode: wherever the kernels can run
        assert L.lib.awr_wino_eligible(2, 4, 4, 8, 32) == 1 and L.lib.awr_wino_wgrad_eligible(2, 8, 8, 64, 64) == 1 and L.lib.awr_wino_wgrad_eligible(2, 8, 8, 32, 64) == 0
    finally:
        L.lib.awr_set_conv_winograd(was)
    assert L.lib.awr_wino_wgrad(None, None, None, None, 0, 1, 8, 8, 64, 64, None, None, 64, None, None) == -1 and "NULL" in L.last_error()
    assert L.lib.awr_wino_weights(None, 64, 64, 64, 64, 0, None, None) == -1
    assert L.lib.awr_wino_conv(None, None) == -1
