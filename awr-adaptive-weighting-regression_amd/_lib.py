"""ctypes binding of libawr_hip.so (the C ABI declared in include/awr_hip.h).

The product path has NO fallback: if the shared library is missing this module raises at import.
Set AWR_AUTO_BUILD=1 to let it invoke hipcc once (the same thing __graft_entry__.build() does).
"""
import ctypes as C
import os

import torch

from . import build as _build

c_f32p = C.c_void_p      # device pointers travel as integers
c_stream = C.c_void_p


class AwrError(RuntimeError):
    pass


def _load():
    probe = os.environ.get("AWR_LIB_PATH")      # kernel-study builds (tools/probe_gemm.sh); never set in production
    if probe:
        return C.CDLL(probe)
    if not os.path.exists(_build.LIB):
        if os.environ.get("AWR_AUTO_BUILD", "0") == "1":
            _build.build_lib(verbose=False)
        else:
            raise AwrError("libawr_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or set AWR_AUTO_BUILD=1. There is no CPU/PyTorch fallback for the AWR hot path." % _build.LIB)
    return C.CDLL(_build.LIB)


lib = _load()


class Phase(C.Structure):
    _fields_ = [("py", C.c_int), ("px", C.c_int), ("ntaps", C.c_int), ("tap", C.c_int32 * 16)]


class ConvArgs(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("w", C.c_void_p), ("out", C.c_void_p),
                ("in_scale", C.c_void_p), ("in_shift", C.c_void_p), ("bias", C.c_void_p),
                ("out_scale", C.c_void_p), ("out_shift", C.c_void_p), ("res", C.c_void_p), ("stats", C.c_void_p),
                ("bnr_y", C.c_void_p), ("bnr_coef", C.c_void_p),
                ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Cin", C.c_int),
                ("Hq", C.c_int), ("Wq", C.c_int),
                ("Hout", C.c_int), ("Wout", C.c_int), ("N", C.c_int),
                ("so", C.c_int), ("si", C.c_int), ("T", C.c_int),
                ("relu_in", C.c_int), ("relu_out", C.c_int), ("nphase", C.c_int), ("tile_m", C.c_int), ("tile_n", C.c_int),
                ("ph", Phase * 4), ("w_split", C.c_void_p), ("stat_slots", C.c_int), ("stat_slot_base", C.c_int), ("in2", C.c_void_p), ("Cin1", C.c_int),
                ("partial", C.c_void_p), ("split_k", C.c_int), ("split_max", C.c_int), ("bnr_act", C.c_void_p),
                ("bnr2_y", C.c_void_p), ("bnr2_coef", C.c_void_p), ("stats2", C.c_void_p), ("w2", C.c_void_p), ("bias2", C.c_void_p),
                ("N1", C.c_int), ("N1x", C.c_int), ("in_bnb_y", C.c_void_p), ("in_bnb_coef", C.c_void_p), ("accum", C.c_int), ("in_split", C.c_void_p), ("pool_out", C.c_void_p), ("out_nt", C.c_int)]


class PackJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("split", C.c_void_p), ("d0", C.c_int), ("d1", C.c_int), ("T", C.c_int), ("transpose", C.c_int),
                ("rows", C.c_int), ("ld", C.c_int), ("first", C.c_int64), ("cols", C.c_int), ("reserved", C.c_int)]


class UnpackJob(C.Structure):
    _fields_ = [("packed", C.c_void_p), ("grad", C.c_void_p), ("d0", C.c_int), ("d1", C.c_int), ("T", C.c_int), ("ld", C.c_int),
                ("first", C.c_int64), ("slots", C.c_int), ("slot_stride", C.c_int)]


def job_table(jobs, device):
    """ctypes structs -> device byte tensor (the batched kernels read the table from HBM)."""
    arr = (type(jobs[0]) * len(jobs))(*jobs)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


class WgradArgs(C.Structure):
    _fields_ = [("D", C.c_void_p), ("G", C.c_void_p), ("R", C.c_void_p),
                ("d_scale", C.c_void_p), ("d_shift", C.c_void_p), ("g_scale", C.c_void_p), ("g_shift", C.c_void_p),
                ("d_colsum", C.c_void_p), ("d_relu", C.c_int), ("g_relu", C.c_int),
                ("B", C.c_int), ("Hd", C.c_int), ("Wd", C.c_int), ("Cd", C.c_int),
                ("Hg", C.c_int), ("Wg", C.c_int), ("Cg", C.c_int), ("sg", C.c_int), ("T", C.c_int), ("ld", C.c_int),
                ("tile_m", C.c_int), ("tile_n", C.c_int), ("target_blocks", C.c_int), ("algo", C.c_int),
                ("dy", C.c_int8 * 16), ("dx", C.c_int8 * 16), ("split_stride", C.c_int64), ("max_split", C.c_int)]


class WinoArgs(C.Structure):
    """awr_wino_args (include/awr_hip.h)"""
    _fields_ = [("in_", C.c_void_p), ("U", C.c_void_p), ("bias", C.c_void_p), ("in_scale", C.c_void_p), ("in_shift", C.c_void_p), ("out", C.c_void_p),
                ("stats", C.c_void_p), ("res", C.c_void_p), ("bnr_y", C.c_void_p), ("bnr_coef", C.c_void_p), ("bnr_act", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("N", C.c_int), ("relu", C.c_int), ("relu_in", C.c_int), ("nslots", C.c_int),
                ("out_scale", C.c_void_p), ("out_shift", C.c_void_p)]


class NyuSample(C.Structure):
    """awr_nyu_sample (include/awr_hip.h): one image of a device-side NYU batch."""
    _fields_ = [("frame", C.c_int64), ("ustart", C.c_int32), ("vstart", C.c_int32), ("cw", C.c_int32), ("ch", C.c_int32),
                ("rw", C.c_int32), ("rh", C.c_int32), ("ox", C.c_int32), ("oy", C.c_int32), ("ifx", C.c_double), ("ify", C.c_double),
                ("zstart", C.c_double), ("zend", C.c_double), ("op", C.c_int32), ("norm32", C.c_int32), ("m", C.c_double * 9),
                ("zstart2", C.c_double), ("zend2", C.c_double), ("lo", C.c_double), ("far", C.c_double), ("center_z", C.c_double),
                ("half", C.c_double)]


_I, _F, _L, _P, _D = C.c_int, C.c_float, C.c_int64, C.c_void_p, C.c_double
_PP = C.POINTER(C.c_void_p)
BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)      # awr_bucket_cb
_SIGS = {
    "awr_version": ([], C.c_int),
    "awr_last_error": ([], C.c_char_p),
    "awr_device_info": ([C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, _I], C.c_int),
    "awr_head_forward": ([_P, _P, _I, _I, _I, _I, _F, _P, _P, _P], C.c_int),
    "awr_head_backward": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _I, _P], C.c_int),
    "awr_joint2offset": ([_P, _P, _I, _I, _I, _I, _F, _P, _P], C.c_int),
    "awr_huber": ([_P, _P, _L, _F, _F, _P, _P, _I, _P], C.c_int),
    "awr_dense_loss": ([_P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P, _P, _I, _P], C.c_int),
    "awr_zero_f64": ([_P, _L, _P], C.c_int),
    "awr_loss_finalize": ([_P, _I, _P, _P], C.c_int),
    "awr_loss_finalize_reset": ([_P, _I, _P, _P], C.c_int),
    "awr_adam_step": ([_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _L, _F, _P], C.c_int),
    "awr_sgd_step": ([_P, _P, _P, _L, _F, _F, _F, _L, _F, _P], C.c_int),
    "awr_pack_weight": ([_P, _I, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "awr_unpack_wgrad": ([_P, _I, _I, _I, _I, _P, _I, _P], C.c_int),
    "awr_pack_weights_batched": ([_P, _I, _L, _P], C.c_int),
    "awr_unpack_wgrads_batched": ([_P, _I, _L, _P], C.c_int),
    "awr_conv_gemm": ([C.POINTER(ConvArgs), _P], C.c_int),
    "awr_maxpool_fwd_stats": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P], C.c_int),
    "awr_upsample2_add_stats": ([_P, _P, _I, _I, _I, _I, _P, _P, _I, _P], C.c_int),
    "awr_conv_wgrad": ([C.POINTER(WgradArgs), _P], C.c_int),
    "awr_conv_wgrad_algo_ok": ([C.POINTER(WgradArgs), _I], C.c_int),
    "awr_debug_force_tile": ([_I, _I], C.c_int),
    "awr_debug_set_knob": ([C.c_char_p, _I], C.c_int),
    "awr_set_gemm_products": ([_I], C.c_int),
    "awr_split_weight": ([_P, _P, _L, _P], C.c_int),
    "awr_get_gemm_products": ([], C.c_int),
    "awr_get_wgrad_products": ([], C.c_int),
    "awr_set_gemm_staging": ([_I], C.c_int),
    "awr_get_gemm_staging": ([], C.c_int),
    "awr_set_gemm_accum": ([_I], C.c_int),
    "awr_get_gemm_accum": ([], C.c_int),
    "awr_set_gemm_accum_auto": ([_I, _I], C.c_int),
    "awr_get_gemm_accum_auto": ([C.POINTER(_I), C.POINTER(_I)], C.c_int),
    "awr_resolve_gemm_accum": ([_I, _I], C.c_int),
    "awr_stem_im2col": ([_P, _I, _I, _I, _P, _P], C.c_int),
    "awr_stem_stats": ([_P, _P, _P, _I, _I, _I, _P, _I, _P], C.c_int),
    "awr_stem_conv": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P], C.c_int),
    "awr_stem_slots": ([_I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "awr_set_deterministic": ([_I], C.c_int),
    "awr_get_deterministic": ([], C.c_int),
    "awr_conv_wgrad_splits": ([C.POINTER(WgradArgs), C.POINTER(C.c_int)], C.c_int),
    "awr_stem_pool": ([_P, _P, _P, _P, _I, _I, _I, _P, _P, _P], C.c_int),
    "awr_stem_bwd_reduce": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P], C.c_int),
    "awr_stem_bwd_wgrad": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P], C.c_int),
    "awr_bn_bwd_finalize": ([_P, _I, _L, _P, _P, _P, _P, _P, _I, _I, _P], C.c_int),
    "awr_bn_bwd_finalize_lin": ([_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P], C.c_int),
    "awr_bn_bwd_apply_only": ([_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P], C.c_int),
    "awr_bn_finalize": ([_P, _I, _L, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _P], C.c_int),
    "awr_bn_fold_eval": ([_I, _P, _P, _P, _P, _F, _P, _P, _P], C.c_int),
    "awr_channel_stats": ([_P, _L, _I, _P, _I, _P], C.c_int),
    "awr_bn_apply": ([_P, _P, _P, _P, _I, _P, _L, _I, _P], C.c_int),
    "awr_bn_bwd_reduce": ([_P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _I, _P], C.c_int),
    "awr_bn_bwd_apply": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _I, _I, _P], C.c_int),
    "awr_relu_bwd": ([_P, _P, _P, _L, _P], C.c_int),
    "awr_add": ([_P, _P, _P, _L, _P], C.c_int),
    "awr_bias_grad": ([_P, _L, _I, _P, _I, _P], C.c_int),
    "awr_maxpool_fwd": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P], C.c_int),
    "awr_maxpool_bwd": ([_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P], C.c_int),
    "awr_upsample2_add": ([_P, _P, _I, _I, _I, _I, _P, _P], C.c_int),
    "awr_upsample2_bwd": ([_P, _I, _I, _I, _I, _P, _I, _P], C.c_int),
    "awr_nhwc_to_nchw": ([_P, _I, _I, _I, _I, _P, _P], C.c_int),
    "awr_nchw_to_nhwc": ([_P, _I, _I, _I, _I, _P, _P], C.c_int),
    # network-level API (csrc/awr_net.hip)
    "awr_net_create": ([_I, _I, _I, _I, _PP], C.c_int),
    "awr_net_destroy": ([_P], C.c_int),
    "awr_net_sizes": ([_P, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L), C.POINTER(_L), C.POINTER(_I), C.POINTER(_I)], C.c_int),
    "awr_net_tensor_info": ([_P, _L, C.POINTER(C.c_char_p), C.POINTER(_I), C.POINTER(_I), C.POINTER(_L), C.POINTER(_L), C.POINTER(_I)], C.c_int),
    "awr_net_bind": ([_P, _P, _P, _P], C.c_int),
    "awr_plan_create": ([_P, _I, _I, _I, C.c_uint, _I, _I, _P, _PP, _PP, _PP], C.c_int),
    "awr_plan_destroy": ([_P], C.c_int),
    "awr_plan_info": ([_P, C.POINTER(_L), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)], C.c_int),
    "awr_plan_bucket": ([_P, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_I)], C.c_int),
    "awr_plan_op": ([_P, _I, _I, C.POINTER(C.c_char_p), C.POINTER(_D), C.POINTER(_I)], C.c_int),
    "awr_plan_head_nhwc": ([_P, _I, _PP, _PP, C.POINTER(_I)], C.c_int),
    "awr_plan_set_nhwc_boundary": ([_P, _I], C.c_int),
    "awr_head_nhwc_scratch": ([_I, _I, _I], C.c_int64),
    "awr_head_forward_nhwc": ([_P, _I, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P], C.c_int),
    "awr_head_loss_step_nhwc": ([_P, _I, _P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "awr_plan_tensor": ([_P, _I, C.POINTER(C.c_char_p), C.POINTER(_I), _PP, _PP, C.POINTER(_I), _PP, _PP], C.c_int),
    "awr_plan_set_streams": ([_P, _I, _I], C.c_int),
    "awr_stream_pool_info": ([C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "awr_plan_set_bucket_callback": ([_P, _P, _P], C.c_int),        # (plan, awr_bucket_cb or NULL, user)
    "awr_plan_refresh_weights": ([_P, _P], C.c_int),
    "awr_plan_forward": ([_P, _P], C.c_int),
    "awr_plan_backward": ([_P, _P], C.c_int),
    "awr_plan_run_timed": ([_P, _I, _P, C.POINTER(_F)], C.c_int),
    "awr_plan_autotune": ([_P, _I, _P], C.c_int),
    "awr_plan_gemm": ([_P, _I, C.POINTER(C.c_char_p), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_F), C.POINTER(_I)], C.c_int),
    "awr_plan_set_gemm": ([_P, _I, _I, _I, _I, _F], C.c_int),
    "awr_plan_gemm_algo": ([_P, _I, C.POINTER(_I)], C.c_int),
    "awr_plan_set_gemm_algo": ([_P, _I, _I], C.c_int),
    # data-parallel API (csrc/awr_dp.hip)
    "awr_dp_available": ([C.POINTER(_I), C.POINTER(C.c_char_p)], C.c_int),
    "awr_dp_unique_id": ([_P], C.c_int),
    "awr_dp_init": ([_I, _I, _P, _PP], C.c_int),
    "awr_dp_destroy": ([_P], C.c_int),
    "awr_dp_info": ([_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)], C.c_int),
    "awr_dp_allreduce": ([_P, _P, _L, _P], C.c_int),
    "awr_dp_broadcast": ([_P, _P, _L, _I, _P], C.c_int),
    "awr_dp_wait": ([_P, _P], C.c_int),
    "awr_plan_set_dp": ([_P, _P], C.c_int),
    # Winograd F(2x2, 3x3) (csrc/awr_wino.hip)
    "awr_wino_weights": ([_P, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "awr_wino_conv3x3": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P], C.c_int),
    "awr_wino2_conv3x3": ([_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P], C.c_int),
    "awr_wino_conv": ([C.POINTER(WinoArgs), _P], C.c_int),
    "awr_wino_dgrad_or_direct": ([C.POINTER(ConvArgs), _P, _P], C.c_int),
    "awr_wino_dgrad_supported": ([C.POINTER(ConvArgs)], C.c_int),
    "awr_set_conv_winograd": ([_I], C.c_int),
    "awr_get_conv_winograd": ([], C.c_int),
    "awr_wino_eligible": ([_I, _I, _I, _I, _I], C.c_int),
    "awr_wino_wgrad": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P], C.c_int),
    "awr_wino_wgrad_scratch": ([_I, _I, _I, _I, _I], C.c_int64),
    "awr_wino_wgrad_eligible": ([_I, _I, _I, _I, _I], C.c_int),
    "awr_plan_winograd": ([_P, C.POINTER(_I), C.POINTER(_D)], C.c_int),
    # NYU data path (csrc/awr_nyu.hip)
    "awr_nyu_crop": ([_P, _I, _I, _I, _P, _I, _I, _P, _P, _P], C.c_int),
    "awr_nyu_warp": ([_P, _I, _I, _P, _I, _F, _I, _I, _I, _P, _P], C.c_int),
    "awr_nyu_normalize": ([_P, _P, _P, _I, _L, _P, _P], C.c_int),
    "awr_nyu_augment": ([_P, _P, _P, _I, _I, _P, _P, _P], C.c_int),
    "awr_nyu_batch": ([_P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P], C.c_int),
}

# entry points of study builds only (hipcc -DAWR_STUDY, AWR_BUILD_STUDY=1 for awr_amd.build): measured-and-rejected forms that the default
# library neither contains nor exports (tests/test_abi.py checks that); HAS_STUDY says which kind of library is loaded
_STUDY_SIGS = {
    "awr_conv_gemm_part": ([C.POINTER(ConvArgs), _I, _I, _P], C.c_int),
    "awr_split_act": ([_P, _P, _P, _I, _L, _I, _P, _P], C.c_int),
}
STUDY_EXPORTS = tuple(_STUDY_SIGS)
HAS_STUDY = all(hasattr(lib, n) for n in _STUDY_SIGS)
if HAS_STUDY:
    _SIGS.update(_STUDY_SIGS)
EXPORTS = tuple(k for k in _SIGS if k not in _STUDY_SIGS)
MISSING = []                           # header/library mismatch; tests/test_abi.py requires this to be empty
for _name, (_args, _ret) in _SIGS.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError:
        MISSING.append(_name)
        continue
    _fn.argtypes = _args
    _fn.restype = _ret


def last_error():
    return lib.awr_last_error().decode()


def check(rc, what=""):
    if rc != 0:
        raise AwrError("%s failed (%d): %s" % (what or "libawr_hip call", rc, last_error()))


def ptr(t):
    """Device pointer of a contiguous fp32/other CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise AwrError("AWR HIP path needs CUDA(HIP) tensors; got a %s tensor -- there is no CPU fallback" % t.device)
    if not t.is_contiguous():
        raise AwrError("tensor must be contiguous")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    if name in MISSING:
        raise AwrError("libawr_hip.so does not export %s -- rebuild it (__graft_entry__.build())" % name)
    check(getattr(lib, name)(*args), name)
