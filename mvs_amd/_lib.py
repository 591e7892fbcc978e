"""ctypes binding of libmvs_hip.so (the C ABI declared in include/mvs_hip.h).

There is deliberately NO fallback: if the HIP library is missing or fails to
load, every op raises.  torch must be imported first so that the HIP runtime
already in the process (torch's bundled libamdhip64, SONAME libamdhip64.so.7)
is the one the library binds to -- device pointers and streams are then shared.
"""
import ctypes
import os

import torch  # noqa: F401  (loads the HIP runtime the library must share)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmvs_hip.so")

MVS_LAYOUT_NCHW = 0
MVS_LAYOUT_NHWC = 1
MVS_LAYOUT_C8 = 2
MVS_LAYOUT_C16 = 3
MVS_LAYOUT_C4 = 4
MVS_LAYOUT_C8H = 5
MVS_LAYOUT_C8P = 6
MVS_LAYOUT_C8PT = 7
MVS_LAYOUT_C8PH = 8

_c_f = ctypes.c_void_p   # device pointers travel as integers
_c_i = ctypes.c_int
_c_l = ctypes.c_int64

_SIGS = {
    "mvs_version": (ctypes.c_int, []),
    "mvs_last_error_string": (ctypes.c_char_p, []),
    "mvs_arch": (ctypes.c_char_p, []),
    "mvs_nchw_to_nhwc_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_l, _c_f]),
    "mvs_nhwc_to_nchw_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_l, _c_f]),
    "mvs_images_u8_to_planar_f32": (_c_i, [_c_f] + [_c_i] * 5 + [_c_f, _c_f]),
    "mvs_proj_matrices_f32": (_c_i, [_c_f, _c_f, ctypes.c_float, _c_i, _c_f, _c_f]),
    "mvs_warp_fwd_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i] + [_c_i] * 6 + [_c_f, _c_f]),
    "mvs_warp_bwd_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i] + [_c_i] * 6 + [_c_f, _c_f]),
    "mvs_costvol_variance_fwd_f32": (_c_i, [_c_f] * 4 + [_c_i] * 11 + [_c_f, _c_f]),
    "mvs_costvol_variance_workspace_bytes": (ctypes.c_size_t, [_c_i] * 8),
    "mvs_costvol_variance_workspace_bytes2": (ctypes.c_size_t, [_c_i] * 9),
    "mvs_costvol_variance_fwd_ws_f32": (_c_i, [_c_f] * 4 + [_c_i] * 12 + [_c_f, _c_f, ctypes.c_size_t, _c_f]),
    "mvs_costvol_variance_handover_bytes": (ctypes.c_size_t, [_c_i] * 5),
    "mvs_costvol_variance_fwd_ws3_f32": (_c_i, [_c_f] * 4 + [_c_i] * 9 + [_c_f, _c_f, _c_f, _c_f, ctypes.c_size_t, _c_f, _c_f, _c_f, _c_f]),
    "mvs_c8p_bytes": (ctypes.c_size_t, [_c_i] * 6),
    "mvs_c8_to_c8p_f32": (_c_i, [_c_f, _c_f] + [_c_i] * 6 + [_c_f, _c_f]),
    "mvs_conv3d_c8p_f16x3_f32": (_c_i, [_c_f] * 7 + [_c_i] * 8 + [_c_f, _c_f, _c_f]),
    "mvs_conv3d_c8_handed_f16x3_f32": (_c_i, [_c_f] * 8 + [_c_i] * 6 + [_c_f, _c_f, _c_f]),
    "mvs_conv3d_f16x3_pack_veto_word": (ctypes.c_void_p, [_c_f, _c_i]),
    "mvs_costreg_fwd4_f32": (_c_i, [_c_f] * 7 + [_c_i] * 7 + [_c_f, ctypes.c_size_t, _c_f, _c_f]),
    "mvs_costvol_variance_fwd_ws2_f32": (_c_i, [_c_f] * 4 + [_c_i] * 12 + [_c_f, _c_f, ctypes.c_size_t, _c_f, _c_f]),
    "mvs_selftest_div_by_views_f32": (_c_i, [_c_i, _c_f, _c_f]),
    "mvs_costvol_variance_bwd_f32": (_c_i, [_c_f] * 5 + [_c_i] * 10 + [_c_f, _c_f, _c_f]),
    "mvs_conv3d_f32": (_c_i, [_c_f] * 6 + [_c_i] * 11 + [_c_f, _c_f]),
    "mvs_conv3d_absmax_f32": (_c_i, [_c_f] * 6 + [_c_i] * 11 + [_c_f, _c_f, _c_f]),
    "mvs_costreg_workspace_bytes": (ctypes.c_size_t, [_c_i] * 5),
    "mvs_costreg_fwd_f32": (_c_i, [_c_f, _c_i, _c_f] + [_c_i] * 7 + [_c_f, ctypes.c_size_t, _c_f, _c_f]),
    "mvs_costreg_fwd2_f32": (_c_i, [_c_f, _c_i, _c_f, _c_f] + [_c_i] * 7 + [_c_f, ctypes.c_size_t, _c_f, _c_f, _c_f]),
    "mvs_costreg_fwd3_f32": (_c_i, [_c_f, _c_i, _c_f, _c_f, _c_f] + [_c_i] * 7 + [_c_f, ctypes.c_size_t, _c_f, _c_f, _c_f]),
    "mvs_costreg_tail_packed_bytes": (ctypes.c_size_t, []),
    "mvs_costreg_tail_pack_weights_f32": (_c_i, [_c_f, _c_f, _c_f]),
    "mvs_costreg_tail_supported": (_c_i, [_c_i] * 4),
    "mvs_costreg_tail_f16_f32": (_c_i, [_c_f] * 10 + [_c_i] * 4 + [_c_f, _c_f, _c_f]),
    "mvs_costreg_tail_guarded_f16_f32": (_c_i, [_c_f] * 8 + [_c_i] * 4 + [_c_f] * 4),
    "mvs_conv3d_packed_weight_floats": (_c_l, [_c_i] * 4),
    "mvs_conv3d_pack_weights_f32": (_c_i, [_c_f] + [_c_i] * 4 + [_c_f, _c_f]),
    "mvs_conv3d_mfma_supported": (_c_i, [_c_i] * 4),
    "mvs_conv3d_bf16x6_packed_bytes": (ctypes.c_size_t, [_c_i]),
    "mvs_conv3d_pack_weights_bf16x6_f32": (_c_i, [_c_f, _c_i, _c_f, _c_f]),
    "mvs_conv3d_c8_bf16x6_f32": (_c_i, [_c_f] * 5 + [_c_i] * 6 + [_c_f, _c_f]),
    "mvs_conv3d_f16x3_packed_bytes": (ctypes.c_size_t, [_c_i]),
    "mvs_conv3d_pack_weights_f16x3_f32": (_c_i, [_c_f, _c_i, _c_f, _c_f]),
    "mvs_absmax_f32": (_c_i, [_c_f, _c_l, _c_f, _c_f]),
    "mvs_guard_fallback_count": (_c_i, [ctypes.POINTER(ctypes.c_ulonglong)]),
    "mvs_guard_resolve_all_devices": (_c_i, []),
    "mvs_stream_capture_id": (_c_i, [_c_f, ctypes.POINTER(ctypes.c_ulonglong)]),
    "mvs_conv3d_c8_f16x3_f32": (_c_i, [_c_f] * 6 + [_c_i] * 6 + [_c_f, _c_f, _c_f]),
    "mvs_conv_split_supported": (_c_i, [_c_i] * 4),
    "mvs_conv_split_packed_bytes": (ctypes.c_size_t, [_c_i] * 4),
    "mvs_conv_split_pack_weights_f32": (_c_i, [_c_f] + [_c_i] * 4 + [_c_f, _c_f]),
    "mvs_pack_batch_begin": (_c_i, []),
    "mvs_pack_batch_end": (_c_i, [_c_f]),
    "mvs_conv_split_f32": (_c_i, [_c_f] * 5 + [_c_i] * 10 + [_c_f, _c_f]),
    "mvs_conv_split_f16_packed_bytes": (ctypes.c_size_t, [_c_i] * 4),
    "mvs_conv_split_pack_weights_f16_f32": (_c_i, [_c_f] + [_c_i] * 4 + [_c_f, _c_f]),
    "mvs_conv_split_f16_f32": (_c_i, [_c_f] * 6 + [_c_i] * 10 + [_c_f, _c_f, _c_f]),
    "mvs_conv2d_pair_packed_bytes": (ctypes.c_size_t, [_c_i]),
    "mvs_conv2d_pair_supported": (_c_i, [_c_i] * 4),
    "mvs_conv2d_pair_pack_weights_f32": (_c_i, [_c_f, _c_f, _c_i, _c_f, _c_f]),
    "mvs_conv2d_pair_f16_f32": (_c_i, [_c_f] * 7 + [_c_i] * 6 + [_c_f] * 4),
    "mvs_conv2d_pair_guarded_f16_f32": (_c_i, [_c_f] * 9 + [_c_i] * 6 + [_c_f] * 5),
    "mvs_deconv_split_supported": (_c_i, [_c_i] * 2),
    "mvs_deconv_split_packed_bytes": (ctypes.c_size_t, [_c_i] * 2),
    "mvs_deconv_split_pack_weights_f32": (_c_i, [_c_f] + [_c_i] * 2 + [_c_f, _c_f]),
    "mvs_deconv_split_f16_packed_bytes": (ctypes.c_size_t, [_c_i] * 2),
    "mvs_deconv_split_pack_weights_f16_f32": (_c_i, [_c_f] + [_c_i] * 2 + [_c_f, _c_f]),
    "mvs_deconv_split_f16_f32": (_c_i, [_c_f] * 6 + [_c_i] * 7 + [_c_f, _c_f, _c_f]),
    "mvs_deconv_split_f32": (_c_i, [_c_f] * 5 + [_c_i] * 7 + [_c_f, _c_f]),
    "mvs_conv3d_wgrad_f32": (_c_i, [_c_f, _c_f] + [_c_i] * 7 + [_c_f, _c_f, ctypes.c_size_t, _c_f]),
    "mvs_conv3d_wgrad_workspace_bytes": (ctypes.c_size_t, [_c_i] * 7),
    "mvs_conv3d_wgrad_c8_f32": (_c_i, [_c_f, _c_f] + [_c_i] * 5 + [_c_f, _c_f, ctypes.c_size_t, _c_f]),
    "mvs_conv3d_wgrad_c8_f16_supported": (_c_i, [_c_i] * 5),
    "mvs_conv3d_wgrad_c8_f16_workspace_bytes": (ctypes.c_size_t, [_c_i] * 5),
    "mvs_conv3d_wgrad_c8_f16_f32": (_c_i, [_c_f] * 4 + [_c_i] * 5 + [_c_f, _c_f, ctypes.c_size_t, _c_f]),
    "mvs_conv2d_wgrad_f32": (_c_i, [_c_f, _c_f] + [_c_i] * 8 + [_c_f, _c_f, ctypes.c_size_t, _c_f]),
    "mvs_conv2d_wgrad_workspace_bytes": (ctypes.c_size_t, [_c_i] * 7),
    "mvs_interleave2x2_f32": (_c_i, [_c_f] + [_c_i] * 4 + [_c_f, _c_f]),
    "mvs_rot_trans_f32": (_c_i, [_c_f, _c_i, _c_i, _c_f, _c_f]),
    "mvs_conv3d_wgrad_supported": (_c_i, [_c_i] * 3),
    "mvs_bn_train_workspace_bytes": (ctypes.c_size_t, [_c_i]),
    "mvs_bn_train_fwd_f32": (_c_i, [_c_f] * 4 + [_c_l, _c_i, ctypes.c_float, ctypes.c_float, _c_i] + [_c_f] * 7
                             + [ctypes.c_size_t, _c_f]),
    "mvs_bn_train_bwd_f32": (_c_i, [_c_f] * 6 + [_c_l, _c_i, _c_i] + [_c_f] * 4 + [ctypes.c_size_t, _c_f]),
    "mvs_bn_train_fwd_groups_f32": (_c_i, [_c_f] * 4 + [_c_i, _c_l, _c_i, ctypes.c_float, ctypes.c_float, _c_i] + [_c_f] * 7
                                    + [ctypes.c_size_t, _c_f]),
    "mvs_bn_train_bwd_groups_f32": (_c_i, [_c_f] * 6 + [_c_i, _c_l, _c_i, _c_i] + [_c_f] * 4 + [ctypes.c_size_t, _c_f]),
    "mvs_cas_depth_hypotheses_f32": (_c_i, [_c_f] + [_c_i] * 8 + [ctypes.c_float, _c_f, _c_f]),
    "mvs_geo_consistency_f32": (_c_i, [_c_f] * 3 + [_c_i] * 3 + [_c_f] * 6),
    "mvs_cvp_interval_sum_f64": (_c_i, [_c_f, _c_f, _c_i, _c_i, ctypes.c_double, _c_f, _c_f]),
    "mvs_downsample_bilinear_half_f32": (_c_i, [_c_f, _c_l, _c_i, _c_i, _c_f, _c_f]),
    "mvs_upsample_bicubic2x_f32": (_c_i, [_c_f, _c_l, _c_i, _c_i, _c_f, _c_f]),
    "mvs_cvp_hypothesis_mats_f64": (_c_i, [_c_f] * 6),
    "mvs_cvp_hypotheses_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "mvs_fusibile_fuse_f32": (_c_i, [_c_f] * 3 + [_c_i] * 4 + [ctypes.c_float, ctypes.c_float, _c_i] + [_c_f] * 4),
    "mvs_conv2d_f32": (_c_i, [_c_f] * 5 + [_c_i] * 9 + [_c_f, _c_f]),
    "mvs_conv2d_absmax_f32": (_c_i, [_c_f] * 5 + [_c_i] * 9 + [_c_f, _c_f, _c_f]),
    "mvs_feature_head_supported": (_c_i, [_c_i] * 2),
    "mvs_fpn_tail_supported": (_c_i, [_c_i] * 2),
    "mvs_fpn_tail_packed_bytes": (ctypes.c_size_t, []),
    "mvs_fpn_tail_pack_weights_f32": (_c_i, [_c_f, _c_f, _c_f]),
    "mvs_fpn_tail_f32": (_c_i, [_c_f] * 6 + [_c_i] * 3 + [_c_f, _c_f]),
    "mvs_feature_head_packed_bytes": (ctypes.c_size_t, []),
    "mvs_feature_head_pack_weights_f32": (_c_i, [_c_f, _c_f, _c_f]),
    "mvs_feature_head_f32": (_c_i, [_c_f] * 7 + [_c_i] * 3 + [_c_f, _c_f]),
    "mvs_feature_head_absmax_f32": (_c_i, [_c_f] * 7 + [_c_i] * 3 + [_c_f, _c_f, _c_f]),
    "mvs_conv2d_packed_weight_floats": (_c_l, [_c_i] * 4),
    "mvs_conv2d_pack_weights_f32": (_c_i, [_c_f] + [_c_i] * 4 + [_c_f, _c_f]),
    "mvs_conv2d_supported": (_c_i, [_c_i] * 4),
    "mvs_softmax_regress_conf_f32": (_c_i, [_c_f, _c_f] + [_c_i] * 6 + [_c_f] * 4),
    "mvs_softmax_regress_bwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_f] + [_c_i] * 4 + [_c_f, _c_f]),
}

# only in the tuning build (include/mvs_hip_tuning.h; MVS_HIP_TUNING=1 loads libmvs_hip_tuning.so)
_TUNING_SIGS = {
    "mvs_c8h_bytes": (ctypes.c_size_t, [_c_i] * 5),
    "mvs_c8_to_c8h_f32": (_c_i, [_c_f, _c_f] + [_c_i] * 5 + [_c_f, _c_f]),
    "mvs_conv3d_c8h_f16x3_f32": (_c_i, [_c_f] * 6 + [_c_i] * 6 + [_c_f, _c_f, _c_f]),
}


class ConvLayer(ctypes.Structure):
    """mvs_conv_layer of include/mvs_hip.h"""
    _fields_ = [("weight", ctypes.c_void_p), ("packed", ctypes.c_void_p),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("packed_split", ctypes.c_void_p)]


_lib = None


class MvsHipError(RuntimeError):
    pass


def load():
    """Load libmvs_hip.so (once).  Raises MvsHipError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if os.environ.get("MVS_HIP_TUNING") == "1":   # kernel-tuning scripts: the -DMVS_TUNING build (python -m mvs_amd.build --tuning)
        path = LIB_PATH.replace(".so", "_tuning.so")
    if not os.path.exists(path):
        raise MvsHipError(
            f"{path} not found: build it with `python -m mvs_amd.build` (hipcc, gfx950). "
            "mvs_amd has no CPU or PyTorch fallback for the cost-volume path.")
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    sigs = dict(_SIGS)
    if path != LIB_PATH:
        sigs.update(_TUNING_SIGS)
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here = ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # resolve the range guard's device counter now (hipGetSymbolAddress on first use may load the code object: not something to
    # meet for the first time inside a HIP-graph capture)
    # -- on every visible device (a DataParallel replica's first launch on another device would otherwise do it)
    if torch.cuda.is_available():
        lib.mvs_guard_resolve_all_devices()
    return lib


def exported_symbols():
    return sorted(_SIGS)


def tuning_build_loaded():
    """True when the library in this process is the -DMVS_TUNING build (it also exports include/mvs_hip_tuning.h)."""
    return os.environ.get("MVS_HIP_TUNING") == "1"


def check(rc, what):
    if rc != 0:
        msg = load().mvs_last_error_string().decode(errors="replace")
        raise MvsHipError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a contiguous float32 CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MvsHipError("mvs_amd ops need device (HIP) tensors; got a CPU tensor")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise MvsHipError(f"expected contiguous float32, got {t.dtype} contiguous={t.is_contiguous()}")
    if t.device.index != torch.cuda.current_device():
        # kernels launch on the CURRENT device and its current stream (stream() below): a tensor
        # of another GPU would be dereferenced there.  nn.DataParallel sets the device per replica.
        raise MvsHipError(f"tensor on cuda:{t.device.index} but the current device is "
                          f"cuda:{torch.cuda.current_device()}: wrap the call in torch.cuda.device(...)")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
