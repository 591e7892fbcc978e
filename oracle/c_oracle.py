"""ctypes front end to oracle/libmvs_oracle.so (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Builds the library with
`make -C oracle` on first use if the .so is missing (gcc only; no GPU).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmvs_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.c_int


def build(force=False):
    src = os.path.join(_HERE, "mvs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmvs_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_f)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"oracle {name} failed rc={rc}")


def rot_trans(src_proj, ref_proj):
    """rows of (src_proj @ inv(ref_proj))[:3,:4] -> [B,12] float32, evaluated in
    fp32 numpy (the product path uses torch for this step, module.py:63-65)."""
    src_proj = np.asarray(src_proj, np.float32)
    ref_proj = np.asarray(ref_proj, np.float32)
    proj = src_proj @ np.linalg.inv(ref_proj).astype(np.float32)
    return np.ascontiguousarray(proj[:, :3, :4].reshape(-1, 12), np.float32)


def warp(src, rt, depth, align_corners=False):
    src, rt, depth = _c(src), _c(rt), _c(depth)
    B, C, H, W = src.shape
    D = depth.shape[1]
    mode = 0 if depth.ndim == 2 else 1
    out = np.empty((B, C, D, H, W), np.float32)
    _check(lib().orc_warp_f32(_p(src), _p(rt), _p(depth), _i(mode), _i(B), _i(C), _i(D), _i(H),
                              _i(W), _i(int(align_corners)), _p(out)), "warp")
    return out


def costvol_variance(ref, srcs, rt, depth, align_corners=False, alias_quirk=False):
    """ref [B,C,H,W]; srcs [V-1,B,C,H,W]; rt [V-1,B,12]; depth [B,D] or [B,D,H,W]."""
    ref, srcs, rt, depth = _c(ref), _c(srcs), _c(rt), _c(depth)
    B, C, H, W = ref.shape
    V = srcs.shape[0] + 1
    D = depth.shape[1]
    mode = 0 if depth.ndim == 2 else 1
    out = np.empty((B, C, D, H, W), np.float32)
    _check(lib().orc_costvol_variance_f32(_p(ref), _p(srcs), _p(rt), _p(depth), _i(mode), _i(B),
                                          _i(V), _i(C), _i(D), _i(H), _i(W),
                                          _i(int(align_corners)), _i(int(alias_quirk)), _p(out)),
           "costvol_variance")
    return out


def costvol_variance_bwd(grad_var, ref, srcs, rt, depth, align_corners=False):
    grad_var, ref, srcs, rt, depth = _c(grad_var), _c(ref), _c(srcs), _c(rt), _c(depth)
    B, C, H, W = ref.shape
    V = srcs.shape[0] + 1
    D = depth.shape[1]
    mode = 0 if depth.ndim == 2 else 1
    g_ref = np.empty_like(ref)
    g_srcs = np.empty_like(srcs)
    _check(lib().orc_costvol_variance_bwd_f32(_p(grad_var), _p(ref), _p(srcs), _p(rt), _p(depth),
                                              _i(mode), _i(B), _i(V), _i(C), _i(D), _i(H), _i(W),
                                              _i(int(align_corners)), _p(g_ref), _p(g_srcs)),
           "costvol_variance_bwd")
    return g_ref, g_srcs


def conv3d(x, w, scale=None, shift=None, residual=None, relu=False, stride=1):
    x, w = _c(x), _c(w)
    scale, shift, residual = _c(scale), _c(shift), _c(residual)
    B, Ci, D, H, W = x.shape
    Co = w.shape[0]
    assert w.shape == (Co, Ci, 3, 3, 3)
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    out = np.empty((B, Co, Do, Ho, Wo), np.float32)
    _check(lib().orc_conv3d_f32(_p(x), _p(w), _p(scale), _p(shift), _p(residual), _i(int(relu)),
                                _i(B), _i(Ci), _i(Co), _i(D), _i(H), _i(W), _i(stride), _p(out)),
           "conv3d")
    return out


def deconv3d(x, w, scale=None, shift=None, residual=None, relu=False, stride=2):
    x, w = _c(x), _c(w)
    scale, shift, residual = _c(scale), _c(shift), _c(residual)
    B, Ci, D, H, W = x.shape
    Co = w.shape[1]
    assert w.shape == (Ci, Co, 3, 3, 3)
    out = np.empty((B, Co, D * stride, H * stride, W * stride), np.float32)
    _check(lib().orc_deconv3d_f32(_p(x), _p(w), _p(scale), _p(shift), _p(residual),
                                  _i(int(relu)), _i(B), _i(Ci), _i(Co), _i(D), _i(H), _i(W),
                                  _i(stride), _p(out)), "deconv3d")
    return out


def softmax_regress_conf(cost, depth, clamp_idx=False, want_prob=False):
    cost, depth = _c(cost), _c(depth)
    B, D, H, W = cost.shape
    mode = 0 if depth.ndim == 2 else 1
    dep = np.empty((B, H, W), np.float32)
    conf = np.empty((B, H, W), np.float32)
    prob = np.empty((B, D, H, W), np.float32) if want_prob else None
    _check(lib().orc_softmax_regress_conf_f32(_p(cost), _p(depth), _i(mode), _i(int(clamp_idx)),
                                              _i(B), _i(D), _i(H), _i(W), _p(dep), _p(conf),
                                              _p(prob)), "softmax_regress_conf")
    return (dep, conf, prob) if want_prob else (dep, conf)


def masked_smooth_l1(est, gt, mask):
    est, gt, mask = _c(est).ravel(), _c(gt).ravel(), _c(mask).ravel()
    out = ctypes.c_float()
    _check(lib().orc_masked_smooth_l1_f32(_p(est), _p(gt), _p(mask), ctypes.c_size_t(est.size),
                                          ctypes.byref(out)), "smooth_l1")
    return float(out.value)


def bn_fold(gamma, beta, mean, var, eps=1e-5):
    """Eval-mode BatchNorm3d as a per-channel affine (module.py:30)."""
    gamma, beta, mean, var = (np.asarray(t, np.float32) for t in (gamma, beta, mean, var))
    scale = (gamma / np.sqrt(var + np.float32(eps))).astype(np.float32)
    shift = (beta - mean * scale).astype(np.float32)
    return scale, shift


COSTREG_LAYERS = (
    # name, kind, Cin, Cout, stride   (mvsnet.py:53-81)
    ("conv0", "conv", 32, 8, 1), ("conv1", "conv", 8, 16, 2), ("conv2", "conv", 16, 16, 1),
    ("conv3", "conv", 16, 32, 2), ("conv4", "conv", 32, 32, 1), ("conv5", "conv", 32, 64, 2),
    ("conv6", "conv", 64, 64, 1), ("conv7", "deconv", 64, 32, 2), ("conv9", "deconv", 32, 16, 2),
    ("conv11", "deconv", 16, 8, 2), ("prob", "conv", 8, 1, 1),
)


def costregnet(x, sd, prefix="cost_regularization."):
    """CostRegNet.forward in eval mode (mvsnet.py:83-93) from a state_dict of
    numpy arrays with the reference's key names."""
    def cbr(name, t, stride):
        s, b = bn_fold(sd[f"{prefix}{name}.bn.weight"], sd[f"{prefix}{name}.bn.bias"],
                       sd[f"{prefix}{name}.bn.running_mean"], sd[f"{prefix}{name}.bn.running_var"])
        return conv3d(t, sd[f"{prefix}{name}.conv.weight"], s, b, None, True, stride)

    def dbr(name, t, skip):
        s, b = bn_fold(sd[f"{prefix}{name}.1.weight"], sd[f"{prefix}{name}.1.bias"],
                       sd[f"{prefix}{name}.1.running_mean"], sd[f"{prefix}{name}.1.running_var"])
        return deconv3d(t, sd[f"{prefix}{name}.0.weight"], s, b, skip, True, 2)

    c0 = cbr("conv0", x, 1)
    c2 = cbr("conv2", cbr("conv1", c0, 2), 1)
    c4 = cbr("conv4", cbr("conv3", c2, 2), 1)
    t = cbr("conv6", cbr("conv5", c4, 2), 1)
    t = dbr("conv7", t, c4)
    t = dbr("conv9", t, c2)
    t = dbr("conv11", t, c0)
    return conv3d(t, sd[f"{prefix}prob.weight"], None, sd[f"{prefix}prob.bias"], None, False, 1)
