"""Synthetic DTU-shaped inputs for the MVSNet cost-volume path.

There is no DTU data (and no network) in either container, so every parity
test, golden fixture and bench line is driven by the deterministic recipe in
SURVEY.md section 8(d): DTU-like intrinsics, cameras on an arc looking at a
point 680 mm in front of the reference view, and the DTU depth sweep
425 mm + k * 2.5 mm * 1.06.

The projection matrices follow the reference loader's contract
(MVSNet/datasets/dtu_yao_eval.py:93-95): a 4x4 matrix whose top 3x4 block is
K_feat @ [R | t] (intrinsics already at feature resolution) and whose last row
is (0, 0, 0, 1).

Only numpy is needed for the geometry; random_state_dict() imports torch lazily.
"""
import math

import numpy as np

# DTU full-resolution intrinsics (1600x1200 images), SURVEY.md 8(d).
DTU_FX, DTU_FY, DTU_CX, DTU_CY = 2892.33, 2883.18, 823.205, 619.071
DTU_DEPTH_MIN = 425.0
DTU_INTERVAL = 2.5 * 1.06  # interval_scale 1.06, MVSNet/eval.sh
DTU_TARGET_Z = 680.0

_THETA_DEG = (0.0, -8.0, 8.0, -4.0, 4.0, -12.0, 12.0)
_PHI_DEG = (0.0, 3.0, -3.0, -6.0, 6.0, 2.0, -2.0)
# rig 1 (second parity scene): wider baselines, camera roll and unequal distances to the
# target, so that footprints are longer and more oblique and more taps leave the source images
_RIGS = {
    0: dict(theta=_THETA_DEG, phi=_PHI_DEG, psi=(0.0,) * 7, dist=(1.0,) * 7),
    1: dict(theta=(0.0, -13.0, 11.0, -6.5, 7.5, -17.0, 16.0), phi=(0.0, 5.0, -4.5, -8.0, 9.0, 3.0, -3.5),
            psi=(0.0, 2.0, -3.0, 1.5, -1.0, 4.0, -2.5), dist=(1.0, 0.96, 1.05, 1.02, 0.93, 1.08, 0.9)),
    # rig 2 (third CVP parity scene): narrow baselines on one side, wide on the other, mostly vertical
    # disparity for two views, stronger roll, cameras closer to and farther from the target
    2: dict(theta=(0.0, -3.0, 15.0, 2.0, -10.0, 6.0, -19.0), phi=(0.0, 9.0, 1.5, -11.0, -2.0, 6.5, 4.0),
            psi=(0.0, -5.0, 3.5, 6.0, -2.0, -7.0, 1.0), dist=(1.0, 1.1, 0.88, 0.97, 1.12, 0.92, 1.04)),
}


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def _rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


def feature_intrinsics(feat_h, feat_w):
    """DTU intrinsics divided by 4 (dtu_yao_eval.py:54), rescaled so the same
    field of view covers a feat_h x feat_w feature map (296x400 is native)."""
    sx, sy = feat_w / 400.0, feat_h / 296.0
    K = np.array([[DTU_FX / 4 * sx, 0, DTU_CX / 4 * sx],
                  [0, DTU_FY / 4 * sy, DTU_CY / 4 * sy],
                  [0, 0, 1]], dtype=np.float64)
    return K


def arc_extrinsics(nviews, rig=0):
    """World->camera [R|t] 4x4 for `nviews` cameras on an arc around the
    target point (0, 0, 680); view 0 is the reference (identity).  rig 0 = SURVEY 8(d)'s recipe,
    rig 1 = the second parity scene (wider baselines, roll, unequal distances)."""
    g = _RIGS[rig]
    assert 1 <= nviews <= len(g["theta"])
    target = np.array([0.0, 0.0, DTU_TARGET_Z])
    out = []
    for i in range(nviews):
        R = _rot_z(math.radians(g["psi"][i])) @ _rot_y(math.radians(g["theta"][i])) @ _rot_x(math.radians(g["phi"][i]))
        C = target - R.T @ np.array([0.0, 0.0, DTU_TARGET_Z * g["dist"][i]])
        t = -R @ C
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = t
        out.append(E)
    return np.stack(out)


def proj_matrices(nviews, feat_h, feat_w, batch=1, rig=0):
    """[B, V, 4, 4] float32 projection matrices in the reference loader's
    convention (top 3x4 = K @ E[:3,:4])."""
    K = feature_intrinsics(feat_h, feat_w)
    E = arc_extrinsics(nviews, rig)
    P = E.copy()
    for i in range(nviews):
        P[i, :3, :4] = K @ E[i, :3, :4]
    P = P.astype(np.float32)
    return np.broadcast_to(P, (batch,) + P.shape).copy()


def cas_proj_matrices(nviews, feat_h, feat_w, batch=1, rig=0):
    """[B, V, 2, 4, 4] CasMVSNet convention: [.,.,0] extrinsic, [.,.,1,:3,:3]
    intrinsic (CasMVSNet/datasets/general_eval.py:158-180)."""
    K = feature_intrinsics(feat_h, feat_w)
    E = arc_extrinsics(nviews, rig)
    P = np.zeros((nviews, 2, 4, 4), dtype=np.float32)
    for i in range(nviews):
        P[i, 0] = E[i]
        P[i, 1, :3, :3] = K
    return np.broadcast_to(P, (batch,) + P.shape).copy()


def depth_values(ndepth, batch=1, interval=DTU_INTERVAL, depth_min=DTU_DEPTH_MIN):
    """[B, D] float32 plane-sweep hypotheses (dtu_yao_eval.py:99-100).
    To keep the full DTU depth range at small D, callers widen `interval`."""
    dv = (depth_min + interval * np.arange(ndepth, dtype=np.float64)).astype(np.float32)
    return np.broadcast_to(dv, (batch, ndepth)).copy()


def sweep_interval(ndepth):
    """Interval that spans the DTU range (192 planes of 2.65 mm) with `ndepth`
    planes, as the reference's eval script does when numdepth is reduced."""
    return DTU_INTERVAL * 192.0 / ndepth


def smooth_features(rng, shape, scale=1.0):
    """Band-limited random feature maps [..., H, W]: white noise blurred by a
    small separable box filter so bilinear sampling sees realistic gradients."""
    x = rng.standard_normal(shape).astype(np.float32)
    for ax in (-1, -2):
        x = (np.roll(x, 1, ax) + x + np.roll(x, -1, ax)) / 3.0
    return (x * scale).astype(np.float32)


def images(rng, batch, nviews, h, w):
    """[B, V, 3, H, W] uniform[0,1) float32 (the loader divides by 255)."""
    return rng.random((batch, nviews, 3, h, w), dtype=np.float32)


def random_state_dict(seed=0, peaked=30.0):
    """Seeded MVSNet(refine=False) weights with the reference's key names and
    PyTorch's default initialisers; BatchNorm running statistics and affine
    parameters are randomised and the `prob` layer scaled by `peaked` so the
    softmax over depth is not flat ("No real data")."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv_w(key, shape, fan_in, bias=False, nb=0):
        bound = (1.0 / fan_in) ** 0.5  # kaiming_uniform(a=sqrt(5)) bound
        sd[key + ".weight"] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        if bias:
            sd[key + ".bias"] = (torch.rand(nb, generator=g) * 2 - 1) * bound

    def bn(key, c):
        sd[key + ".weight"] = 0.5 + torch.rand(c, generator=g)
        sd[key + ".bias"] = 0.2 * torch.randn(c, generator=g)
        sd[key + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[key + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        sd[key + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    for name, ci, co, k in (("conv0", 3, 8, 3), ("conv1", 8, 8, 3), ("conv2", 8, 16, 5),
                            ("conv3", 16, 16, 3), ("conv4", 16, 16, 3), ("conv5", 16, 32, 5),
                            ("conv6", 32, 32, 3)):
        conv_w(f"feature.{name}.conv", (co, ci, k, k), ci * k * k)
        bn(f"feature.{name}.bn", co)
    conv_w("feature.feature", (32, 32, 3, 3), 32 * 9, True, 32)
    for name, ci, co in (("conv0", 32, 8), ("conv1", 8, 16), ("conv2", 16, 16), ("conv3", 16, 32),
                         ("conv4", 32, 32), ("conv5", 32, 64), ("conv6", 64, 64)):
        conv_w(f"cost_regularization.{name}.conv", (co, ci, 3, 3, 3), ci * 27)
        bn(f"cost_regularization.{name}.bn", co)
    for name, ci, co in (("conv7", 64, 32), ("conv9", 32, 16), ("conv11", 16, 8)):
        conv_w(f"cost_regularization.{name}.0", (ci, co, 3, 3, 3), co * 27)
        bn(f"cost_regularization.{name}.1", co)
    conv_w("cost_regularization.prob", (1, 8, 3, 3, 3), 8 * 27, True, 1)
    sd["cost_regularization.prob.weight"] *= peaked
    return sd


def cas_random_state_dict(seed=0, peaked=100.0):
    """Seeded CascadeMVSNet state_dict (reference key names, 934,304 parameters): PyTorch's
    default conv init, BatchNorm affine/statistics perturbed away from identity, `prob`
    scaled so the softmax over depth is peaked like a trained network's.  Deterministic for
    a given torch build (CPU generator), so goldens store the seed, not 3.7 MB of weights."""
    import torch
    from .models.cas_mvsnet import CascadeMVSNet
    torch.manual_seed(seed)
    net = CascadeMVSNet()
    g = torch.Generator().manual_seed(seed + 1)
    sd = net.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if k.endswith("bn.weight") or k.endswith("running_var"):
                v.copy_(0.5 + torch.rand(v.shape, generator=g))
            elif k.endswith("bn.bias") or k.endswith("running_mean"):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif k.endswith("prob.weight"):
                v.mul_(peaked)
    return {k: v.clone() for k, v in sd.items()}


def cvp_cameras(nsrc, img_h, img_w, batch=1, rig=0):
    """CVP-MVSNet convention (CVP-MVSNet/models/net.py:106): full-image intrinsics
    ref_in [B,3,3], src_in [B,nsrc,3,3], extrinsics ref_ex [B,4,4], src_ex [B,nsrc,4,4]
    (float32), depth range [B] each."""
    K = feature_intrinsics(img_h, img_w).astype(np.float32)
    E = arc_extrinsics(nsrc + 1, rig).astype(np.float32)
    rep = lambda a: np.broadcast_to(a, (batch,) + a.shape).copy()
    return {"ref_in": rep(K), "src_in": rep(np.stack([K] * nsrc)), "ref_ex": rep(E[0]), "src_ex": rep(E[1:]),
            "depth_min": np.full((batch,), DTU_DEPTH_MIN, dtype=np.float32),
            "depth_max": np.full((batch,), DTU_DEPTH_MIN + 47 * 13.5, dtype=np.float32)}   # 48 planes, exact step


def cvp_random_state_dict(seed=0, peaked=60.0):
    """Seeded CVP-MVSNet state_dict (reference key names, 551,585 parameters): default conv
    init, BatchNorm affine/statistics perturbed away from identity, `prob0` scaled so the
    softmax over depth is peaked.  Deterministic for a given torch build."""
    import torch
    from .models.cvp_mvsnet import CVPMVSNet
    torch.manual_seed(seed)
    net = CVPMVSNet(nscale=2, nsrc=2)
    g = torch.Generator().manual_seed(seed + 1)
    sd = net.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if ".bn." in k or k.startswith("cost_reg_refine.conv5.1.") or k.startswith("cost_reg_refine.conv6.1."):
                if k.endswith("weight") or k.endswith("running_var"):
                    v.copy_(0.5 + torch.rand(v.shape, generator=g))
                elif k.endswith("bias") or k.endswith("running_mean"):
                    v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif k.endswith("prob0.weight"):
                v.mul_(peaked)
    return {k: v.clone() for k, v in sd.items()}


def plane_depth_from_cameras(Ks, Es, h, w, normal=(0.05, -0.08, 1.0), offset=DTU_TARGET_Z):
    """Depth maps [V,h,w] float32 of the world plane normal . X = offset for arbitrary pinhole cameras
    (K [3,3] at this resolution, E [4,4] world->camera)."""
    n = np.asarray(normal, dtype=np.float64)
    ys, xs = np.mgrid[0:h, 0:w]
    pix = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    out = []
    for K, E in zip(Ks, Es):
        K, E = np.asarray(K, dtype=np.float64), np.asarray(E, dtype=np.float64)
        rays = np.linalg.inv(K) @ pix
        R, t = E[:3, :3], E[:3, 3]
        out.append(((offset + n @ R.T @ t) / (n @ R.T @ rays)).reshape(h, w))
    return np.stack(out).astype(np.float32)


def plane_depth_maps(nviews, feat_h, feat_w, normal=(0.05, -0.08, 1.0), offset=DTU_TARGET_Z):
    """Exact per-view depth maps [V,h,w] (float32) of the world plane normal . X = offset seen from the
    arc cameras of proj_matrices() -- geometrically consistent across views by construction
    (inputs for the depth-map filter that follows the path).  Also returns K [3,3], E [V,4,4] float32."""
    K = feature_intrinsics(feat_h, feat_w)
    E = arc_extrinsics(nviews)
    n = np.asarray(normal, dtype=np.float64)
    ys, xs = np.mgrid[0:feat_h, 0:feat_w]
    rays = np.linalg.inv(K) @ np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])     # camera-frame rays, z = 1
    out = []
    for v in range(nviews):
        R, t = E[v, :3, :3], E[v, :3, 3]
        out.append(((offset + n @ R.T @ t) / (n @ R.T @ rays)).reshape(feat_h, feat_w))
    return np.stack(out).astype(np.float32), K.astype(np.float32), E.astype(np.float32)
