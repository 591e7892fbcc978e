"""Test infrastructure: numpy restatement of the reference's depth-map fusion kernel
(fusibile/fusibile.cu:138-277 with its helpers :56-65, :126-132, config.h:33-35,160-188, and the camera
set-up of cameraGeometryUtils.h:386-439).  Vectorised over the pixels of one reference view, float32.

PARITY UNPINNED: fusibile needs CUDA and OpenCV (neither in the image) and the reference holds no
fixture for it; CUDA's linear texture filter (1.8 fixed-point weights, clamped addressing for
unnormalised coordinates) is restated from its documentation, as in mvs_amd/csrc/fusibile.hip.
Only tests/ and scripts/ may import this module."""
import numpy as np

f32 = np.float32


def camera_records(Ps, f):
    """Ps [N,3,4] float32 -> [N,28] float32: P (12), inverse(P[:, :3]) (9), P[:, 3] (3), centre (3), f.
    (cameraGeometryUtils.h:387 M_inv = P.colRange(0,3).inv(); C = -M_inv P[:,3]; cpc.cameras[i].f = K(0,0).)"""
    out = np.zeros((len(Ps), 28), dtype=f32)
    for i, P in enumerate(Ps):
        P = np.asarray(P, dtype=f32)
        Minv = np.linalg.inv(P[:, :3]).astype(f32)
        out[i, :12] = P.reshape(-1)
        out[i, 12:21] = Minv.reshape(-1)
        out[i, 21:24] = P[:, 3]
        out[i, 24:27] = -(Minv @ P[:, 3])
        out[i, 27] = f
    return out


def _tex_linear(img, x, y):
    """tex2D(img, x, y) with cudaFilterModeLinear, unnormalised coordinates, clamped addressing.
    img [H,W,C]; x, y arrays."""
    H, W = img.shape[:2]
    xb, yb = x - f32(0.5), y - f32(0.5)
    fx, fy = np.floor(xb), np.floor(yb)
    a = (np.floor((xb - fx) * f32(256) + f32(0.5)) * f32(1 / 256)).astype(f32)[..., None]
    b = (np.floor((yb - fy) * f32(256) + f32(0.5)) * f32(1 / 256)).astype(f32)[..., None]
    i0, i1 = np.clip(fx.astype(np.int64), 0, W - 1), np.clip(fx.astype(np.int64) + 1, 0, W - 1)
    j0, j1 = np.clip(fy.astype(np.int64), 0, H - 1), np.clip(fy.astype(np.int64) + 1, 0, H - 1)
    one = f32(1)
    return ((one - a) * (one - b) * img[j0, i0] + a * (one - b) * img[j0, i1] +
            (one - a) * b * img[j1, i0] + a * b * img[j1, i1]).astype(f32)


def _lift(cam, px, py, depth):
    Minv, Pc = cam[12:21].reshape(3, 3), cam[21:24]
    v = np.stack([depth * px - Pc[0], depth * py - Pc[1], depth - Pc[2]], -1).astype(f32)
    return (v @ Minv.T).astype(f32)


def fuse_view(nd, colors, cams, ref, disp_thresh, normal_thresh, num_consistent):
    """nd [N,H,W,4]; colors [N,H,W,4] or None; cams [N,28] -> (points [H,W,3], normals [H,W,3], colors [H,W,3] or None,
    count [H,W]); pixels without a fused point hold zeros."""
    N, H, W, _ = nd.shape
    ys, xs = np.mgrid[0:H, 0:W]
    px, py = xs.astype(f32), ys.astype(f32)
    rc = cams[ref]
    normal = nd[ref]
    X = _lift(rc, px, py, normal[..., 3])
    cX, cN = X.copy(), normal[..., :3].copy()
    cT = colors[ref][..., :3].copy() if colors is not None else None
    cnt = np.zeros((H, W), dtype=np.int32)
    with np.errstate(all="ignore"):
        for i in range(N):
            if i == ref:
                continue
            c = cams[i]
            P = c[:12].reshape(3, 4)
            t = (X @ P[:, :3].T + P[:, 3]).astype(f32)
            qx, qy, depth = t[..., 0] / t[..., 2], t[..., 1] / t[..., 2], t[..., 2]
            inside = (qx >= 0) & (qx < W) & (qy >= 0) & (qy < H)
            qxs, qys = np.where(inside, qx, 0).astype(f32), np.where(inside, qy, 0).astype(f32)
            s = _tex_linear(nd[i], qxs + f32(0.5), qys + f32(0.5))
            baseline = f32(np.sqrt(np.sum((rc[24:27] - c[24:27]) ** 2, dtype=f32)))
            ok = inside & (np.abs(rc[27] * baseline / depth - rc[27] * baseline / s[..., 3]) < disp_thresh)
            angle = np.arccos(np.sum(s[..., :3] * normal[..., :3], -1, dtype=f32))
            angle = np.where(np.isnan(angle), f32(0), angle)
            ok &= angle < normal_thresh
            tX = _lift(c, np.trunc(qxs), np.trunc(qys), s[..., 3])
            cX += np.where(ok[..., None], tX, 0)
            cN += np.where(ok[..., None], s[..., :3], 0)
            if cT is not None:
                cT += np.where(ok[..., None], _tex_linear(colors[i], qxs + f32(0.5), qys + f32(0.5))[..., :3], 0)
            cnt += ok
    k = (cnt.astype(f32) + f32(1))[..., None]
    keep = (cnt >= num_consistent)[..., None]
    pts = np.where(keep, cX / k, 0).astype(f32)
    nrm = np.where(keep, cN / k, 0).astype(f32)
    col = np.where(keep, cT / k, 0).astype(f32) if cT is not None else None
    return pts, nrm, col, cnt
