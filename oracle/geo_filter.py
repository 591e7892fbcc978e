"""CPU restatement (numpy) of the reference's geometric-consistency filter -- TEST INFRASTRUCTURE.

Follows MVSNet/eval.py:136-187 (reproject_with_depth), :190-214 (check_geometric_consistency)
and the per-reference-view fusion of filter_depth (:255-262): same numpy operations in the same
order and dtypes (float32 matrices and maps, float64 per-pixel arithmetic), except cv2.remap.

PARITY UNPINNED for the sampling step: OpenCV is absent from this image, so `remap_linear` restates
cv2.remap(..., INTER_LINEAR) from OpenCV's published algorithm (imgproc/imgwarp.cpp: coordinates
rounded to 1/32 pixel with cvRound, a 32x32 table of float bilinear weights, constant border 0)
and could not be checked against the library.  Everything else is plain numpy as in the reference.
Only tests/ and scripts/ import this file.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def remap_linear(src, map_x, map_y):
    """cv2.remap(src, map_x, map_y, cv2.INTER_LINEAR) for a float32 image, BORDER_CONSTANT 0."""
    h, w = src.shape
    with np.errstate(invalid="ignore", over="ignore"):
        fx = map_x.astype(np.float64) * INTER_TAB_SIZE
        fy = map_y.astype(np.float64) * INTER_TAB_SIZE
        bad = ~(np.isfinite(fx) & np.isfinite(fy)) | (np.abs(fx) > 2 ** 30) | (np.abs(fy) > 2 ** 30)
        sx = np.where(bad, -(1 << 20), np.rint(np.where(bad, 0, fx))).astype(np.int64)   # cvRound: half to even
        sy = np.where(bad, -(1 << 20), np.rint(np.where(bad, 0, fy))).astype(np.int64)
    x0, y0 = sx >> INTER_BITS, sy >> INTER_BITS
    ax = (sx & (INTER_TAB_SIZE - 1)).astype(np.float32) / np.float32(INTER_TAB_SIZE)
    ay = (sy & (INTER_TAB_SIZE - 1)).astype(np.float32) / np.float32(INTER_TAB_SIZE)
    one = np.float32(1.0)
    w00, w01 = (one - ay) * (one - ax), (one - ay) * ax
    w10, w11 = ay * (one - ax), ay * ax

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        return np.where(ok, src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], np.float32(0))

    out = tap(y0, x0) * w00 + tap(y0, x0 + 1) * w01 + tap(y0 + 1, x0) * w10 + tap(y0 + 1, x0 + 1) * w11
    return out.astype(np.float32)


def reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """eval.py:136-187."""
    h, w = depth_ref.shape
    x_ref, y_ref = np.meshgrid(np.arange(0, w), np.arange(0, h))
    x_ref, y_ref = x_ref.reshape([-1]), y_ref.reshape([-1])
    with np.errstate(divide="ignore", invalid="ignore"):
        xyz_ref = np.matmul(np.linalg.inv(K_ref), np.vstack((x_ref, y_ref, np.ones_like(x_ref))) * depth_ref.reshape([-1]))
        xyz_src = np.matmul(np.matmul(E_src, np.linalg.inv(E_ref)), np.vstack((xyz_ref, np.ones_like(x_ref))))[:3]
        K_xyz_src = np.matmul(K_src, xyz_src)
        xy_src = K_xyz_src[:2] / K_xyz_src[2:3]
        x_src = xy_src[0].reshape([h, w]).astype(np.float32)
        y_src = xy_src[1].reshape([h, w]).astype(np.float32)
        sampled = remap_linear(depth_src, x_src, y_src)
        xyz_src = np.matmul(np.linalg.inv(K_src), np.vstack((xy_src, np.ones_like(x_ref))) * sampled.reshape([-1]))
        xyz_rep = np.matmul(np.matmul(E_ref, np.linalg.inv(E_src)), np.vstack((xyz_src, np.ones_like(x_ref))))[:3]
        depth_rep = xyz_rep[2].reshape([h, w]).astype(np.float32)
        K_xyz_rep = np.matmul(K_ref, xyz_rep)
        xy_rep = K_xyz_rep[:2] / K_xyz_rep[2:3]
    return (depth_rep, xy_rep[0].reshape([h, w]).astype(np.float32), xy_rep[1].reshape([h, w]).astype(np.float32),
            x_src, y_src)


def check_geometric_consistency(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """eval.py:190-214 -> (mask, depth_reprojected with 0 outside the mask, x_src, y_src)."""
    h, w = depth_ref.shape
    x_ref, y_ref = np.meshgrid(np.arange(0, w), np.arange(0, h))
    depth_rep, x_rep, y_rep, x_src, y_src = reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src)
    with np.errstate(divide="ignore", invalid="ignore"):
        dist = np.sqrt((x_rep - x_ref) ** 2 + (y_rep - y_ref) ** 2)
        rel = np.abs(depth_rep - depth_ref) / depth_ref
        mask = np.logical_and(dist < 1, rel < 0.01)
    depth_rep[~mask] = 0
    return mask, depth_rep, x_src, y_src


def fuse_reference_view(depth_ref, K_ref, E_ref, src_depths, src_Ks, src_Es):
    """The per-reference-view part of filter_depth (eval.py:239-262): geo_mask_sum [H,W] int32,
    depth_est_averaged [H,W] float64, and the per-view masks / reprojected depths."""
    geo_sum, masks, deps = 0, [], []
    for d, K, E in zip(src_depths, src_Ks, src_Es):
        m, dr, _, _ = check_geometric_consistency(depth_ref, K_ref, E_ref, d, K, E)
        geo_sum = geo_sum + m.astype(np.int32)
        masks.append(m)
        deps.append(dr)
    averaged = (sum(deps) + depth_ref) / (geo_sum + 1)
    return geo_sum, averaged, np.stack(masks), np.stack(deps)
