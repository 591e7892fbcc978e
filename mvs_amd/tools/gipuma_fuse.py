#!/usr/bin/env python3
"""The reference's Gipuma post-processing (CasMVSNet/gipuma.py:160-213 `gipuma_filter`, which shells out
to the CUDA `fusibile` binary) with the fusion on the MI355X:

    python -m mvs_amd.tools.gipuma_fuse --outdir outputs --testlist lists/dtu/test.txt \\
        --prob_threshold 0.8 --disp_threshold 0.25 --num_consistent 3

per scan in {outdir}/{scan} (images/, cams/, depth_est/, confidence/ as the depth stage left them):
  1. probability filter: depth_est/{id}_prob_filtered.pfm (depth where confidence >= threshold, else 0);
  2. points_mvsnet/: cams/{image}.P, images/, 2333__{id}/disp.dmb + normals.dmb (the exchange files the
     reference writes for fusibile -- kept so that either fuser can be run on them);
  3. fusion of every reference view against all others (mvs_fusibile_fuse_f32) and
     points_mvsnet/consistencyCheck-hip/final3d_model.ply in fusibile's PLY layout (xyz float, rgb uchar).
"""
import argparse
import ctypes
import os
import shutil

import numpy as np
import torch

from .. import _lib
from ..datasets import read_pfm, save_pfm
from ..datasets import gipuma_io as gio


def probability_filter(dense_folder, prob_threshold):
    """gipuma.py:139-155"""
    for image_name in sorted(os.listdir(os.path.join(dense_folder, "images"))):
        prefix = os.path.splitext(image_name)[0]
        depth, _ = read_pfm(os.path.join(dense_folder, "depth_est", prefix + ".pfm"))
        prob, _ = read_pfm(os.path.join(dense_folder, "confidence", prefix + ".pfm"))
        depth = np.array(depth)
        depth[prob < prob_threshold] = 0
        save_pfm(os.path.join(dense_folder, "depth_est", prefix + "_prob_filtered.pfm"), depth)


def mvsnet_to_gipuma(dense_folder, point_folder):
    """gipuma.py:110-137"""
    for sub in ("", "cams", "images"):
        os.makedirs(os.path.join(point_folder, sub), exist_ok=True)
    names = sorted(os.listdir(os.path.join(dense_folder, "images")))
    for image_name in names:
        prefix = os.path.splitext(image_name)[0]
        gio.mvsnet_to_gipuma_cam(os.path.join(dense_folder, "cams", prefix + "_cam.txt"),
                                 os.path.join(point_folder, "cams", image_name + ".P"))
        shutil.copy(os.path.join(dense_folder, "images", image_name), os.path.join(point_folder, "images", image_name))
        sub = os.path.join(point_folder, "2333__" + prefix)
        os.makedirs(sub, exist_ok=True)
        depth, _ = read_pfm(os.path.join(dense_folder, "depth_est", prefix + "_prob_filtered.pfm"))
        gio.write_gipuma_dmb(os.path.join(sub, "disp.dmb"), depth)
        gio.write_gipuma_dmb(os.path.join(sub, "normals.dmb"), gio.fake_gipuma_normal(gio.read_gipuma_dmb(os.path.join(sub, "disp.dmb"))))
    return names


def camera_records(Ps, f):
    """[N,28] float32: P, inverse(P[:, :3]), P[:, 3], centre, f (cameraGeometryUtils.h:386-439)."""
    out = np.zeros((len(Ps), 28), dtype=np.float32)
    for i, P in enumerate(Ps):
        P = np.asarray(P, dtype=np.float32)
        Minv = np.linalg.inv(P[:, :3]).astype(np.float32)
        out[i, :12], out[i, 12:21], out[i, 21:24] = P.reshape(-1), Minv.reshape(-1), P[:, 3]
        out[i, 24:27] = -(Minv @ P[:, 3])
        out[i, 27] = f
    return out


def fuse_views(nd, colors, cams, disp_thresh, normal_thresh, num_consistent):
    """nd [N,H,W,4], colors [N,H,W,4] or None, cams [N,28] (device) -> list over reference views of
    (points [H,W,4], colors [H,W,4] or None) device tensors."""
    lib = _lib.load()
    N, H, W, _ = nd.shape
    res = []
    for ref in range(N):
        pts = torch.empty((H, W, 4), device=nd.device, dtype=torch.float32)
        nrm = torch.empty_like(pts)
        col = torch.empty_like(pts) if colors is not None else None
        _lib.check(lib.mvs_fusibile_fuse_f32(_lib.ptr(nd), _lib.ptr(colors), _lib.ptr(cams), N, H, W, ref,
                                             float(disp_thresh), float(normal_thresh), int(num_consistent),
                                             _lib.ptr(pts), _lib.ptr(nrm), _lib.ptr(col), _lib.stream()),
                   "mvs_fusibile_fuse_f32")
        res.append((pts, col))
    return res


def write_ply(path, xyz, rgb):
    """fusibile's binary PLY (displayUtils.h:80-136): float xyz + uchar red/green/blue."""
    n = xyz.shape[0]
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    bad = ~np.isfinite(xyz).all(1)
    xyz = np.where(bad[:, None], 0, xyz)
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["r"], rec["g"], rec["b"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                 "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode())
        rec.tofile(f)


def fuse_scan(dense_folder, disp_threshold, num_consistent, normal_thresh_deg=360.0, device="cuda:0"):
    from PIL import Image
    point_folder = os.path.join(dense_folder, "points_mvsnet")
    names = mvsnet_to_gipuma(dense_folder, point_folder)
    dev = torch.device(device)
    Ps, nds, cols = [], [], []
    for image_name in names:
        prefix = os.path.splitext(image_name)[0]
        with open(os.path.join(point_folder, "cams", image_name + ".P")) as f:
            Ps.append(np.array(f.read().split(), dtype=np.float32).reshape(3, 4))
        depth = gio.read_gipuma_dmb(os.path.join(point_folder, "2333__" + prefix, "disp.dmb"))
        normal = gio.read_gipuma_dmb(os.path.join(point_folder, "2333__" + prefix, "normals.dmb"))
        nds.append(np.ascontiguousarray(np.concatenate([normal, depth[..., None]], -1), dtype=np.float32))
        img = np.asarray(Image.open(os.path.join(point_folder, "images", image_name)).convert("RGB"), dtype=np.float32)
        if img.shape[:2] != depth.shape:   # depth maps are quarter size; fusibile reads images of the depth size
            img = np.asarray(Image.fromarray(img.astype(np.uint8)).resize((depth.shape[1], depth.shape[0]), Image.BILINEAR),
                             dtype=np.float32)
        cols.append(np.concatenate([img[..., ::-1], np.zeros_like(img[..., :1])], -1))   # OpenCV order b, g, r, alpha
    K0, _ = gio.read_camera_parameters(os.path.join(dense_folder, "cams", os.path.splitext(names[0])[0] + "_cam.txt"))
    cams = torch.from_numpy(camera_records(Ps, float(K0[0, 0]))).to(dev)
    nd = torch.from_numpy(np.ascontiguousarray(np.stack(nds))).to(dev)
    colors = torch.from_numpy(np.ascontiguousarray(np.stack(cols), dtype=np.float32)).to(dev)
    fused = fuse_views(nd, colors, cams, disp_threshold, np.float32(normal_thresh_deg * np.pi / 180.0), num_consistent)
    xyz, rgb = [], []
    for pts, col in fused:
        p = pts.reshape(-1, 4)[:, :3]
        keep = (p != 0).all(1)                                   # fusibile.cu:309
        xyz.append(p[keep].cpu().numpy())
        c = col.reshape(-1, 4)[keep][:, :3].cpu().numpy()
        rgb.append(np.stack([c[:, 2], c[:, 1], c[:, 0]], 1).astype(np.int32).astype(np.uint8))   # (int) texture4[2], [1], [0]
    out_dir = os.path.join(point_folder, "consistencyCheck-hip")
    os.makedirs(out_dir, exist_ok=True)
    ply = os.path.join(out_dir, "final3d_model.ply")
    write_ply(ply, np.concatenate(xyz), np.concatenate(rgb))
    return ply, sum(len(x) for x in xyz)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Gipuma-style depth-map fusion (CasMVSNet/gipuma.py gipuma_filter)")
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--testlist", required=True)
    ap.add_argument("--prob_threshold", type=float, default=0.8)
    ap.add_argument("--disp_threshold", type=float, default=0.25)
    ap.add_argument("--num_consistent", type=int, default=3)
    args = ap.parse_args(argv)
    with open(args.testlist) as f:
        scans = [ln.rstrip() for ln in f.readlines() if ln.strip()]
    for scan in scans:
        dense = os.path.join(args.outdir, scan)
        probability_filter(dense, args.prob_threshold)
        ply, n = fuse_scan(dense, args.disp_threshold, args.num_consistent)
        print(f"{scan}: {n} points -> {ply}")


if __name__ == "__main__":
    main()
