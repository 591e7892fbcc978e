#!/usr/bin/env python3
"""Filter + fusion stage of the reference's eval.py (MVSNet/eval.py:217-326 filter_depth) with the
geometric-consistency check on the MI355X (mvs_geo_consistency_f32):

    python -m mvs_amd.tools.fuse_depth --testpath DTU/ --testlist lists/dtu/test.txt --outdir outputs

Per scan it reads pair.txt, cams, images and the depth_est / confidence PFMs that
`mvs_amd.tools.eval_depth` (or the reference) wrote, and writes
{outdir}/{scan}/mask/{ref:08d}_{photo,geo,final}.png and {outdir}/mvsnet{scan_id:03d}_l3.ply
(binary little-endian PLY: x, y, z float + red, green, blue uchar -- the layout plyfile writes).
photo mask: confidence > 0.8; geo mask: >= 3 consistent source views (the reference's constants).
The sampling inside the check restates cv2.remap (OpenCV is not in this image: see
oracle/geo_filter.py for what that leaves unpinned).
"""
import argparse
import os

import numpy as np
import torch

from .. import ops
from ..datasets import read_cam_file, read_pair_file, read_pfm


def read_img01(path):
    from PIL import Image
    return np.array(Image.open(path), dtype=np.float32) / 255.0


def save_mask(path, mask):
    from PIL import Image
    assert mask.dtype == np.bool_
    Image.fromarray(mask.astype(np.uint8) * 255).save(path)


def write_ply(path, xyz, rgb):
    """Binary little-endian PLY, vertex element (x, y, z f4; red, green, blue u1) -- eval.py:305-325."""
    v = np.empty(len(xyz), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    v["x"], v["y"], v["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    v["red"], v["green"], v["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
              "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(v))
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(v.tobytes())


def read_ply(path):
    """Reader for the files write_ply makes (tests, quick looks)."""
    with open(path, "rb") as f:
        n = None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line == "end_header":
                break
        v = np.frombuffer(f.read(), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"),
                                           ("blue", "u1")], count=n)
    return np.stack([v["x"], v["y"], v["z"]], 1), np.stack([v["red"], v["green"], v["blue"]], 1)


def filter_depth(scan_folder, out_folder, plyfilename, device="cuda:0", conf_thresh=0.8, min_views=3):
    pairs = read_pair_file(os.path.join(scan_folder, "pair.txt"))
    cams, depths = {}, {}

    def cam(v):        # intrinsics at feature resolution (rows 0-1 / 4), extrinsics  (eval.py:49-59)
        if v not in cams:
            K, E, _, _ = read_cam_file(os.path.join(scan_folder, "cams", f"{v:0>8}_cam.txt"), 1.0, 4.0)
            cams[v] = (K, E)
        return cams[v]

    def depth(v):
        if v not in depths:
            depths[v] = np.ascontiguousarray(read_pfm(os.path.join(out_folder, "depth_est", f"{v:0>8}.pfm"))[0], dtype=np.float32)
        return depths[v]

    vertexs, vertex_colors = [], []
    os.makedirs(os.path.join(out_folder, "mask"), exist_ok=True)
    for ref_view, src_views in pairs:
        K_ref, E_ref = cam(ref_view)
        ref_img = read_img01(os.path.join(scan_folder, "images", f"{ref_view:0>8}.jpg"))
        ref_depth = depth(ref_view)
        confidence = read_pfm(os.path.join(out_folder, "confidence", f"{ref_view:0>8}.pfm"))[0]
        photo_mask = confidence > conf_thresh
        res = ops.geo_consistency(torch.from_numpy(ref_depth).to(device), K_ref, E_ref,
                                  torch.from_numpy(np.stack([depth(s) for s in src_views])).to(device),
                                  [cam(s)[0] for s in src_views], [cam(s)[1] for s in src_views], per_view=False)
        geo_mask = res["geo_mask_sum"].cpu().numpy() >= min_views
        depth_avg = res["depth_averaged"].cpu().numpy()
        final_mask = np.logical_and(photo_mask, geo_mask)
        for kind, m in (("photo", photo_mask), ("geo", geo_mask), ("final", final_mask)):
            save_mask(os.path.join(out_folder, "mask", f"{ref_view:0>8}_{kind}.png"), m)
        print("processing {}, ref-view{:0>2}, photo/geo/final-mask:{}/{}/{}".format(
            scan_folder, ref_view, photo_mask.mean(), geo_mask.mean(), final_mask.mean()))
        h, w = depth_avg.shape
        x, y = np.meshgrid(np.arange(0, w), np.arange(0, h))
        x, y, d = x[final_mask], y[final_mask], depth_avg[final_mask]
        color = ref_img[1:-16:4, 1::4, :][final_mask]          # hardcoded for DTU in the reference (eval.py:289)
        xyz_ref = np.matmul(np.linalg.inv(K_ref), np.vstack((x, y, np.ones_like(x))) * d)
        xyz_world = np.matmul(np.linalg.inv(E_ref), np.vstack((xyz_ref, np.ones_like(x))))[:3]
        vertexs.append(xyz_world.transpose((1, 0)))
        vertex_colors.append((color * 255).astype(np.uint8))
    write_ply(plyfilename, np.concatenate(vertexs, axis=0), np.concatenate(vertex_colors, axis=0))
    print("saving the final model to", plyfilename)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Filter depth maps and fuse them into a point cloud (MVSNet/eval.py filter_depth)")
    ap.add_argument("--testpath", required=True)
    ap.add_argument("--testlist", required=True)
    ap.add_argument("--outdir", default="./outputs")
    ap.add_argument("--min_views", type=int, default=3,
                    help="consistent source views required (3 hard-coded in eval.py:260; CasMVSNet/test.py --thres_view)")
    ap.add_argument("--conf", type=float, default=0.8,
                    help="photometric confidence threshold (0.8 hard-coded in eval.py:237; CasMVSNet/test.py --conf)")
    args = ap.parse_args(argv)
    with open(args.testlist) as f:
        scans = [ln.rstrip() for ln in f.readlines()]
    for scan in scans:
        scan_id = int(scan[4:])
        filter_depth(os.path.join(args.testpath, scan), os.path.join(args.outdir, scan),
                     os.path.join(args.outdir, "mvsnet{:0>3}_l3.ply".format(scan_id)), conf_thresh=args.conf,
                     min_views=args.min_views)


if __name__ == "__main__":
    main()
