"""DTU evaluation samples exactly as the reference's loader hands them to the model
(MVSNet/datasets/dtu_yao_eval.py:9-108; file formats: SURVEY.md Appendix A).

One sample = the reference view + the first nviews-1 source views of pair.txt:
    imgs [V,3,1184,1600] float32 in [0,1] (1600x1200 JPEG, bottom 16 rows cropped)
    proj_matrices [V,4,4] float32: extrinsic with its top 3x4 replaced by K/4 @ E[:3,:4]
    depth_values [D] float32: arange(depth_min, interval*(D-0.5)+depth_min, interval)
    filename "<scan>/{}/<ref:08d>{}"
"""
import os

import numpy as np


def read_pair_file(path):
    """-> [(ref_view, [src views, best first]), ...]  (dtu_yao_eval.py:30-39, eval.py:82-91)."""
    with open(path) as f:
        tokens = f.read().split("\n")
    n = int(tokens[0])
    out = []
    for i in range(n):
        ref = int(tokens[1 + 2 * i].rstrip())
        src = [int(x) for x in tokens[2 + 2 * i].rstrip().split()[1::2]]
        out.append((ref, src))
    return out


def read_cam_file(path, interval_scale=1.0, intrinsics_div=4.0):
    """-> (intrinsics [3,3] with rows 0-1 divided by `intrinsics_div`, extrinsics [4,4],
    depth_min, depth_interval * interval_scale)  (dtu_yao_eval.py:46-58)."""
    with open(path) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    intrinsics[:2, :] /= intrinsics_div
    fields = lines[11].split()
    return intrinsics, extrinsics, float(fields[0]), float(fields[1]) * interval_scale


def read_image(path, expect_hw=(1200, 1600), crop_bottom=16):
    from PIL import Image
    img = np.array(Image.open(path), dtype=np.float32) / 255.0
    assert img.shape[:2] == tuple(expect_hw), f"{path}: {img.shape[:2]} != {tuple(expect_hw)}"
    return img[:-crop_bottom, :] if crop_bottom else img


class MVSDataset:
    """Same constructor and sample dict as the reference's class; usable with
    torch.utils.data.DataLoader (it only needs __len__ / __getitem__)."""

    def __init__(self, datapath, listfile, mode, nviews, ndepths=192, interval_scale=1.06, **kwargs):
        assert mode == "test"
        self.datapath, self.listfile, self.mode = datapath, listfile, mode
        self.nviews, self.ndepths, self.interval_scale = nviews, ndepths, interval_scale
        self.image_hw = kwargs.get("image_hw", (1200, 1600))
        self.crop_bottom = kwargs.get("crop_bottom", 16)
        with open(listfile) as f:
            scans = [ln.rstrip() for ln in f.readlines()]
        self.metas = [(scan, ref, src) for scan in scans
                      for ref, src in read_pair_file(os.path.join(datapath, scan, "pair.txt"))]

    def __len__(self):
        return len(self.metas)

    def __getitem__(self, idx):
        scan, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.nviews - 1]
        imgs, projs, depth_values = [], [], None
        for i, vid in enumerate(view_ids):
            imgs.append(read_image(os.path.join(self.datapath, scan, "images", f"{vid:0>8}.jpg"),
                                   self.image_hw, self.crop_bottom))
            K, E, dmin, dint = read_cam_file(os.path.join(self.datapath, scan, "cams", f"{vid:0>8}_cam.txt"),
                                             self.interval_scale)
            P = E.copy()
            P[:3, :4] = np.matmul(K, P[:3, :4])
            projs.append(P)
            if i == 0:
                depth_values = np.arange(dmin, dint * (self.ndepths - 0.5) + dmin, dint, dtype=np.float32)
        return {"imgs": np.stack(imgs).transpose([0, 3, 1, 2]),
                "proj_matrices": np.stack(projs),
                "depth_values": depth_values,
                "filename": scan + "/{}/" + f"{view_ids[0]:0>8}" + "{}"}
