"""DTU training samples (Yao's preprocessed set) as the reference's training loader hands them
to the model and the loss (MVSNet/datasets/dtu_yao.py:10-122; consumed by train.py:204-230):

    <datapath>/Cameras/pair.txt, Cameras/<vid:08d>_cam.txt        (intrinsics already at 1/4 scale: not divided)
    <datapath>/Rectified/<scan>_train/rect_<vid+1:03d>_<light>_r5000.png     640x512 RGB
    <datapath>/Depths/<scan>_train/depth_map_<vid:04d>.pfm, depth_visual_<vid:04d>.png   160x128

One sample per (scan, reference view, light condition 0..6):
    imgs [V,3,H,W] float32 in [0,1]; proj_matrices [V,4,4] (E with its top 3x4 = K @ E[:3,:4]);
    depth [h,w] (reference view's ground truth); depth_values [D] = arange(min, interval*D + min, interval);
    mask [h,w] float32 in [0,1] (the loss keeps mask > 0.5).
"""
import os

import numpy as np

from .data_io import read_pfm
from .dtu_eval import read_cam_file, read_pair_file

N_LIGHTS = 7


def read_png01(path):
    from PIL import Image
    return np.array(Image.open(path), dtype=np.float32) / 255.0


class MVSDataset:
    """Same constructor and sample dict as the reference's class (dtu_yao.py)."""

    def __init__(self, datapath, listfile, mode, nviews, ndepths=192, interval_scale=1.06, **kwargs):
        assert mode in ("train", "val", "test")
        self.datapath, self.listfile, self.mode = datapath, listfile, mode
        self.nviews, self.ndepths, self.interval_scale = nviews, ndepths, interval_scale
        with open(listfile) as f:
            scans = [ln.rstrip() for ln in f.readlines()]
        pairs = read_pair_file(os.path.join(datapath, "Cameras", "pair.txt"))
        self.metas = [(scan, light, ref, src) for scan in scans for ref, src in pairs for light in range(N_LIGHTS)]

    def __len__(self):
        return len(self.metas)

    def __getitem__(self, idx):
        scan, light, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.nviews - 1]
        imgs, projs = [], []
        depth = mask = depth_values = None
        for i, vid in enumerate(view_ids):
            imgs.append(read_png01(os.path.join(self.datapath, "Rectified", f"{scan}_train",
                                                f"rect_{vid + 1:0>3}_{light}_r5000.png")))
            K, E, dmin, dint = read_cam_file(os.path.join(self.datapath, "Cameras", f"{vid:0>8}_cam.txt"),
                                             self.interval_scale, intrinsics_div=1.0)
            P = E.copy()
            P[:3, :4] = np.matmul(K, P[:3, :4])
            projs.append(P)
            if i == 0:
                depth_values = np.arange(dmin, dint * self.ndepths + dmin, dint, dtype=np.float32)
                mask = read_png01(os.path.join(self.datapath, "Depths", f"{scan}_train", f"depth_visual_{vid:0>4}.png"))
                depth = np.array(read_pfm(os.path.join(self.datapath, "Depths", f"{scan}_train",
                                                       f"depth_map_{vid:0>4}.pfm"))[0], dtype=np.float32)
        return {"imgs": np.stack(imgs).transpose([0, 3, 1, 2]), "proj_matrices": np.stack(projs),
                "depth": depth, "depth_values": depth_values, "mask": mask}
