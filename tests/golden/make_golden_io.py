"""Golden vectors for the on-disk contract (PFM, cam files, pair.txt, DTU eval sample).

Run in the build container (needs /root/reference):   python tests/golden/make_golden_io.py
Builds a small synthetic scan (text files from tests/golden/io_fixture.py, 1600x1200 JPEGs) and a small
training set (640x512 PNGs, depth PFMs, masks), runs the reference's own loaders
(MVSNet/datasets/dtu_yao_eval.py, dtu_yao.py) and PFM code
(MVSNet/datasets/data_io.py) on it and stores what they return.  Only data is stored.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from io_fixture import build_scan, build_train_set  # noqa: E402

REF = "/root/reference/MVSNet"


def main():
    for k in [k for k in sys.modules if k == "datasets" or k.startswith("datasets.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    from datasets import data_io as ref_io
    from datasets.dtu_yao_eval import MVSDataset as RefDataset
    from datasets.dtu_yao import MVSDataset as RefTrainDataset
    sys.path.pop(0)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        listfile = build_scan(tmp)
        ds = RefDataset(tmp, listfile, "test", 3, 192, 1.06)
        out["n_samples"] = np.int64(len(ds))
        out["metas"] = np.array([[m[1]] + list(m[2]) for m in ds.metas], dtype=np.int64)
        for i in (0, len(ds) - 1):
            s = ds[i]
            out[f"s{i}_proj"] = s["proj_matrices"]
            out[f"s{i}_depth_values"] = s["depth_values"]
            out[f"s{i}_filename"] = np.array(s["filename"])
            out[f"s{i}_imgs_shape"] = np.array(s["imgs"].shape)
            out[f"s{i}_imgs_probe"] = s["imgs"][:, :, ::97, ::131].copy()
        K, E, dmin, dint = ds.read_cam_file(os.path.join(tmp, "scan1/cams/00000002_cam.txt"))
        out.update(cam_K=K, cam_E=E, cam_dmin=np.float64(dmin), cam_dint=np.float64(dint))
        rng = np.random.default_rng(5)
        for name, arr in (("grey", rng.standard_normal((5, 7)).astype(np.float32) * 400),
                          ("color", rng.random((4, 6, 3), dtype=np.float32))):
            path = os.path.join(tmp, name + ".pfm")
            ref_io.save_pfm(path, arr)
            out[f"pfm_{name}_array"] = arr
            out[f"pfm_{name}_bytes"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
            back, scale = ref_io.read_pfm(path)
            assert np.array_equal(back, arr) and scale == 1.0
    with tempfile.TemporaryDirectory() as tmp:   # training layout through the reference's dtu_yao.py
        listfile = build_train_set(tmp)
        ds = RefTrainDataset(tmp, listfile, "train", 3, 192, 1.06)
        out["t_n_samples"] = np.int64(len(ds))
        out["t_metas"] = np.array([[m[1], m[2]] + list(m[3]) for m in ds.metas], dtype=np.int64)
        for i in (0, 9, len(ds) - 1):
            s = ds[i]
            out[f"t{i}_proj"] = s["proj_matrices"]
            out[f"t{i}_depth_values"] = s["depth_values"]
            out[f"t{i}_depth"] = s["depth"]
            out[f"t{i}_mask"] = s["mask"]
            out[f"t{i}_imgs_shape"] = np.array(s["imgs"].shape)
            out[f"t{i}_imgs_probe"] = s["imgs"][:, :, ::37, ::41].copy()
    np.savez_compressed(os.path.join(HERE, "g10_io.npz"), **out)
    print("wrote g10_io.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
