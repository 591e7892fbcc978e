"""On-disk contract vs golden vectors captured from the reference's
own loader and PFM code (tests/golden/make_golden_io.py)."""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from io_fixture import build_scan, build_train_set  # noqa: E402


@pytest.fixture(scope="module")
def scan(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("dtu"))
    return root, build_scan(root)


def test_pfm_bytes_and_roundtrip(tmp_path):
    from mvs_amd.datasets import read_pfm, save_pfm
    g = load_golden("g10_io")
    for name in ("grey", "color"):
        arr = g[f"pfm_{name}_array"]
        path = str(tmp_path / (name + ".pfm"))
        save_pfm(path, arr)
        assert open(path, "rb").read() == g[f"pfm_{name}_bytes"].tobytes()   # byte-identical file
        back, scale = read_pfm(path)
        assert scale == 1.0 and back.dtype.kind == "f" and np.array_equal(back, arr)
    with pytest.raises(Exception):
        save_pfm(str(tmp_path / "x.pfm"), np.zeros((2, 2), dtype=np.float64))
    # the eval driver's writer: rows already in file order (flipped on the GPU), the buffer written as it is
    from mvs_amd.datasets import save_pfm_rows_bottom_up
    grey = np.ascontiguousarray(g["pfm_grey_array"].reshape(g["pfm_grey_array"].shape[:2]))
    save_pfm_rows_bottom_up(str(tmp_path / "rows.pfm"), np.ascontiguousarray(grey[::-1]))
    assert (tmp_path / "rows.pfm").read_bytes() == g["pfm_grey_bytes"].tobytes()
    with pytest.raises(Exception):
        save_pfm_rows_bottom_up(str(tmp_path / "rows2.pfm"), grey[::-1])      # not contiguous
    bad = tmp_path / "bad.pfm"
    bad.write_bytes(b"P6\n1 1\n255\n\0\0\0")
    with pytest.raises(Exception):
        read_pfm(str(bad))


def test_cam_file_and_pairs(scan):
    from mvs_amd.datasets import read_cam_file, read_pair_file
    root, _ = scan
    g = load_golden("g10_io")
    K, E, dmin, dint = read_cam_file(os.path.join(root, "scan1/cams/00000002_cam.txt"), interval_scale=1.06)
    assert np.array_equal(K, g["cam_K"]) and np.array_equal(E, g["cam_E"])
    assert dmin == float(g["cam_dmin"]) and dint == float(g["cam_dint"])
    pairs = read_pair_file(os.path.join(root, "scan1/pair.txt"))
    assert [[r] + s for r, s in pairs] == g["metas"].tolist()


def test_eval_samples_match_reference_loader(scan):
    from mvs_amd.datasets import MVSDataset, find_dataset_def
    from mvs_amd.datasets.dtu_eval import read_image
    root, listfile = scan
    g = load_golden("g10_io")
    ds = find_dataset_def("dtu_yao_eval")(root, listfile, "test", 3, 192, 1.06)
    assert isinstance(ds, MVSDataset) and len(ds) == int(g["n_samples"])
    for i in (0, len(ds) - 1):
        s = ds[i]
        assert np.array_equal(s["proj_matrices"], g[f"s{i}_proj"])           # bit-identical fp32
        assert np.array_equal(s["depth_values"], g[f"s{i}_depth_values"])
        assert s["filename"] == str(g[f"s{i}_filename"])
        assert list(s["imgs"].shape) == g[f"s{i}_imgs_shape"].tolist() and s["imgs"].dtype == np.float32
        assert np.array_equal(s["imgs"][:, :, ::97, ::131], g[f"s{i}_imgs_probe"])
    with pytest.raises(AssertionError):
        read_image(os.path.join(root, "scan1/images/00000000.jpg"), expect_hw=(100, 100))


def test_eval_tool_argument_surface():
    from mvs_amd.tools import eval_depth
    with pytest.raises(SystemExit):
        eval_depth.main(["--help"])


@pytest.mark.gpu
def test_eval_tool_writes_reference_output_tree(scan, tmp_path):
    """mvs_amd.tools.eval_depth (eval.py save_depth): checkpoint in the reference's format
    (DataParallel 'module.' keys) -> {outdir}/{scan}/depth_est|confidence/{ref:08d}.pfm whose
    contents are the model's outputs for that sample."""
    import torch
    from mvs_amd import synth
    from mvs_amd.datasets import MVSDataset, read_pfm
    from mvs_amd.models import MVSNet
    from mvs_amd.tools import eval_depth
    root, listfile = scan
    sd = synth.random_state_dict(0)
    ckpt = str(tmp_path / "model_000000.ckpt")
    torch.save({"epoch": 0, "model": {"module." + k: v for k, v in sd.items()}}, ckpt)
    outdir = str(tmp_path / "out")
    eval_depth.main(["--testpath", root, "--testlist", listfile, "--loadckpt", ckpt, "--outdir", outdir,
                     "--nviews", "3", "--numdepth", "48", "--num_workers", "0"])
    ds = MVSDataset(root, listfile, "test", 3, 48, 1.06)
    model = MVSNet(refine=False)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    for i in range(len(ds)):
        s = ds[i]
        with torch.no_grad():
            out = model(*(torch.from_numpy(s[k])[None].cuda() for k in ("imgs", "proj_matrices", "depth_values")))
        for kind, key in (("depth_est", "depth"), ("confidence", "photometric_confidence")):
            arr, scale = read_pfm(os.path.join(outdir, s["filename"].format(kind, ".pfm")))
            assert scale == 1.0 and arr.shape == (296, 400)
            assert np.array_equal(arr, out[key][0].cpu().numpy())


def test_train_samples_match_reference_loader(tmp_path):
    """Yao's training layout through the mirror of MVSNet/datasets/dtu_yao.py: sample list
    (scan x reference view x 7 lights), projection matrices, depth hypotheses, ground-truth
    depth, mask and images -- bit-identical to what the reference's loader returned."""
    from mvs_amd.datasets import find_dataset_def
    root = str(tmp_path)
    listfile = build_train_set(root)
    g = load_golden("g10_io")
    ds = find_dataset_def("dtu_yao")(root, listfile, "train", 3, 192, 1.06)
    assert len(ds) == int(g["t_n_samples"]) == 21
    assert [[m[1], m[2]] + list(m[3]) for m in ds.metas] == g["t_metas"].tolist()
    for i in (0, 9, len(ds) - 1):
        s = ds[i]
        assert set(s) == {"imgs", "proj_matrices", "depth", "depth_values", "mask"}
        assert np.array_equal(s["proj_matrices"], g[f"t{i}_proj"])
        assert np.array_equal(s["depth_values"], g[f"t{i}_depth_values"]) and s["depth_values"].dtype == np.float32
        assert np.array_equal(s["depth"], g[f"t{i}_depth"]) and s["depth"].dtype == np.float32
        assert np.array_equal(s["mask"], g[f"t{i}_mask"])
        assert list(s["imgs"].shape) == g[f"t{i}_imgs_shape"].tolist()
        assert np.array_equal(s["imgs"][:, :, ::37, ::41], g[f"t{i}_imgs_probe"])
    # the loss keeps mask > 0.5 (train.py:224 / mvsnet.py:201-203): the fixture has both kinds of pixels
    assert 0 < float((ds[0]["mask"] > 0.5).mean()) < 1


def test_ply_writer_layout_and_roundtrip(tmp_path):
    """The fused point cloud file: binary little-endian PLY with the vertex layout plyfile writes in
    the reference (eval.py:305-325): float x, y, z + uchar red, green, blue, 15 bytes per vertex."""
    from mvs_amd.tools.fuse_depth import read_ply, write_ply
    rng = np.random.default_rng(1)
    xyz = rng.standard_normal((7, 3)) * 100
    rgb = rng.integers(0, 256, (7, 3), dtype=np.uint8)
    path = str(tmp_path / "cloud.ply")
    write_ply(path, xyz, rgb)
    raw = open(path, "rb").read()
    head = (b"ply\nformat binary_little_endian 1.0\nelement vertex 7\nproperty float x\nproperty float y\n"
            b"property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
    assert raw.startswith(head) and len(raw) == len(head) + 7 * 15
    back_xyz, back_rgb = read_ply(path)
    assert np.array_equal(back_xyz, xyz.astype(np.float32)) and np.array_equal(back_rgb, rgb)


@pytest.mark.gpu
def test_fuse_tool_masks_and_point_cloud(scan, tmp_path):
    """mvs_amd.tools.fuse_depth (eval.py filter_depth) on the fixture scan with exact depth maps of a
    tilted plane: masks on disk, every fused point on the plane, colours from the reference images,
    a low-confidence patch and an inconsistent patch left out."""
    from mvs_amd import synth
    from mvs_amd.datasets import read_cam_file, save_pfm
    from mvs_amd.tools import fuse_depth
    from PIL import Image
    root, listfile = scan
    outdir = str(tmp_path / "out")
    normal, offset = (0.05, -0.08, 1.0), 80.0
    cams = [read_cam_file(os.path.join(root, "scan1", "cams", f"{v:0>8}_cam.txt"), 1.0, 4.0)[:2] for v in range(3)]
    depths = synth.plane_depth_from_cameras([c[0] for c in cams], [c[1] for c in cams], 296, 400, normal, offset)
    depths[1, 100:140, 150:200] *= 1.08                       # view 1 disagrees here
    conf = np.full((296, 400), 0.9, dtype=np.float32)
    conf[200:220, 50:90] = 0.5                                # photometric mask off
    for v in range(3):
        for kind, arr in (("depth_est", depths[v]), ("confidence", conf)):
            os.makedirs(os.path.join(outdir, "scan1", kind), exist_ok=True)
            save_pfm(os.path.join(outdir, "scan1", kind, f"{v:0>8}.pfm"), arr)
    fuse_depth.main(["--testpath", root, "--testlist", listfile, "--outdir", outdir, "--min_views", "2"])
    xyz, rgb = fuse_depth.read_ply(os.path.join(outdir, "mvsnet001_l3.ply"))
    total = 0
    for v in range(3):
        masks = {k: np.array(Image.open(os.path.join(outdir, "scan1", "mask", f"{v:0>8}_{k}.png"))) > 0
                 for k in ("photo", "geo", "final")}
        assert np.array_equal(masks["photo"], conf > 0.8)
        assert np.array_equal(masks["final"], masks["photo"] & masks["geo"])
        assert not masks["photo"][205, 60] and masks["geo"].mean() > 0.3
        total += int(masks["final"].sum())
    assert len(xyz) == total and total > 10000
    off_plane = np.abs(xyz.astype(np.float64) @ np.array(normal) - offset)     # mm
    # on the plane, except where a sample straddles the edge of view 1's wrong patch (the 1 % test lets
    # a blend of right and wrong depths through: the algorithm's behaviour, not the kernel's)
    assert np.percentile(off_plane, 99) < 0.05 and off_plane.max() < 0.01 * 700
    m0 = np.array(Image.open(os.path.join(outdir, "scan1", "mask", "00000000_final.png"))) > 0
    img0 = np.array(Image.open(os.path.join(root, "scan1", "images", "00000000.jpg")), dtype=np.float32) / 255.0
    assert np.array_equal(rgb[:int(m0.sum())], (img0[1:-16:4, 1::4, :][m0] * 255).astype(np.uint8))
    g1 = np.array(Image.open(os.path.join(outdir, "scan1", "mask", "00000001_geo.png"))) > 0
    assert not g1[110:130, 160:190].any()                     # view 1's wrong depths pass nowhere


@pytest.mark.gpu
def test_device_pipeline_tensors_bit_equal_to_reference_loader(scan):
    """uint8 upload + mvs_images_u8_to_planar_f32 + mvs_proj_matrices_f32 give the
    tensors the reference's loader gave (g10_io.npz), and every sample equals the host loader's."""
    import torch
    from mvs_amd.datasets import MVSDataset
    from mvs_amd.datasets.device_pipeline import DeviceScanPipeline
    root, listfile = scan
    g = load_golden("g10_io")
    ds = MVSDataset(root, listfile, "test", 3, 192, 1.06)
    pipe = DeviceScanPipeline(root, listfile, 3, 192, 1.06)
    samples = list(pipe)
    torch.cuda.synchronize()
    assert len(samples) == len(ds) == int(g["n_samples"]) and pipe.stats["decoded"] == 3   # each JPEG once
    for i, s in enumerate(samples):
        host = ds[i]
        assert np.array_equal(s["imgs"][0].cpu().numpy(), host["imgs"])                     # bit-identical pixels
        assert np.array_equal(s["proj_matrices"][0].cpu().numpy(), host["proj_matrices"])
        assert np.array_equal(s["depth_values"][0].cpu().numpy(), host["depth_values"])
        assert s["filename"][0] == host["filename"]
    for i in (0, len(ds) - 1):
        assert np.array_equal(samples[i]["proj_matrices"][0].cpu().numpy(), g[f"s{i}_proj"])
        assert np.array_equal(samples[i]["imgs"][0].cpu().numpy()[:, :, ::97, ::131], g[f"s{i}_imgs_probe"])


@pytest.mark.gpu
def test_images_u8_kernel_odd_sizes(scan):
    """widths that are not multiples of 4, crops on both axes"""
    import torch
    from mvs_amd import ops
    rng = np.random.default_rng(3)
    for (n, hs, ws, h, w) in ((2, 9, 13, 7, 13), (1, 5, 16, 5, 10), (3, 4, 7, 3, 5)):
        a = rng.integers(0, 256, (n, hs, ws, 3), dtype=np.uint8)
        got = ops.images_u8_to_planar(torch.from_numpy(a).cuda(), h, w).cpu().numpy()
        want = (a.astype(np.float32) / 255.)[:, :h, :w].transpose(0, 3, 1, 2)
        assert np.array_equal(got, want)


@pytest.mark.gpu
def test_eval_tool_device_pipeline_writes_the_same_files(scan, tmp_path):
    import torch
    from mvs_amd import synth
    from mvs_amd.tools import eval_depth
    root, listfile = scan
    ckpt = str(tmp_path / "m.ckpt")
    torch.save({"model": synth.random_state_dict(0)}, ckpt)
    common = ["--testpath", root, "--testlist", listfile, "--loadckpt", ckpt, "--nviews", "3", "--numdepth", "48", "--quiet"]
    eval_depth.main(common + ["--outdir", str(tmp_path / "a"), "--num_workers", "0"])
    eval_depth.main(common + ["--outdir", str(tmp_path / "b"), "--device_pipeline"])
    for kind in ("depth_est", "confidence"):
        for i in range(3):
            fa = tmp_path / "a" / "scan1" / kind / f"{i:08d}.pfm"
            fb = tmp_path / "b" / "scan1" / kind / f"{i:08d}.pfm"
            assert fa.read_bytes() == fb.read_bytes()
