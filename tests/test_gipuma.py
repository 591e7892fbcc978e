"""Gipuma exchange formats against bytes written by the reference's own functions
(tests/golden/make_golden_gipuma.py), and the fusion kernel against its numpy restatement
(oracle/fusibile.py; parity of the fusion itself is unpinned, see its header)."""
import os

import numpy as np
import pytest

from conftest import load_golden


def test_dmb_and_cam_files_byte_identical_to_reference(tmp_path):
    from mvs_amd.datasets import gipuma_io as gio
    g = load_golden("g15_gipuma")
    p = str(tmp_path / "disp.dmb")
    gio.write_gipuma_dmb(p, g["depth"])
    assert open(p, "rb").read() == g["depth_dmb"].tobytes()
    assert np.array_equal(gio.read_gipuma_dmb(p), g["depth_back"])
    n = str(tmp_path / "normals.dmb")
    gio.write_gipuma_dmb(n, gio.fake_gipuma_normal(gio.read_gipuma_dmb(p)))
    assert open(n, "rb").read() == g["normal_dmb"].tobytes()
    assert np.array_equal(gio.read_gipuma_dmb(n), g["normal_back"])
    cam_in = str(tmp_path / "c_cam.txt")
    open(cam_in, "wb").write(g["cam_txt"].tobytes())
    gio.mvsnet_to_gipuma_cam(cam_in, str(tmp_path / "c.P"))
    assert open(str(tmp_path / "c.P"), "rb").read() == g["cam_P"].tobytes()


def _scene(N=4, H=40, W=56, seed=0):
    """Cameras on an arc looking at a fronto-parallel plane with a bump; depth maps rendered analytically."""
    from mvs_amd import synth
    rng = np.random.default_rng(seed)
    K = synth.feature_intrinsics(H, W)
    E = synth.arc_extrinsics(N)
    Ps = np.stack([(K @ E[i][:3]).astype(np.float32) for i in range(N)])
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    nd = np.zeros((N, H, W, 4), dtype=np.float32)
    for i in range(N):
        R, t = E[i][:3, :3], E[i][:3, 3]
        rays = np.linalg.inv(K) @ np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])
        # plane z_world = 680 in world coordinates: X = R^T (d * ray - t); solve for d per pixel
        A = (R.T @ rays)[2]
        b = (R.T @ t)[2]
        d = (680.0 + b) / A
        depth = d.reshape(H, W).astype(np.float32)
        if i == 2:
            depth[5:12, 8:20] *= 1.05            # an inconsistent patch in one view
        nd[i, ..., 3] = depth
        nd[i, ..., :3] = np.float32(1 / 1.732050808)
    nd[1, 0:3, :, :] = 0                           # masked depth (probability filter) in another
    colors = rng.integers(0, 255, (N, H, W, 4)).astype(np.float32)
    return Ps, nd, colors, float(K[0, 0])


@pytest.mark.gpu
def test_fusibile_kernel_vs_numpy_restatement():
    import torch
    from mvs_amd.tools.gipuma_fuse import camera_records, fuse_views
    from oracle import fusibile as orc
    Ps, nd, colors, f = _scene()
    cams = camera_records(Ps, f)
    assert np.array_equal(cams, orc.camera_records(Ps, f))
    dev = torch.device("cuda:0")
    got = fuse_views(torch.from_numpy(nd).to(dev), torch.from_numpy(colors).to(dev), torch.from_numpy(cams).to(dev),
                     0.25, np.float32(2 * np.pi), 2)
    n_points = 0
    for ref, (pts, col) in enumerate(got):
        want_p, _, want_c, cnt = orc.fuse_view(nd, colors, cams, ref, np.float32(0.25), np.float32(2 * np.pi), 2)
        p = pts.cpu().numpy()[..., :3]
        # threshold flips (|disparity difference| within rounding of 0.25) change a pixel's view count
        same = np.isclose(p, want_p, rtol=0, atol=2e-2).all(-1)
        assert same.mean() > 0.995, (ref, same.mean())
        kept = (want_p != 0).all(-1) & same
        assert np.abs(p[kept] - want_p[kept]).max() < 2e-2
        # random 0..255 colours: one quantum (1/256) of a bilinear weight, flipped by the last bit of the
        # projected coordinate, moves a sample by up to ~1 grey level
        dc = np.abs(col.cpu().numpy()[..., :3][kept] - want_c[kept])
        assert dc.max() < 1.1 and np.median(dc) < 1e-3
        n_points += int(kept.sum())
    assert n_points > 1000   # the plane fuses; the bumped patch and the masked rows do not agree everywhere


@pytest.mark.gpu
def test_gipuma_fuse_tool_end_to_end(tmp_path):
    """depth stage outputs -> probability filter -> exchange files -> fused PLY with fusibile's header."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from io_fixture import cam_text
    from PIL import Image
    from mvs_amd.datasets import save_pfm
    from mvs_amd.tools import gipuma_fuse
    scan = tmp_path / "out" / "scan1"
    for sub in ("images", "cams", "depth_est", "confidence"):
        os.makedirs(scan / sub)
    H, W = 32, 48
    rng = np.random.default_rng(1)
    for v in range(3):
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(scan / "images" / f"{v:08d}.jpg")
        open(scan / "cams" / f"{v:08d}_cam.txt", "w").write(cam_text(v))
        save_pfm(str(scan / "depth_est" / f"{v:08d}.pfm"), np.full((H, W), 600.0 + v, dtype=np.float32))
        conf = np.full((H, W), 0.9, dtype=np.float32)
        conf[:4] = 0.1
        save_pfm(str(scan / "confidence" / f"{v:08d}.pfm"), conf)
    lst = tmp_path / "list.txt"
    lst.write_text("scan1\n")
    gipuma_fuse.main(["--outdir", str(tmp_path / "out"), "--testlist", str(lst), "--num_consistent", "1",
                      "--disp_threshold", "50"])
    ply = scan / "points_mvsnet" / "consistencyCheck-hip" / "final3d_model.ply"
    head = open(ply, "rb").read(200).split(b"end_header\n")[0].decode()
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex ") and "property uchar blue" in head
    n = int(head.split("element vertex ")[1].split("\n")[0])
    assert 0 < n <= 3 * H * W
    assert os.path.getsize(ply) == len(head) + len("end_header\n") + n * 15
    assert (scan / "points_mvsnet" / "2333__00000001" / "normals.dmb").exists()
