"""A small synthetic DTU-style scan directory (3 viewpoints), written the same way by the
golden generator and by the tests: cam files, pair.txt, list file, 1600x1200 JPEGs."""
import os

import numpy as np


def cam_text(vid):
    a = 0.05 * vid
    c, s = np.cos(a), np.sin(a)
    E = np.array([[c, 0, s, -120.0 - 7.5 * vid], [0, 1, 0, 40.25 + vid], [-s, 0, c, 600.5 - 3 * vid], [0, 0, 0, 1]])
    K = np.array([[2892.33, 0, 823.205 + vid], [0, 2883.18, 619.071 - vid], [0, 0, 1]])
    rows = ["extrinsic"] + [" ".join(f"{x:.6g}" for x in r) + " " for r in E] + ["", "intrinsic"] + \
           [" ".join(f"{x:.6g}" for x in r) + " " for r in K] + ["", f"425 {2.5 + 0.125 * vid:g} ", ""]
    return "\n".join(rows)


PAIR_TEXT = "3\n0\n2 1 2036.53 2 1243.89 \n1\n2 0 2036.53 2 1243.89 \n2\n2 1 1500.5 0 1243.89 \n"


def build_scan(root, scan="scan1", hw=(1200, 1600)):
    from PIL import Image
    os.makedirs(os.path.join(root, scan, "cams"), exist_ok=True)
    os.makedirs(os.path.join(root, scan, "images"), exist_ok=True)
    with open(os.path.join(root, scan, "pair.txt"), "w") as f:
        f.write(PAIR_TEXT)
    ys, xs = np.mgrid[0:hw[0], 0:hw[1]]
    for vid in range(3):
        with open(os.path.join(root, scan, "cams", f"{vid:0>8}_cam.txt"), "w") as f:
            f.write(cam_text(vid))
        img = np.stack([(xs * (vid + 1) + ys) % 256, (xs + 2 * ys * (vid + 1)) % 256, (xs * ys // 97 + 40 * vid) % 256],
                       axis=-1).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, scan, "images", f"{vid:0>8}.jpg"), quality=90)
    listfile = os.path.join(root, "test.txt")
    with open(listfile, "w") as f:
        f.write(scan + "\n")
    return listfile


def build_train_set(root, scan="scan7", hw=(512, 640)):
    """Yao's training layout (dtu_yao.py): shared Cameras/, Rectified/<scan>_train, Depths/<scan>_train;
    3 viewpoints x 7 light conditions of 640x512 PNGs, quarter-size depth PFMs and mask PNGs."""
    from PIL import Image
    import struct
    os.makedirs(os.path.join(root, "Cameras"), exist_ok=True)
    os.makedirs(os.path.join(root, "Rectified", scan + "_train"), exist_ok=True)
    os.makedirs(os.path.join(root, "Depths", scan + "_train"), exist_ok=True)
    with open(os.path.join(root, "Cameras", "pair.txt"), "w") as f:
        f.write(PAIR_TEXT)
    ys, xs = np.mgrid[0:hw[0], 0:hw[1]]
    qy, qx = np.mgrid[0:hw[0] // 4, 0:hw[1] // 4]
    for vid in range(3):
        with open(os.path.join(root, "Cameras", f"{vid:0>8}_cam.txt"), "w") as f:
            f.write(cam_text(vid))
        for light in range(7):
            img = np.stack([(xs * (vid + 1) + ys + 9 * light) % 256, (xs + 2 * ys * (vid + 1)) % 256,
                            (xs * ys // 53 + 40 * vid + light) % 256], axis=-1).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(root, "Rectified", scan + "_train",
                                                   f"rect_{vid + 1:0>3}_{light}_r5000.png"))
        depth = (500.0 + 0.75 * qx + 0.5 * qy + 10 * vid).astype("<f4")
        with open(os.path.join(root, "Depths", scan + "_train", f"depth_map_{vid:0>4}.pfm"), "wb") as f:
            f.write(f"Pf\n{depth.shape[1]} {depth.shape[0]}\n-1\n".encode())
            f.write(np.flipud(depth).tobytes())
        mask = (((qx // 8 + qy // 8 + vid) % 3) > 0).astype(np.uint8) * 255
        Image.fromarray(mask).save(os.path.join(root, "Depths", scan + "_train", f"depth_visual_{vid:0>4}.png"))
    listfile = os.path.join(root, "train.txt")
    with open(listfile, "w") as f:
        f.write(scan + "\n")
    return listfile
