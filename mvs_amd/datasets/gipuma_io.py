"""Gipuma / fusibile exchange formats (CasMVSNet/gipuma.py:20-108): .dmb images, .P camera files,
the constant "fake" normal maps.  Byte-identical to what the reference's functions write
(tests/golden/g15_gipuma.npz)."""
import struct

import numpy as np


def read_gipuma_dmb(path):
    """gipuma.py:20-31 -> [H,W] or [H,W,C] float32."""
    with open(path, "rb") as f:
        _type, height, width, channel = struct.unpack("<iiii", f.read(16))
        array = np.fromfile(f, np.float32)
    array = array.reshape((width, height, channel), order="F")
    return np.transpose(array, (1, 0, 2)).squeeze()


def write_gipuma_dmb(path, image):
    """gipuma.py:34-54: header (1, height, width, channels) + the image channel-planar."""
    image = np.asarray(image)
    height, width = image.shape[0], image.shape[1]
    channels = image.shape[2] if image.ndim == 3 else 1
    if image.ndim == 3:
        image = np.transpose(image, (2, 0, 1)).squeeze()
    with open(path, "wb") as f:
        f.write(struct.pack("<iiii", 1, height, width, channels))
        image.tofile(f)


def read_camera_parameters(path):
    """gipuma.py:8-18: (intrinsics [3,3], extrinsics [4,4]) float32, intrinsics NOT divided."""
    with open(path) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    return intrinsics, extrinsics


def projection_matrix(intrinsic, extrinsic):
    """gipuma.py:69-77: float64 zeros 4x4 with K in its corner, times the float32 extrinsic -> [3,4] float64."""
    K = np.zeros((4, 4))
    K[:3, :3] = intrinsic
    return np.matmul(K, extrinsic)[0:3][:]


def write_gipuma_cam(path, P):
    """gipuma.py:79-85: three rows of str(value) + ' ', a blank line."""
    with open(path, "w") as f:
        for i in range(3):
            for j in range(4):
                f.write(str(P[i][j]) + " ")
            f.write("\n")
        f.write("\n")


def mvsnet_to_gipuma_cam(in_path, out_path):
    K, E = read_camera_parameters(in_path)
    write_gipuma_cam(out_path, projection_matrix(K, E))


def fake_gipuma_normal(depth):
    """gipuma.py:90-108: (1,1,1)/1.732050808 where depth > 0, else 0 -> [H,W,3] float32."""
    depth = np.asarray(depth)
    normal = np.tile(np.ones_like(depth).reshape(depth.shape[0], depth.shape[1], 1), [1, 1, 3]) / 1.732050808
    mask = np.float32(np.tile(np.squeeze(np.where(depth > 0, 1, 0)).reshape(depth.shape[0], depth.shape[1], 1), [1, 1, 3]))
    return np.float32(np.multiply(normal, mask))
