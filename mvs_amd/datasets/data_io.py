"""PFM files as the reference reads and writes them (MVSNet/datasets/data_io.py:6-71).

Format: "Pf\\n" (one channel) or "PF\\n" (three), "<width> <height>\\n", a scale line whose
SIGN is the byte order (negative = little-endian), then float32 rows stored bottom-to-top.
"""
import re
import sys

import numpy as np

_DIMS = re.compile(rb"^(\d+)\s(\d+)\s$")


def read_pfm(filename):
    """-> (array [H,W] or [H,W,3] float32 in file byte order, top row first; scale > 0)."""
    with open(filename, "rb") as f:
        magic = f.readline().rstrip()
        if magic not in (b"PF", b"Pf"):
            raise Exception("Not a PFM file.")
        m = _DIMS.match(f.readline())
        if not m:
            raise Exception("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        order = "<" if scale < 0 else ">"
        data = np.fromfile(f, order + "f")
    shape = (height, width, 3) if magic == b"PF" else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)


def save_pfm(filename, image, scale=1):
    """image: float32 [H,W], [H,W,1] or [H,W,3]."""
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        magic = b"PF\n"
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        magic = b"Pf\n"
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    with open(filename, "wb") as f:
        f.write(magic)
        f.write(b"%d %d\n" % (image.shape[1], image.shape[0]))
        f.write(b"%f\n" % (-scale if little else scale))
        np.flipud(image).tofile(f)
