"""PFM files as the reference reads and writes them (MVSNet/datasets/data_io.py:6-71).

Format: "Pf\\n" (one channel) or "PF\\n" (three), "<width> <height>\\n", a scale line whose
SIGN is the byte order (negative = little-endian), then float32 rows stored bottom-to-top.
"""
import re
import sys

import numpy as np

_CHANNELS = {b"Pf": 1, b"PF": 3}
_DIMS = re.compile(rb"^(\d+)\s(\d+)\s$")


def _parse_header(stream):
    """-> (channels, width, height, byte-order char, |scale|); raises on a malformed header."""
    channels = _CHANNELS.get(stream.readline().rstrip())
    if channels is None:
        raise Exception("Not a PFM file.")
    dims = _DIMS.match(stream.readline())
    if dims is None:
        raise Exception("Malformed PFM header.")
    signed_scale = float(stream.readline().rstrip())
    return channels, int(dims.group(1)), int(dims.group(2)), ("<" if signed_scale < 0 else ">"), abs(signed_scale)


def read_pfm(filename):
    """-> (float32 array [H,W] or [H,W,3] in the file's byte order, top row first; scale > 0)."""
    with open(filename, "rb") as stream:
        channels, width, height, order, scale = _parse_header(stream)
        payload = np.frombuffer(stream.read(), dtype=order + "f4")
    rows = payload.reshape((height, width, 3) if channels == 3 else (height, width))
    return rows[::-1], scale      # stored bottom-to-top


def save_pfm_rows_bottom_up(filename, rows, scale=1):
    """PFM of a float32 [H,W] image whose rows are ALREADY in file order (bottom row first, as save_pfm stores
    them) and contiguous: header + the buffer itself, no intermediate copy -- the write releases the GIL, which the
    eval driver's launching thread needs (mvs_amd/tools/eval_depth.py)."""
    if rows.dtype.name != "float32" or rows.ndim != 2 or not rows.flags["C_CONTIGUOUS"]:
        raise Exception("rows must be a C-contiguous float32 [H,W] array.")
    little_endian = rows.dtype.byteorder == "<" or (rows.dtype.byteorder == "=" and sys.byteorder == "little")
    with open(filename, "wb") as stream:
        stream.write(b"Pf\n%d %d\n%f\n" % (rows.shape[1], rows.shape[0], -scale if little_endian else scale))
        stream.write(memoryview(rows).cast("B"))


def save_pfm(filename, image, scale=1):
    """image: float32 [H,W], [H,W,1] or [H,W,3]."""
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    grey = image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1)
    if not grey and not (image.ndim == 3 and image.shape[2] == 3):
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    order = image.dtype.byteorder
    little_endian = order == "<" or (order == "=" and sys.byteorder == "little")
    header = b"%s\n%d %d\n%f\n" % (b"Pf" if grey else b"PF", image.shape[1], image.shape[0],
                                  -scale if little_endian else scale)
    with open(filename, "wb") as stream:
        stream.write(header)
        stream.write(np.ascontiguousarray(image[::-1]).tobytes())
