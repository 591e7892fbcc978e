"""A rendered synthetic scene for the trained-weights parity case.

Every other fixture drives the path with white-noise images and seeded random weights.  A network that has
actually been TRAINED needs photo-consistent views, so this module renders them: a textured height field
(a tilted plane plus smooth relief) seen from the DTU-like camera arc of mvs_amd.synth, with the exact
per-pixel depth of every view as ground truth.  It stands in for DTU scan9, which neither container has
(/root/reference/MVSNet/datasets/dtu_yao.py:60-65,95-96 = what the real loader would read).

Reproducibility rule: the GPU box regenerates the images from the seed (113 MB per full-size case cannot
be a fixture) and the stored reference outputs were computed from THIS container's images, so the renderer
uses only + - * / floor on float64 arrays -- IEEE-exact, the same bits on any host.  No sin / exp / pow,
no BLAS, no LAPACK (a vectorised libm differs by an ulp between AVX2 and AVX-512 hosts).  The camera
matrices come from mvs_amd.synth (math.cos / math.sin on a handful of scalars, shared with every other
fixture).  Images are quantised to 8 bits and divided by 255 as a decoded JPEG would be
(dtu_yao_eval.py:60-67).
"""
import numpy as np

from . import synth

_TABLE = 256  # lattice period of the value-noise tables (power of two)


def _fade(f):
    return f * f * f * (f * (f * 6.0 - 15.0) + 10.0)


def _value_noise(table, u, v):
    """C2 value noise on a periodic integer lattice: table [_TABLE,_TABLE] float64, u / v lattice coordinates."""
    iu, iv = np.floor(u), np.floor(v)
    fu, fv = _fade(u - iu), _fade(v - iv)
    iu = iu.astype(np.int64) & (_TABLE - 1)
    iv = iv.astype(np.int64) & (_TABLE - 1)
    iu1, iv1 = (iu + 1) & (_TABLE - 1), (iv + 1) & (_TABLE - 1)
    a, b, c, d = table[iv, iu], table[iv, iu1], table[iv1, iu], table[iv1, iu1]
    top = a + fu * (b - a)
    return top + fv * ((c + fu * (d - c)) - top)


class Scene:
    """z = z0 - nx x - ny y + relief(x, y) in the world frame of the arc cameras (reference camera at the origin
    looking down +z), with an albedo texture painted on it."""

    def __init__(self, seed, pixel_mm, z0=synth.DTU_TARGET_Z, tilt=(0.05, -0.08), relief_mm=18.0, relief_wavelength=120.0,
                 octaves=6):
        rng = np.random.default_rng(seed)
        self.z0, self.nx, self.ny = float(z0), float(tilt[0]), float(tilt[1])
        self.relief = [(relief_mm * 0.5 ** o, relief_wavelength * 0.5 ** o, rng.random((_TABLE, _TABLE))) for o in range(3)]
        # texture octaves in units of the reference view's pixel footprint: 2.5, 5, 10 ... pixels, amplitude ~ sqrt(wavelength)
        self.tex = []
        for ch in range(3):
            self.tex.append([(2.0 ** (0.5 * o), 2.5 * pixel_mm * 2.0 ** o, rng.random((_TABLE, _TABLE)),
                              rng.random(2) * _TABLE) for o in range(octaves)])
        self.norm = sum(a for a, _, _, _ in self.tex[0])

    def height(self, x, y):
        z = self.z0 - self.nx * x - self.ny * y
        for amp, lam, tab in self.relief:
            z = z + amp * (_value_noise(tab, x / lam + 17.0, y / lam + 5.0) - 0.5)
        return z

    def albedo(self, x, y):
        out = []
        for ch in range(3):
            s = np.zeros_like(x)
            for amp, lam, tab, off in self.tex[ch]:
                s = s + amp * _value_noise(tab, x / lam + off[0], y / lam + off[1])
            out.append(0.5 + 2.4 * (s / self.norm - 0.5))   # value noise sums crowd around 0.5: stretch the contrast
        return out


def _camera_rays(K, E, h, w):
    """Camera centre C [3] and world-frame ray directions (dx, dy, dz) [h,w] through the pixel centres; explicit arithmetic."""
    K, E = np.asarray(K, np.float64), np.asarray(E, np.float64)
    R, t = E[:3, :3], E[:3, 3]
    C = [-(R[0, i] * t[0] + R[1, i] * t[1] + R[2, i] * t[2]) for i in range(3)]
    ys, xs = np.mgrid[0:h, 0:w]
    rx = (xs.astype(np.float64) - K[0, 2]) / K[0, 0]
    ry = (ys.astype(np.float64) - K[1, 2]) / K[1, 1]
    d = [R[0, i] * rx + R[1, i] * ry + R[2, i] for i in range(3)]   # R^T (rx, ry, 1)
    return C, d, R, t


def render(scene, K_img, Es, h, w, gains=None, iters=14):
    """Images [V,3,h,w] float32 (8-bit quantised / 255) and per-view depth maps [V,h,w] float32 of `scene` for
    pinhole cameras K_img [3,3] (image resolution), Es [V,4,4] world->camera."""
    imgs, depths = [], []
    for v, E in enumerate(Es):
        C, d, R, t = _camera_rays(K_img, E, h, w)
        s = (scene.z0 - C[2]) / d[2]
        for _ in range(iters):               # fixed-point: the surface is a height field with gentle slopes
            s = (scene.height(C[0] + s * d[0], C[1] + s * d[1]) - C[2]) / d[2]
        X = [C[i] + s * d[i] for i in range(3)]
        depths.append(R[2, 0] * X[0] + R[2, 1] * X[1] + R[2, 2] * X[2] + t[2])
        g, b = (1.0, 0.0) if gains is None else gains[v]
        chans = [np.floor(np.minimum(np.maximum(a * g + b, 0.0), 1.0) * 255.0 + 0.5) / 255.0 for a in scene.albedo(X[0], X[1])]
        imgs.append(np.stack(chans))
    return np.stack(imgs).astype(np.float32), np.stack(depths).astype(np.float32)


def image_intrinsics(feat_h, feat_w):
    """Image-resolution K whose rows 0-1 divided by 4 are synth.feature_intrinsics (dtu_yao_eval.py:54)."""
    K = synth.feature_intrinsics(feat_h, feat_w).copy()
    K[:2] *= 4.0
    return K


def eval_case(seed, H, W, V, D, rig=0, interval=None):
    """One parity case: images, projection matrices and depth planes in the reference loader's convention
    (dtu_yao_eval.py:93-100), plus the exact depth of the reference view at feature resolution."""
    fh, fw = H // 4, W // 4
    K_img = image_intrinsics(fh, fw)
    pixel_mm = synth.DTU_TARGET_Z / K_img[0, 0]
    scene = Scene(seed, pixel_mm)
    Es = synth.arc_extrinsics(V, rig)
    rng = np.random.default_rng(seed + 1000)
    gains = [(1.0, 0.0)] + [(0.92 + 0.16 * rng.random(), 0.04 * rng.random() - 0.02) for _ in range(V - 1)]
    imgs, _ = render(scene, K_img, Es, H, W, gains)
    _, depth_feat = render(scene, synth.feature_intrinsics(fh, fw), Es[:1], fh, fw)
    return dict(imgs=imgs[None], proj=synth.proj_matrices(V, fh, fw, rig=rig),
                depth_values=synth.depth_values(D, interval=synth.DTU_INTERVAL if interval is None else interval),
                gt=depth_feat, shape=(H, W, V, D))
