"""Input recipes of the config-sized parity cases (numpy + seeded weights only; no GPU, no reference).

Shared by tests/golden/make_golden_configs.py (which runs the imported reference on them in the build
container) and the -m gpu tests (which run the HIP path on the same inputs and compare with the stored
outputs).  scene 0 = the recipe bench.py times (image seed 0, weights seed 0, SURVEY 8(d) camera arc);
scene 1 = the second parity scene of VERDICT r02 item 3: image seed 1, weight seed 1, camera rig 1."""
import numpy as np

from mvs_amd import synth


def train_case():
    """BASELINE configs[4], one GPU's share: 640x512, V=3, D=192, B=1, ground truth = plane at 680 mm + noise."""
    H, W, V, D = 512, 640, 3, 192
    rng = np.random.default_rng(17)
    imgs = synth.images(rng, 1, V, H, W)
    gt = (synth.DTU_TARGET_Z + 20 * rng.standard_normal((1, H // 4, W // 4))).astype(np.float32)
    mask = (rng.random((1, H // 4, W // 4)) > 0.1).astype(np.float32)
    return dict(imgs=imgs, proj=synth.proj_matrices(V, H // 4, W // 4), depth_values=synth.depth_values(D),
                gt=gt, mask=mask, sd=synth.random_state_dict(0), shape=(H, W, V, D))


def eval_small_case():
    """BASELINE configs[0]: 640x512, N=3, D=48 (interval 2.5 * 1.06 * 4)."""
    H, W, V, D = 512, 640, 3, 48
    rng = np.random.default_rng(18)
    return dict(imgs=synth.images(rng, 1, V, H, W), proj=synth.proj_matrices(V, H // 4, W // 4),
                depth_values=synth.depth_values(D, interval=synth.sweep_interval(D)),
                sd=synth.random_state_dict(0), shape=(H, W, V, D))


def mvsnet_fullsize_case(scene=0):
    """BASELINE configs[1]: 1600x1184, N=5, D=192."""
    H, W, V, D = 1184, 1600, 5, 192
    rng = np.random.default_rng(scene)
    return dict(imgs=synth.images(rng, 1, V, H, W), proj=synth.proj_matrices(V, H // 4, W // 4, rig=scene),
                depth_values=synth.depth_values(D), sd=synth.random_state_dict(scene), shape=(H, W, V, D))


def cas_fullsize_case(scene=0):
    """BASELINE configs[2]: CasMVSNet 1600x1184, N=5, 48/32/8 hypotheses."""
    H, W, V = 1184, 1600, 5
    rng = np.random.default_rng(scene)
    proj = {f"stage{s + 1}": synth.cas_proj_matrices(V, H // sc, W // sc, rig=scene) for s, sc in enumerate((4, 2, 1))}
    return dict(imgs=synth.images(rng, 1, V, H, W), proj=proj, depth_values=synth.depth_values(192),
                sd=synth.cas_random_state_dict(scene), shape=(H, W, V))


def cvp_fullsize_case(scene=0):
    """BASELINE configs[3]: CVP-MVSNet 1920x1056, 7 views, 5 levels."""
    H, W, nsrc, nscale = 1056, 1920, 6, 5
    rng = np.random.default_rng(scene)
    return dict(imgs=synth.images(rng, 1, nsrc + 1, H, W), cams=synth.cvp_cameras(nsrc, H, W, rig=scene),
                sd=synth.cvp_random_state_dict(scene), nsrc=nsrc, nscale=nscale, shape=(H, W, nsrc, nscale))
