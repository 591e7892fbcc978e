"""Golden vector for the 3-stage CasMVSNet forward (BASELINE config 3 at small size).

Run in the build container (needs /root/reference):   python tests/golden/make_golden_cas.py
Imports the reference's CascadeMVSNet (CasMVSNet/models/cas_mvsnet.py:69) with the
seeded weights of mvs_amd.synth.cas_random_state_dict, runs its CPU forward on a
synthetic 64x96, 3-view input with the default cascade (ndepths 48/32/8, interval
ratios 4/2/1) and stores inputs + per-stage outputs.  Only data is stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import T, _import_ref, save, synth  # noqa: E402

SEED = 11


def main():
    torch.set_num_threads(8)
    rng = np.random.default_rng(SEED)
    mods = _import_ref("CasMVSNet", stubs=("torchvision", "torchvision.utils", "cv2"))
    net = mods["models.cas_mvsnet"].CascadeMVSNet(refine=False)
    net.load_state_dict(synth.cas_random_state_dict(SEED))
    net.eval()
    B, V, H, W = 1, 3, 64, 96
    imgs = synth.images(rng, B, V, H, W)
    proj = {f"stage{s + 1}": synth.cas_proj_matrices(V, H // sc, W // sc, batch=B)
            for s, sc in enumerate((4, 2, 1))}
    dv = synth.depth_values(192, batch=B)
    feats = {}
    hk = net.feature.register_forward_hook(
        lambda m, i, o: feats.setdefault("f", []).append({k: v.clone() for k, v in o.items()}))
    with torch.no_grad():
        out = net(T(imgs), {k: T(v) for k, v in proj.items()}, T(dv))
    hk.remove()
    arrs = dict(imgs=imgs, depth_values=dv, seed=np.int64(SEED))
    for k, v in proj.items():
        arrs["proj_" + k] = v
    for s in ("stage1", "stage2", "stage3"):
        arrs[s + "_depth"] = out[s]["depth"]
        arrs[s + "_conf"] = out[s]["photometric_confidence"]
        arrs[s + "_feat_ref"] = feats["f"][0][s]
    save("g9_cas_cascade", **arrs)
    for s in ("stage1", "stage2", "stage3"):
        d = out[s]["depth"]
        print(s, tuple(d.shape), float(d.min()), float(d.max()),
              float(out[s]["photometric_confidence"].mean()))


if __name__ == "__main__":
    main()
