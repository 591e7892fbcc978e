"""Golden vector for the CVP-MVSNet forward (BASELINE configs[3] at small size).

Run in the build container (needs /root/reference):   python tests/golden/make_golden_cvp.py
Imports the reference's `network` (CVP-MVSNet/models/net.py:99) on the CPU -- its hard-coded
.cuda() calls made no-ops and the pdb.set_trace() left in its forward (net.py:152) disabled
for this process -- loads the seeded weights of mvs_amd.synth.cvp_random_state_dict, runs a
64x96, 2-source, 2-level input and stores inputs + outputs.  Only data is stored.
"""
import os
import pdb
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import T, save, synth  # noqa: E402

SEED = 21
REF = "/root/reference/CVP-MVSNet"


def main():
    warnings.filterwarnings("ignore")
    torch.set_num_threads(8)
    for s in ("torchvision", "torchvision.utils", "cv2"):
        sys.modules.setdefault(s, types.ModuleType(s))
    for k in [k for k in sys.modules if k in ("models", "utils") or k.startswith("models.")]:
        del sys.modules[k]
    torch.Tensor.cuda = lambda self, *a, **k: self      # the reference hard-codes .cuda()
    pdb.set_trace = lambda *a, **k: None                # ... and a breakpoint in its forward
    sys.path.insert(0, REF)
    from models import net as refnet
    sys.path.pop(0)
    np.seterr(all="warn")                               # modules.py sets 'raise' process-wide
    nscale, nsrc, H, W = 2, 2, 64, 96
    net = refnet.network(types.SimpleNamespace(nscale=nscale, nsrc=nsrc, mode="test"))
    net.load_state_dict(synth.cvp_random_state_dict(SEED))
    net.eval()
    rng = np.random.default_rng(SEED)
    imgs = synth.images(rng, 1, nsrc + 1, H, W)
    cams = synth.cvp_cameras(nsrc, H, W)
    with torch.no_grad():
        out = net(T(imgs[:, 0]), T(imgs[:, 1:]), T(cams["ref_in"]), T(cams["src_in"]), T(cams["ref_ex"]),
                  T(cams["src_ex"]), T(cams["depth_min"]), T(cams["depth_max"]))
    arrs = dict(imgs=imgs, seed=np.int64(SEED), nscale=np.int64(nscale), nsrc=np.int64(nsrc), **cams)
    for i, d in enumerate(out["depth_est_list"]):
        arrs[f"depth_level{i}"] = d
    arrs["prob_confidence"] = out["prob_confidence"]
    save("g11_cvp", **arrs)
    for i, d in enumerate(out["depth_est_list"]):
        print("level", i, tuple(d.shape), float(d.min()), float(d.max()))
    print("conf mean", float(out["prob_confidence"].mean()))


if __name__ == "__main__":
    main()
