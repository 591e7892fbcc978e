"""Golden vectors for the cascade / pyramid glue from the REFERENCE's own functions (not a mirror):
  * CasMVSNet: F.interpolate(bilinear) -> models.module.get_depth_range_samples -> F.interpolate(trilinear),
    exactly the lines cas_mvsnet.py:129-152 run between two stages;
  * CVP-MVSNet: models.modules.calDepthHypo in test mode (modules.py:122-219) on an upsampled depth map.
Run in the build container:  python tests/golden/make_golden_glue.py      Only data is stored."""
import os
import pdb
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import T, _import_ref, save, synth  # noqa: E402


def main():
    arrs = {}
    mods = _import_ref("CasMVSNet", stubs=("torchvision", "torchvision.utils", "cv2"))
    gdrs = mods["models.module"].get_depth_range_samples
    g = torch.Generator().manual_seed(4)
    for name, (B, hp, wp, H, W, scale, nd) in {"s2": (2, 37, 50, 148, 200, 2, 32), "s3": (1, 74, 100, 148, 200, 1, 8)}.items():
        prev = 500 + 300 * torch.rand(B, hp, wp, generator=g)
        interval = 2.65 * scale
        cur = F.interpolate(prev.unsqueeze(1), [H, W], mode="bilinear", align_corners=False).squeeze(1)
        samples = gdrs(cur_depth=cur, ndepth=nd, depth_inteval_pixel=interval, dtype=torch.float32, device=cur.device,
                       shape=[B, H, W], max_depth=935.0, min_depth=425.0)
        out = F.interpolate(samples.unsqueeze(1), [nd, H // scale, W // scale], mode="trilinear", align_corners=False).squeeze(1)
        arrs[f"cas_{name}_prev"] = prev
        arrs[f"cas_{name}_out"] = out
        arrs[f"cas_{name}_meta"] = np.array([H, W, scale, nd], dtype=np.int64)
        arrs[f"cas_{name}_interval"] = np.float64(interval)
    # ---- CVP calDepthHypo
    for s in ("torchvision", "torchvision.utils", "cv2"):
        sys.modules.setdefault(s, types.ModuleType(s))
    for k in [k for k in sys.modules if k in ("models", "utils") or k.startswith("models.")]:
        del sys.modules[k]
    torch.Tensor.cuda = lambda self, *a, **k: self
    pdb.set_trace = lambda *a, **k: None
    sys.path.insert(0, "/root/reference/CVP-MVSNet")
    from models import modules as refmod
    sys.path.pop(0)
    np.seterr(all="warn")
    cams = synth.cvp_cameras(2, 128, 160, batch=2)
    depth = torch.rand((2, 64, 80), generator=g) * 300 + 500
    K_ref = T(cams["ref_in"]).clone(); K_ref[:, :2] /= 2
    K_src = T(cams["src_in"]).clone(); K_src[:, :, :2] /= 2
    args = types.SimpleNamespace(mode="test")
    with torch.no_grad():
        hyp = refmod.calDepthHypo(args, depth.clone(), K_ref, K_src, T(cams["ref_ex"]), T(cams["src_ex"]),
                                  T(cams["depth_min"]), T(cams["depth_max"]), 0)
    arrs.update(cvp_depth_up=depth, cvp_K_ref=K_ref, cvp_K_src=K_src, cvp_ref_ex=cams["ref_ex"], cvp_src_ex=cams["src_ex"],
                cvp_hypos=hyp)
    save("g16_glue", **arrs)
    print({k: tuple(np.asarray(v).shape) for k, v in arrs.items()})


if __name__ == "__main__":
    main()
