#!/usr/bin/env python3
"""What the range guard's fp32 path costs at BASELINE configs[1]: the CostRegNet call on the bench's variance volume, clean and with
ONE NaN voxel (every two-piece layer then takes guard_direct_conv, mvs_amd/csrc/conv_guard.h).  python scripts/exp_guard_cost.py"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import ops, synth  # noqa: E402
from mvs_amd.models import MVSNet  # noqa: E402

dev = torch.device("cuda:0")
H, W, V, D = 1184, 1600, 5, 192
rng = np.random.default_rng(0)
imgs = torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev)
proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
model = MVSNet(refine=False)
model.load_state_dict(synth.random_state_dict(0))
model = model.to(dev).eval()
res = {}
with torch.no_grad():
    for name, poison in (("clean", False), ("one NaN pixel in a source image", True)):
        x = imgs.clone()
        if poison:
            x[0, 1, 0, 600, 800] = float("nan")
        model(x, proj, dv)
        torch.cuda.synchronize()
        n0 = ops.guard_fallback_count()
        t0 = time.perf_counter()
        out = model(x, proj, dv)
        torch.cuda.synchronize()
        res[name] = {"forward_ms": round((time.perf_counter() - t0) * 1e3, 2), "guard_fallbacks": ops.guard_fallback_count() - n0,
                     "nan_depth_pixels": int(torch.isnan(out["depth"]).sum()), "pixels": out["depth"].numel()}
        print(name, res[name], flush=True)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(REPO, "gpurun_out", "exp_guard_cost.json"), "w"), indent=1)
