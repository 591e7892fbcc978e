#!/usr/bin/env python3
"""Experiment: reference views alternated over S HIP streams (steps of different views overlap
on the GPU).  python scripts/exp_two_streams.py [S] [steps]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import synth  # noqa: E402
from mvs_amd.models import MVSNet  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
H, W, V, D = 1184, 1600, 5, 192
rng = np.random.default_rng(0)
imgs = torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev)
proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
model = MVSNet(refine=False)
model.load_state_dict(synth.random_state_dict(0))
model = model.to(dev).eval()
streams = [torch.cuda.Stream() for _ in range(S)]
with torch.no_grad():
    for i in range(2 * S):
        with torch.cuda.stream(streams[i % S]):
            out = model(imgs, proj, dv)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % S]):
            out = model(imgs, proj, dv)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
print(f"streams={S} steps={K}: {el / K * 1e3:.3f} ms per reference view, {K / el:.1f} depth-maps/s")
