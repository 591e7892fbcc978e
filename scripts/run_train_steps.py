#!/usr/bin/env python3
"""Three eager training steps at BASELINE configs[4] (640x512, V=3, D=192): a plain command for the rocprofv3 counter scripts."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import synth
from mvs_amd.models import MVSNet, mvsnet_loss
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = MVSNet(refine=False).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
H, W, V, D = 512, 640, 3, 192
rng = np.random.default_rng(0)
proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
imgs = torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev)
gt = torch.full((1, H // 4, W // 4), 680.0, device=dev)
mask = torch.ones_like(gt)
for _ in range(3):
    opt.zero_grad()
    out = model(imgs, proj, dv)
    mvsnet_loss(out["depth"], gt, mask).backward()
    opt.step()
torch.cuda.synchronize()
