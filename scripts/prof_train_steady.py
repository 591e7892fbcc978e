#!/usr/bin/env python3
"""Steady-state kernel breakdown of one training step (torch.profiler)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import synth
from mvs_amd.models import MVSNet, mvsnet_loss
dev = torch.device("cuda:0")
feat_impl = sys.argv[1] if len(sys.argv) > 1 else None
torch.manual_seed(1)
model = MVSNet(refine=False).to(dev).train()
if feat_impl:
    model.train_feature_impl = feat_impl
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
H, W, V, D = 512, 640, 3, 192
rng = np.random.default_rng(0)
proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
imgs = torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev)
gt = torch.full((1, H // 4, W // 4), 680.0, device=dev)
mask = torch.ones_like(gt)
def step():
    opt.zero_grad()
    out = model(imgs, proj, dv)
    mvsnet_loss(out["depth"], gt, mask).backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:     # kernels only
    step(); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:70]
tot = sum(e.device_time_total for e in prof.key_averages())
print("train_feature_impl", model.train_feature_impl, "total kernel ms", round(tot / 1e3, 2))
for e in rows:
    print(f"{e.key[:110]:110s} n={e.count:4d} ms={e.device_time_total/1e3:7.2f}")
