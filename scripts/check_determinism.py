"""Repeated forwards of MVSNet (configs[1]) and CascadeMVSNet (configs[2]) on the same input must return the same bits:
the kernels have no atomics on their results and no order-dependent reductions.  python scripts/check_determinism.py"""
import sys, torch
sys.path.insert(0, '.')
from mvs_amd import synth
from mvs_amd.models import MVSNet
dev = torch.device('cuda:0')
model = MVSNet(refine=False); model.load_state_dict(synth.random_state_dict(0), strict=False); model = model.to(dev).eval()
V, H, W, D = 5, 1184, 1600, 192
g = torch.Generator(device=dev).manual_seed(1)
imgs = torch.rand(1, V, 3, H, W, device=dev, generator=g)
proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
with torch.no_grad():
    ref = model(imgs, proj, dv)
    bad = 0
    for i in range(40):
        o = model(imgs, proj, dv)
        if not (torch.equal(o["depth"], ref["depth"]) and torch.equal(o["photometric_confidence"], ref["photometric_confidence"])):
            bad += 1
    print("MVSNet 40 repeated forwards, differing from the first:", bad)
from mvs_amd.models import CascadeMVSNet
cas = CascadeMVSNet(refine=False).to(dev).eval()
cp = {k: torch.from_numpy(synth.cas_proj_matrices(V, H // s, W // s)).to(dev) for k, s in (("stage1", 4), ("stage2", 2), ("stage3", 1))}
with torch.no_grad():
    r = cas(imgs, cp, dv)
    bad = sum(0 if torch.equal(cas(imgs, cp, dv)["depth"], r["depth"]) else 1 for _ in range(10))
    print("CascadeMVSNet 10 repeated forwards, differing:", bad)
