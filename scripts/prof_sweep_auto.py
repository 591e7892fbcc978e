"""kernel trace target: the auto-selected sweep, 6 calls (scripts/prof_clock_cmd.sh-style use with rocprofv3 --kernel-trace)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops, synth
dev = torch.device("cuda:0")
V, h, w, D = 5, 296, 400, 192
feats = torch.from_numpy(synth.smooth_features(np.random.default_rng(0), (V, 1, 32, h, w))).to(dev)
rts = ops.rot_trans_all(torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev), "device")
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
c4 = ops.nchw_to_c4(feats)
for _ in range(6):
    out = ops.costvol_variance_c16(c4[0], c4[1:], rts, dv, out_c8=True, fast=True)
torch.cuda.synchronize()
