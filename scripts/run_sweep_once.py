"""One hand-over sweep (or, argv[1] = fp32, the fp32 sweep) at configs[1]'s shape, a few times: the command rocprofv3 --pmc wraps
(scripts/prof_sweep_handover.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mvs_amd import ops, synth
dev = torch.device("cuda:0")
V, h, w, D = 5, 296, 400, 192
g = torch.Generator(device=dev).manual_seed(1)
f = torch.randn(V, 1, 8, h, w, 4, device=dev, generator=g)
rts = ops.rot_trans_all(torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev), "device")
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
fa = ops.absmax(f)
for _ in range(4):
    if len(sys.argv) > 1 and sys.argv[1] == "fp32":
        ops.costvol_variance_c16(f[0], f[1:], rts, dv, out_c8=True, fast=True, absmax_out=ops.absmax_block(dev))
    else:
        ops.costvol_variance_handover(f[0], f[1:], rts, dv, fa, fast=True)
torch.cuda.synchronize()
