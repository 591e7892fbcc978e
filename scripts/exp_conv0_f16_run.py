"""Runs the two-piece fp16 conv0 kernel a few times on variance-like data at BASELINE configs[1]'s volume (for the counter passes of
scripts/prof_conv0_f16_sq.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops
g = torch.Generator().manual_seed(5)
x = ((torch.randn(1, 192, 296, 4, 400, 8, generator=g) * torch.rand(1, 192, 296, 4, 400, 8, generator=g) ** 4).square()).cuda()
w = (torch.randn(8, 32, 3, 3, 3, generator=g) / 30).cuda()
pf = ops.pack_conv3d_weight_f16x3(w)
mx = ops.absmax(x)
for _ in range(4):
    ops.conv3d_c8_f16x3(x, pf, mx, None, None, None, True)
torch.cuda.synchronize()
