#!/usr/bin/env python3
"""conv0 tuning: cycles of wave 0 per phase, summed per workgroup (MVS_CONV0_VARIANT=36 build
of the persistent kernel writes them into the buffer passed as `residual`)."""
import os
import sys

import numpy as np
import torch

os.environ["MVS_CONV0_VARIANT"] = "36"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
D, h, w = 192, 296, 400
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(1, D, h, 4, w, 8, device=dev, generator=g)
wt = torch.randn(8, 32, 3, 3, 3, device=dev, generator=g) * 0.05
sc = torch.rand(8, device=dev, generator=g) + 0.5
sh = torch.randn(8, device=dev, generator=g) * 0.1
pk = ops.pack_conv3d_weight(wt, False, 1)
dbg = torch.zeros(1, D, h, w, 8, device=dev)
for _ in range(3):
    dbg.zero_()
    y = ops.conv3d(x, wt, sc, sh, dbg, True, False, 1, channels_last=True, packed=pk, impl=ops.IMPL_MFMA, in_c8=True)
torch.cuda.synchronize()
ncu = torch.cuda.get_device_properties(0).multi_processor_count
t = dbg.view(-1).view(torch.int64)[: ncu * 8].reshape(ncu, 8).cpu().numpy().astype(np.float64)
names = ["wait_dma", "barrier", "issue+geom", "mfma", "stores", "acc_drain"]
tot = t[:, :6].sum(1)
ntile = 13 * 74 * 48 / ncu
print("workgroups", ncu, "tiles/wg", ntile, "cycles/wg mean", tot.mean(), "max", tot.max())
for k, nme in enumerate(names):
    print(f"{nme:12s} {t[:, k].mean() / tot.mean() * 100:6.2f} %   per tile {t[:, k].mean() / ntile:9.0f} cycles")
