#!/usr/bin/env python3
"""CostRegNet's conv1 (8 -> 16, stride 2) at config 2's shape: MVS_CONV1_MARCH=1 (z-marching kernel, conv_s2_march.hip) against =0
(per-tile kernel of conv_split.hip); run once per setting (the switch is read once per process)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops
dev = torch.device("cuda:0")
D, H, W = (int(v) for v in (sys.argv[1:4] or (192, 296, 400)))
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(1, D, H, W, 8, device=dev, generator=g).clamp_min(0)
w = torch.randn(16, 8, 3, 3, 3, device=dev, generator=g) * 0.1
sc, sh = 0.5 + torch.rand(16, device=dev, generator=g), torch.randn(16, device=dev, generator=g) * 0.1
pk = ops.pack_conv_weight_split_f16(w, 2)
xa = ops.absmax(x)
fn = lambda: ops.conv_split_f16(x, pk, 16, xa, sc, sh, None, 1, kd=3, stride=2)
for _ in range(3): y = fn()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ref = torch.relu(torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3)[:, :, :32].double().cpu(), w.double().cpu(), stride=2, padding=1)
                 * sc.double().cpu().view(1, 16, 1, 1, 1) + sh.double().cpu().view(1, 16, 1, 1, 1))
err = float((y[:, :15].permute(0, 4, 1, 2, 3).double().cpu() - ref[:, :, :15]).abs().max())
print(json.dumps({"march": os.environ.get("MVS_CONV1_MARCH", "1"), "ms_best": round(min(ts), 4), "ms_avg": round(sum(ts) / len(ts), 4),
                  "max_err_vs_f64_first_planes": err, "GB_s": round((x.numel() + y.numel()) * 4 / min(ts) / 1e6, 1)}))
