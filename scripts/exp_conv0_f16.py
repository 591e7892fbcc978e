"""conv0 (32 -> 8): the two-piece fp16 kernel (three products, mvs_conv3d_c8_f16x3_f32) against the three-piece bf16
kernel (six products) -- time at BASELINE configs[1]'s volume, and the distance of both (and of ATen's fp32 convolution)
from a float64 convolution on smaller volumes, incl. ragged shapes, tiny and huge magnitudes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mvs_amd import ops

res = {"accuracy": {}, "time": {}}


def data(B, D, H, W, C, seed, mag=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, D, H, C // 8, W, 8, generator=g) * torch.rand(B, D, H, C // 8, W, 8, generator=g) ** 4).square() * mag
    w = torch.randn(8, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5
    return x, w, torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1


for (B, D, H, W, C), mag in (((1, 12, 20, 70, 32), 1.0), ((2, 7, 9, 37, 16), 1.0), ((1, 5, 30, 33, 8), 1.0),
                             ((1, 12, 20, 70, 32), 1e-12), ((1, 12, 20, 70, 32), 3e9)):
    x, w, sc, sh = data(B, D, H, W, C, D * 7 + W, mag)
    xn = x.permute(0, 3, 5, 1, 2, 4).reshape(B, C, D, H, W)            # [B,D,H,G,W,8] -> [B,C,D,H,W]
    y64 = F.conv3d(xn.double(), w.double(), padding=1) * sc.double().view(1, 8, 1, 1, 1) + sh.double().view(1, 8, 1, 1, 1)
    y32 = F.conv3d(xn, w, padding=1) * sc.view(1, 8, 1, 1, 1) + sh.view(1, 8, 1, 1, 1)
    xd, wd = x.cuda(), w.cuda()
    yb = ops.conv3d_c8_split(xd, ops.pack_conv3d_weight_split(wd), sc.cuda(), sh.cuda(), None, False)
    yf = ops.conv3d_c8_f16x3(xd, ops.pack_conv3d_weight_f16x3(wd), None, sc.cuda(), sh.cuda(), None, False)
    ref = y64.permute(0, 2, 3, 4, 1)
    e = lambda y: {"max": float((y.double().cpu() - ref).abs().max()), "rms": float((y.double().cpu() - ref).pow(2).mean().sqrt())}
    res["accuracy"][str((B, D, H, W, C, mag))] = {"ymax": float(ref.abs().max()), "aten_fp32": e(y32.permute(0, 2, 3, 4, 1)),
                                                  "bf16x6": e(yb), "f16x3": e(yf)}
    print((B, D, H, W, C, mag), res["accuracy"][str((B, D, H, W, C, mag))], flush=True)

x, w, sc, sh = data(1, 192, 296, 400, 32, 5)
xd, wd, sc, sh = x.cuda(), w.cuda(), sc.cuda(), sh.cuda()
pb, pf = ops.pack_conv3d_weight_split(wd), ops.pack_conv3d_weight_f16x3(wd)
mx = ops.absmax(xd)
fns = {"bf16x6": lambda: ops.conv3d_c8_split(xd, pb, sc, sh, None, True),
       "f16x3": lambda: ops.conv3d_c8_f16x3(xd, pf, mx, sc, sh, None, True),
       "absmax": lambda: ops.absmax(xd, mx)}
for rep in range(2):
    for k, fn in fns.items():
        fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        res["time"].setdefault(k, []).append({"min": round(t[0], 4), "med": round(t[4], 4)})
        print(k, t[0], t[4], flush=True)
d = (fns["bf16x6"]() - fns["f16x3"]()).abs()
res["fullsize_f16x3_vs_bf16x6"] = {"max": float(d.max()), "rms": float(d.pow(2).mean().sqrt())}
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/conv0_f16.json", "w"), indent=1)
