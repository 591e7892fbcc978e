"""conv_split layers: the two-piece fp16 form (three products) against the three-piece bf16 form (six) -- time at the layer
shapes of BASELINE configs[1] and distance of both from a float64 convolution on small volumes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mvs_amd import ops

res = {"accuracy": {}, "time": {}}


def mk(shape, cin, cout, k, kd, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, cin, generator=g).clamp_min(0) * torch.rand(*shape, cin, generator=g)     # post-ReLU-like
    w = torch.randn(cout, cin, *([3] * (kd == 3)), k, k, generator=g) / (k * k * (3 if kd == 3 else 1) * cin) ** 0.5
    return x, w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1


for name, shape, cin, cout, k, kd, st in (("3d 16->16", (1, 9, 21, 37), 16, 16, 3, 3, 1), ("3d 32->32", (1, 8, 10, 35), 32, 32, 3, 3, 1),
                                          ("3d 64->64", (1, 5, 9, 18), 64, 64, 3, 3, 1), ("3d 8->16 s2", (1, 9, 20, 37), 8, 16, 3, 3, 2),
                                          ("2d 16->16", (3, 37, 70), 16, 16, 3, 1, 1), ("2d 32->32", (2, 21, 50), 32, 32, 3, 1, 1),
                                          ("2d 8->16 5x5 s2", (2, 37, 70), 8, 16, 5, 1, 2), ("2d 16->32 5x5 s2", (2, 37, 70), 16, 32, 5, 1, 2)):
    x, w, sc, sh = mk(shape, cin, cout, k, kd, len(name))
    conv = F.conv3d if kd == 3 else F.conv2d
    perm_in = (0, 4, 1, 2, 3) if kd == 3 else (0, 3, 1, 2)
    perm_out = (0, 2, 3, 4, 1) if kd == 3 else (0, 2, 3, 1)
    vw = (1, cout) + (1,) * (3 if kd == 3 else 2)
    y64 = (conv(x.permute(perm_in).double(), w.double(), stride=st, padding=k // 2) * sc.double().view(vw) + sh.double().view(vw)).permute(perm_out)
    y32 = (conv(x.permute(perm_in), w, stride=st, padding=k // 2) * sc.view(vw) + sh.view(vw)).permute(perm_out).double()
    xd, wd = x.cuda(), w.cuda()
    yb = ops.conv_split(xd, ops.pack_conv_weight_split(wd, st), cout, sc.cuda(), sh.cuda(), None, 0, kd=kd, stride=st)
    om = ops.absmax_block("cuda", zero=True)
    yf = ops.conv_split_f16(xd, ops.pack_conv_weight_split_f16(wd, st), cout, None, sc.cuda(), sh.cuda(), None, 0, kd=kd, stride=st, out_absmax=om)
    e = lambda y: {"max": float((y.double().cpu() - y64).abs().max()), "rms": float((y.double().cpu() - y64).pow(2).mean().sqrt())}
    res["accuracy"][name] = {"ymax": float(y64.abs().max()), "aten_fp32": e(y32), "bf16x6": e(yb), "f16x3": e(yf),
                             "out_absmax_ok": bool(ops.absmax_value(om) == yf.abs().max().item())}
    print(name, res["accuracy"][name], flush=True)

for name, shape, cin, cout, k, kd, st in (("conv1 8->16 s2", (1, 192, 296, 400), 8, 16, 3, 3, 2), ("conv2 16->16", (1, 96, 148, 200), 16, 16, 3, 3, 1),
                                          ("conv4 32->32", (1, 48, 74, 100), 32, 32, 3, 3, 1), ("conv6 64->64", (1, 24, 37, 50), 64, 64, 3, 3, 1),
                                          ("feat 16->16", (5, 592, 800), 16, 16, 3, 1, 1), ("feat 32->32", (5, 296, 400), 32, 32, 3, 1, 1),
                                          ("feat 8->16 5x5 s2", (5, 1184, 1600), 8, 16, 5, 1, 2), ("feat 16->32 5x5 s2", (5, 592, 800), 16, 32, 5, 1, 2)):
    x, w, sc, sh = mk(shape, cin, cout, k, kd, 3)
    xd, wd, sc, sh = x.cuda(), w.cuda(), sc.cuda(), sh.cuda()
    pb, pf = ops.pack_conv_weight_split(wd, st), ops.pack_conv_weight_split_f16(wd, st)
    mx = ops.absmax(xd); om = ops.absmax_block("cuda", zero=True)
    fns = {"bf16x6": lambda: ops.conv_split(xd, pb, cout, sc, sh, None, 1, kd=kd, stride=st),
           "f16x3": lambda: ops.conv_split_f16(xd, pf, cout, mx, sc, sh, None, 1, kd=kd, stride=st, out_absmax=om)}
    r = {}
    for rep in range(2):
        for kname, fn in fns.items():
            fn(); torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
            for a, b in ev:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            r[kname] = round(min(r.get(kname, 1e9), min(a.elapsed_time(b) for a, b in ev)), 4)
    res["time"][name] = r
    print(name, r, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/split_f16.json", "w"), indent=1)
