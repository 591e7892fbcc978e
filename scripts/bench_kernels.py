#!/usr/bin/env python3
"""Per-kernel timing at BASELINE configs[1] shapes (HIP events on the launch
stream), for kernel tuning.  python scripts/bench_kernels.py [reps]"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import ops, synth  # noqa: E402


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs)
    return t[len(t) // 2], t[0]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    dev = torch.device("cuda:0")
    D, h, w, V = 192, 296, 400, 5
    g = torch.Generator(device=dev).manual_seed(0)
    res = {}

    def want(name):
        return only is None or any(name.startswith(o) for o in only)

    # ---- conv layers of CostRegNet at their config-2 resolutions
    layers = [("conv0", 32, 8, 1, False, 0), ("conv1", 8, 16, 2, False, 0), ("conv2", 16, 16, 1, False, 1),
              ("conv3", 16, 32, 2, False, 1), ("conv4", 32, 32, 1, False, 2), ("conv5", 32, 64, 2, False, 2),
              ("conv6", 64, 64, 1, False, 3), ("conv7", 64, 32, 2, True, 3), ("conv9", 32, 16, 2, True, 2),
              ("conv11", 16, 8, 2, True, 1), ("prob", 8, 1, 1, False, 0)]
    for name, ci, co, s, tr, lvl in layers:
        if not want(name):
            continue
        f = 2 ** lvl
        x = torch.randn(1, D // f, h // f, w // f, ci, device=dev, generator=g)
        wt = torch.randn((ci, co, 3, 3, 3) if tr else (co, ci, 3, 3, 3), device=dev, generator=g) * 0.05
        sc = torch.rand(co, device=dev, generator=g) + 0.5
        sh = torch.randn(co, device=dev, generator=g) * 0.1
        pk = ops.pack_conv3d_weight(wt, tr, s)
        c8 = name == "conv0" and os.environ.get("MVS_BENCH_C8", "1") == "1"
        if c8:
            x = x.reshape(1, D // f, h // f, ci // 8, w // f, 8)   # same bytes, blocked meaning
        med, best = timeit(lambda: ops.conv3d(x, wt, sc, sh, None, True, tr, s, channels_last=True,
                                              packed=pk, impl=ops.IMPL_MFMA, in_c8=c8), reps)
        nin = (D // f) * (h // f) * (w // f)
        flops = 2 * 27 * ci * co * (nin if tr else nin / (8 if s == 2 else 1))
        res[name] = {"ms": round(med, 4), "best_ms": round(best, 4),
                     "TFLOPs": round(flops / med / 1e9, 2), "frac_fp32_mfma": round(flops / med / 1e9 / 157.3, 4)}
        del x
    # ---- fused warp + variance
    if want("variance"):
        rng = np.random.default_rng(0)
        proj = torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev)
        dv = torch.from_numpy(synth.depth_values(D)).to(dev)
        feats = torch.randn(V, 1, h, w, 32, device=dev, generator=g)
        rts = torch.stack([ops.rot_trans(proj[:, v], proj[:, 0]) for v in range(1, V)])
        med, best = timeit(lambda: ops.costvol_variance_cl(feats[0], feats[1:], rts, dv, out_c8=True), reps)
        byt = (V * 32 * h * w + D + 32 * D * h * w) * 4
        res["variance"] = {"ms": round(med, 4), "best_ms": round(best, 4), "GBs": round(byt / med / 1e6, 1),
                           "frac_hbm": round(byt / med / 1e6 / 8000, 4)}
    if want("variance_lds"):
        proj = torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev)
        dv = torch.from_numpy(synth.depth_values(D)).to(dev)
        feats = torch.randn(V, 1, 32, h, w, device=dev, generator=g)
        fast = os.environ.get("MVS_BENCH_FAST", "0") == "1"
        c4 = ops.variance_persistent_supported(dv, 1, V, 32, h, w) and os.environ.get("MVS_BENCH_C16", "0") != "1"
        f16 = ops.nchw_to_c4(feats) if c4 else ops.nchw_to_c16(feats)
        rts = torch.stack([ops.rot_trans(proj[:, v], proj[:, 0]) for v in range(1, V)])
        med, best = timeit(lambda: ops.costvol_variance_c16(f16[0], f16[1:], rts, dv, out_c8=True, fast=fast), reps)
        byt = (V * 32 * h * w + D + 32 * D * h * w) * 4
        res["variance_lds"] = {"ms": round(med, 4), "best_ms": round(best, 4), "GBs": round(byt / med / 1e6, 1),
                               "frac_hbm": round(byt / med / 1e6 / 8000, 4)}
    if want("regress"):
        cost = torch.randn(1, D, h, w, device=dev, generator=g) * 4
        dv = torch.from_numpy(synth.depth_values(D)).to(dev)
        med, best = timeit(lambda: ops.softmax_regress_conf(cost, dv), reps)
        byt = (D * h * w + 2 * h * w) * 4
        res["regress"] = {"ms": round(med, 4), "best_ms": round(best, 4), "GBs": round(byt / med / 1e6, 1)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
