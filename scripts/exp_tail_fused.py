#!/usr/bin/env python3
"""conv11 + prob: the fused kernel (tail_fused.hip) against the two launches, at config 2's shape (or D H W of the OUTPUT volume).
    python scripts/exp_tail_fused.py [D H W] [--reps N]"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import ops  # noqa: E402


def main():
    argv = list(sys.argv[1:])
    if "--reps" in argv:
        del argv[argv.index("--reps") + 1]
    args = [a for a in argv if not a.startswith("--")]
    D, H, W = (int(v) for v in (args[:3] or (192, 296, 400)))
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(1, D // 2, H // 2, W // 2, 16, device=dev, generator=g).clamp_min(0)
    skip = torch.randn(1, D, H, W, 8, device=dev, generator=g).clamp_min(0)
    w11 = torch.randn(16, 8, 3, 3, 3, device=dev, generator=g) * 0.08
    sc = 0.5 + torch.rand(8, device=dev, generator=g)
    sh = torch.randn(8, device=dev, generator=g) * 0.1
    pw = torch.randn(1, 8, 3, 3, 3, device=dev, generator=g) * 0.1
    pb = torch.randn(1, device=dev, generator=g) * 0.05
    xa, sa = ops.absmax(x), ops.absmax(skip)
    tail = ops.pack_costreg_tail(w11)
    pk = ops.pack_deconv_weight_split_f16(w11)
    pp = ops.pack_conv3d_weight(pw, False, 1)

    def fused():
        return ops.costreg_tail(x, xa, skip, sa, tail, sc, sh, pw, None, pb)[0]

    def unfused():
        d11 = ops.deconv_split_f16(x, pk, 8, xa, sc, sh, skip, True)
        return ops.conv3d(d11, pw, None, pb, None, False, False, 1, channels_last=True, packed=pp)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best, tot = 1e9, 0.0
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b)
            best, tot = min(best, t), tot + t
        return round(best, 4), round(tot / reps, 4)

    cf, cu = fused(), unfused().reshape(1, D, H, W)
    n0 = D * H * W
    res = {"shape": [D, H, W], "fused_ms_best_avg": timeit(fused), "unfused_ms_best_avg": timeit(unfused),
           "max_abs_diff": float((cf - cu).abs().max()), "max_abs": float(cu.abs().max()),
           "algorithmic_GB": round(4.0 * n0 * (2 + 8 + 1) / 1e9, 3)}
    res["fused_TBps"] = round(res["algorithmic_GB"] / res["fused_ms_best_avg"][0], 3)
    print(json.dumps(res))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "exp_tail_fused.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
