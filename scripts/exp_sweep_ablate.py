#!/usr/bin/env python3
"""Where the persistent sweep kernel's time goes at BASELINE configs[1] (tuning build: MVS_SWEEP_PERSIST=16,<flags>,2):
the full kernel against the same kernel without its per-tile plan, without its copies, without its tap set-up, blends or stores
(garbage results by design).   python scripts/exp_sweep_ablate.py [reps]"""
import os as _os; _os.environ.setdefault("MVS_HIP_TUNING", "1")   # needs python -m mvs_amd.build --tuning
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
FLAGS = {"full": 0, "no_dma": 8, "no_taps": 32, "no_blend": 4, "no_store": 2,
         "skeleton (no dma/taps/blend/store)": 8 + 32 + 4 + 2}


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs)
    return round(t[len(t) // 2], 4), round(t[0], 4)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    D, h, w, V = 192, 296, 400, 5
    g = torch.Generator(device=dev).manual_seed(0)
    feats = torch.randn(V, 1, 32, h, w, device=dev, generator=g)
    f4 = ops.nchw_to_c4(feats)
    proj = torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D)).to(dev)
    rts = ops.rot_trans_all(proj)
    amax = ops.absmax_block(dev)
    res = {}
    for fast in (True, False):
        for name, fl in FLAGS.items():
            os.environ["MVS_SWEEP_PERSIST"] = f"16,{fl},2"
            med, best = timeit(lambda: ops.costvol_variance_c16(f4[0], f4[1:], rts, dv, out_c8=True, fast=fast, absmax_out=amax), reps)
            res[f"{'fast' if fast else 'exact'} / {name}"] = {"median_ms": med, "best_ms": best}
            print(f"{'fast' if fast else 'exact'} / {name}: {med} ({best})", flush=True)
    os.environ.pop("MVS_SWEEP_PERSIST", None)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "exp_sweep_ablate.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
