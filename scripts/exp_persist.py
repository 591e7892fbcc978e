#!/usr/bin/env python3
"""Persistent variance kernel vs the per-tile kernel: bit-equality (exact mode), distance of the
fast mode, timing of the tuning variants at BASELINE configs[1].  python scripts/exp_persist.py [reps]"""
import os as _os; _os.environ.setdefault("MVS_HIP_TUNING", "1")   # needs python -m mvs_amd.build --tuning
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
C4 = False


def run(f16, rts, dv, persist, fast=False, out_c8=False, **kw):
    os.environ["MVS_SWEEP_PERSIST"] = persist
    try:
        return ops.costvol_variance_c16(f16[0], f16[1:], rts, dv, out_c8=out_c8, fast=fast, **kw)
    finally:
        os.environ.pop("MVS_SWEEP_PERSIST", None)


def cold_count():
    """records the last persistent launch queued for its cold kernel"""
    torch.cuda.synchronize()
    ws = next(iter(ops._variance_ws.values()), None)
    return int(ws[:4].view(torch.int32)[0]) if ws is not None else -1


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
    return round(t[len(t) // 2], 4), round(t[0], 4)


def scene(B, V, C, D, H, W, seed, wide=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    feats = torch.randn(V, B, C, H, W, device=dev, generator=g)
    proj = synth.proj_matrices(V, H, W, batch=B)
    if wide:   # a scaled source camera: footprints larger than any LDS share
        proj[:, 1, :2, :] *= 3.0
    proj = torch.from_numpy(proj).to(dev)
    dv = torch.from_numpy(synth.depth_values(D, batch=B, interval=synth.sweep_interval(D))).to(dev)
    rts = ops.rot_trans_all(proj)
    return (ops.nchw_to_c4(feats) if C4 else ops.nchw_to_c16(feats)), rts, dv


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    res = {"equal": {}, "fast_maxabs": {}, "time_ms": {}}
    cases = {"b2v3_ragged": (2, 3, 32, 9, 13, 37, 1, False), "b1v5_ragged": (1, 5, 32, 20, 30, 50, 2, False),
             "b1v5_d17": (1, 5, 32, 17, 41, 67, 3, False), "b1v3_wide": (1, 3, 32, 12, 40, 64, 4, True),
             "b1v2_c16": (1, 2, 16, 16, 24, 48, 5, False), "b2v7": (2, 7, 32, 5, 18, 35, 6, False)}
    global C4
    for name, (B, V, C, D, H, W, seed, wide) in cases.items():
        C4 = False
        f16, rts, dv = scene(B, V, C, D, H, W, seed, wide)
        want = run(f16, rts, dv, "0")
        want8 = run(f16, rts, dv, "0", out_c8=True)
        C4 = True
        f16, rts, dv = scene(B, V, C, D, H, W, seed, wide)
        for nw in ("16,0,2", "8,0,2", "16,64,2"):
            got = run(f16, rts, dv, nw)
            got8 = run(f16, rts, dv, nw, out_c8=True)
            res["equal"][f"{name}/nw{nw}"] = bool(torch.equal(got, want)) and bool(torch.equal(got8, want8))
            if not torch.equal(got, want):
                bad = (got != want) & ~(torch.isnan(got) & torch.isnan(want))
                res["equal"][f"{name}/nw{nw}/nbad"] = int(bad.sum())
                res["equal"][f"{name}/nw{nw}/maxabs"] = float((got - want).abs().nan_to_num().max())
                idx = bad.nonzero()[:4].tolist()
                res["equal"][f"{name}/nw{nw}/where"] = idx
            fast = run(f16, rts, dv, nw, fast=True)
            res["fast_maxabs"][f"{name}/nw{nw}"] = float((fast - want).abs().max())
            res["equal"][f"{name}/nw{nw}/cold"] = cold_count()
        res["fast_maxabs"][f"{name}/var_absmax"] = float(want.abs().max())
    print(json.dumps(res, indent=1), flush=True)

    # ---- full size timing
    D, h, w, V = 192, 296, 400, 5
    g = torch.Generator(device=dev).manual_seed(0)
    feats = torch.randn(V, 1, 32, h, w, device=dev, generator=g)
    f16 = ops.nchw_to_c16(feats)
    proj = torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D)).to(dev)
    rts = ops.rot_trans_all(proj)
    want = run(f16, rts, dv, "0", out_c8=True)
    f4 = ops.nchw_to_c4(feats)
    byt = (V * 32 * h * w + D + 32 * D * h * w) * 4
    for persist in ("0", "16,0,2", "16,64,2", "16,0,2", "16,64,2", "16,8,2", "16,72,2"):
        for fast in (False, True):
            if persist == "0" and fast:
                continue
            key = f"persist={persist} fast={int(fast)}"
            f16 = f4
            if persist == "0" or persist.startswith("c16:"):
                f16 = ops.nchw_to_c16(feats)
                persist = persist.replace("c16:", "")
            try:
                got = run(f16, rts, dv, persist, fast=fast, out_c8=True)
                flags = int(persist.split(",")[1]) if "," in persist else 0
                extra = {}
                if flags in (0, 1, 64):
                    extra["equal"] = bool(torch.equal(got, want))
                    extra["maxabs"] = float((got - want).abs().max())
                med, best = timeit(lambda: run(f16, rts, dv, persist, fast=fast, out_c8=True), reps)
                res["time_ms"][key] = {"median": med, "best": best, "TBs": round(byt / best / 1e9, 3),
                                       "cold": cold_count(), **extra}
            except Exception as e:  # noqa: BLE001
                res["time_ms"][key] = {"error": str(e)[:200]}
            print(key, res["time_ms"][key], flush=True)
    res["var_absmax_fullsize"] = float(want.abs().max())
    print(json.dumps(res["time_ms"], indent=1))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "exp_persist.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
