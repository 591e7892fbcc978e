#!/usr/bin/env python3
"""Depth-filter stage at DTU scale: 49 reference views x 10 source views, 296x400 depth maps.
GPU kernel (mvs_geo_consistency_f32, depth maps resident) vs the numpy restatement of
eval.py:136-262 on the host for a sample of reference views.  python scripts/bench_geo_filter.py"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import ops, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    H, W, V, S = 296, 400, 7, 6          # 7 synthetic cameras; each view checked against the 6 others, cycled to 10
    depths, K, E = synth.plane_depth_maps(V, H, W)
    src_ids = [[(r + 1 + k) % V for k in range(10)] for r in range(49)]
    src_ids = [[s if s != r % V else (s + 1) % V for s in ids] for r, ids in enumerate(src_ids)]
    gd = torch.from_numpy(depths).to(dev)
    def one(r):
        ids = src_ids[r]
        return ops.geo_consistency(gd[r % V], K, E[r % V], gd[ids], [K] * 10, E[ids], per_view=False)
    for r in range(3):
        one(r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(49):
        out = one(r)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); one(0); e1.record(); torch.cuda.synchronize()
    res = {"config": {"H": H, "W": W, "ref_views": 49, "src_per_ref": 10}, "gpu_ms_per_scan": round(gpu_ms, 2),
           "kernel_ms_per_ref_view": round(e0.elapsed_time(e1), 4),
           "geo_ge3_fraction": float((out["geo_mask_sum"] >= 3).float().mean())}
    if "--no-cpu" not in sys.argv:
        from oracle import geo_filter as gf
        t0 = time.perf_counter()
        for r in range(2):
            ids = src_ids[r]
            gf.fuse_reference_view(depths[r % V], K, E[r % V], depths[ids], [K] * 10, E[ids])
        res["numpy_ms_per_scan_extrapolated"] = round((time.perf_counter() - t0) / 2 * 49 * 1e3, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
