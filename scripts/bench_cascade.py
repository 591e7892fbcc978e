#!/usr/bin/env python3
"""BASELINE configs[2]: the 3-stage CasMVSNet forward at 1600x1184, N=5, 48/32/8 hypotheses,
on cuda:0 -- per-stage HIP-event times, and with --parity the end-to-end depth difference to
the ATen CPU restatement (oracle/torch_ref.cascade_forward) on the same inputs.
Test/measurement infrastructure (imports oracle/ for the checker only).
    python scripts/bench_cascade.py [H W V] [--parity] [--steps K]"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import ops, synth  # noqa: E402
from mvs_amd.models.cas_mvsnet import CascadeMVSNet  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    H, W, V = (int(x) for x in (args[:3] or (1184, 1600, 5)))
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 5
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    rng = np.random.default_rng(0)
    imgs = torch.from_numpy(synth.images(rng, 1, V, H, W))
    projs = {f"stage{s + 1}": torch.from_numpy(synth.cas_proj_matrices(V, H // sc, W // sc))
             for s, sc in enumerate((4, 2, 1))}
    dv = torch.from_numpy(synth.depth_values(192))
    sd = synth.cas_random_state_dict(0)
    net = CascadeMVSNet()
    net.load_state_dict(sd)
    net.eval().to(dev)
    gi, gp, gd = imgs.to(dev), {k: v.to(dev) for k, v in projs.items()}, dv.to(dev)
    res = {"config": {"H": H, "W": W, "views": V, "ndepths": net.ndepths}}
    with torch.no_grad():
        for _ in range(2):
            out = net(gi, gp, gd)
        torch.cuda.synchronize()
        timer = ops.StageTimer()
        ops.set_timer(timer)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = net(gi, gp, gd)
        torch.cuda.synchronize()
        res["ms_per_ref_view"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
        ops.set_timer(None)
        res["stages_ms"] = {k: [c // steps, round(ms, 3)] for k, (c, ms) in sorted(timer.summary_ms().items())}
        res["peak_mem_gb"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
        if "--profile" in sys.argv:   # kernel breakdown of one steady-state forward
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                net(gi, gp, gd)
                torch.cuda.synchronize()
            rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
            print("total kernel ms", round(sum(e.device_time_total for e in rows) / 1e3, 2), file=sys.stderr)
            for e in rows[:36]:
                print(f"{e.key[:120]:120s} n={e.count:4d} ms={e.device_time_total / 1e3:7.2f}", file=sys.stderr)
        if "--parity" in sys.argv:
            from oracle import torch_ref as tr
            torch.set_num_threads(os.cpu_count())
            st = {}
            t0 = time.perf_counter()
            ref = tr.cascade_forward(imgs, projs, dv, sd, stages=st)
            res["cpu_seconds"] = round(time.perf_counter() - t0, 1)
            res["cpu_stage_seconds"] = {k: round(v, 2) for k, v in st.items()}
            for k in ("stage1", "stage2", "stage3"):
                e = (out[k]["depth"].cpu() - ref[k]["depth"]).abs()
                res[k + "_depth_maxabs_mm"] = float(e.max())
                res[k + "_depth_p999_mm"] = float(e.flatten().kthvalue(int(e.numel() * 0.999)).values)
                res[k + "_conf_maxabs"] = float((out[k]["photometric_confidence"].cpu() -
                                                 ref[k]["photometric_confidence"]).abs().max())
    print(json.dumps(res))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "bench_cascade.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
