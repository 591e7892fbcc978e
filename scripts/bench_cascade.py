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
    argv = list(sys.argv[1:])
    if "--steps" in argv:          # its value is not a positional size
        del argv[argv.index("--steps") + 1]
    args = [a for a in argv if not a.startswith("--")]
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
        # the dominant sweep / regulariser stage against its roofline (HIP events around the stage, steady state)
        feat = {"stage1": (H // 4, W // 4, 32), "stage2": (H // 2, W // 2, 16), "stage3": (H, W, 8)}
        work = {}
        for (st_name, (h, w, c)), nd in zip(feat.items(), net.ndepths):
            n0 = nd * h * w
            work[st_name + ".costvol_variance"] = ("hbm", (V * c * h * w + n0 + c * n0) * 4.0)   # per-pixel hypotheses: D*h*w depths
            fl = 2.0 * 27 * n0 * (c * 8 + 8 * 16 / 8 + 16 * 16 / 8 + 16 * 32 / 64 + 32 * 32 / 64 + 32 * 64 / 512 + 64 * 64 / 512
                                  + 64 * 32 / 512 + 32 * 16 / 64 + 16 * 8 / 8 + 8)
            work[st_name + ".costreg"] = ("mfma", fl)
        ms = {k: v[1] for k, v in res["stages_ms"].items() if k in work}
        dom = max(ms, key=ms.get)
        kind, amount = work[dom]
        if kind == "hbm":
            ach = amount / (ms[dom] * 1e-3) / 1e9
            res["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(ach / 8000.0, 4), "algorithmic_bytes": amount, "ms": ms[dom]}
        else:
            ach = amount / (ms[dom] * 1e-3) / 1e12
            peak = 2500.0 / 3.0 if ops.split_f16_enabled() else 2500.0 / 6.0
            res["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "algorithmic_flops": amount, "ms": ms[dom],
                               "peak_note": "the eleven layers of the stage's regulariser in one C call (mvs_costreg_fwd2_f32); fp16 dense "
                                            "MFMA peak / products per fp32 product of the split-operand kernels"}
        if "--golden" in sys.argv and (H, W, V) == (1184, 1600, 5):
            # the reference's own CascadeMVSNet CPU forward on these inputs (tests/golden/make_golden_fullsize.py: g13; the
            # 1184x1600 stage on the even-row, even-column grid)
            g = dict(np.load(os.path.join(REPO, "tests", "golden", "g13_cas_fullsize.npz")))
            res["golden"] = "tests/golden/g13_cas_fullsize.npz: the imported reference's CPU forward on the same seeded inputs"
            for k in ("stage1", "stage2", "stage3"):
                sub = 2 if k == "stage3" else 1
                res[k + "_depth_maxabs_vs_reference_mm"] = float(
                    (out[k]["depth"][:, ::sub, ::sub].cpu() - torch.from_numpy(g[k + "_depth"])).abs().max())
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
