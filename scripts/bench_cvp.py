#!/usr/bin/env python3
"""BASELINE configs[3]: CVP-MVSNet coarse-to-fine forward at 1920x1056, 7 views (6 sources),
5 pyramid levels, on cuda:0 -- ms per reference view and peak memory.
    python scripts/bench_cvp.py [H W nsrc nscale] [--steps K] [--parity] [--profile]
Test/measurement infrastructure (imports oracle/ for the checker only)."""
import json
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import synth  # noqa: E402
from mvs_amd.models.cvp_mvsnet import network  # noqa: E402


def main():
    argv = list(sys.argv[1:])
    if "--steps" in argv:          # its value is not a positional size
        del argv[argv.index("--steps") + 1]
    args = [a for a in argv if not a.startswith("--")]
    H, W, nsrc, nscale = (int(x) for x in (args[:4] or (1056, 1920, 6, 5)))
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 5
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    rng = np.random.default_rng(0)
    imgs = torch.from_numpy(synth.images(rng, 1, nsrc + 1, H, W)).to(dev)
    cams = {k: torch.from_numpy(v).to(dev) for k, v in synth.cvp_cameras(nsrc, H, W).items()}
    sd = synth.cvp_random_state_dict(0)
    net = network(types.SimpleNamespace(nscale=nscale, nsrc=nsrc, mode="test"))
    net.load_state_dict(sd)
    net.eval().to(dev)

    def step():
        with torch.no_grad():
            return net(imgs[:, 0], imgs[:, 1:], cams["ref_in"], cams["src_in"], cams["ref_ex"], cams["src_ex"],
                       cams["depth_min"], cams["depth_max"])

    for _ in range(2):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    groups = None
    if "--profile" in sys.argv:   # kernel breakdown of one steady-state forward
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
        print("total kernel ms", round(sum(e.device_time_total for e in rows) / 1e3, 2), file=sys.stderr)
        for e in rows[:30]:
            print(f"{e.key[:120]:120s} n={e.count:4d} ms={e.device_time_total / 1e3:7.2f}", file=sys.stderr)
        groups = {"feature_pyramid_2d_convs": 0.0, "costreg_3d_convs": 0.0, "sweep_variance": 0.0, "other": 0.0}
        for e in rows:
            is2d = ("conv2d" in e.key or ("PersistCfg<" in e.key and ", 1, 32, 1, 3>" in e.key)   # one-plane-deep persistent tiles
                    or ("conv_split_kernel" in e.key and ", 1, 1, 3>" in e.key))                 # SplitCfg<Cin, Cout, KD = 1, S = 1, KH = 3>
            is3d = "conv3d" in e.key or "conv_split_kernel" in e.key or "deconv_split_kernel" in e.key
            k = ("feature_pyramid_2d_convs" if is2d else "costreg_3d_convs" if is3d
                 else "sweep_variance" if "variance" in e.key else "other")
            groups[k] += e.device_time_total / 1e3
        for name in [r.key for r in rows[:4]]:   # per-launch durations of the heaviest kernels, in launch order
            durs = [round(ev.device_time_total / 1e3, 3) for ev in prof.events() if ev.key == name]
            print(name[:70], durs, file=sys.stderr)
    res = {"config": {"H": H, "W": W, "views": nsrc + 1, "nscale": nscale},
           "ms_per_ref_view": round(elapsed / steps * 1e3, 3),
           "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           "depth_shapes": [list(d.shape) for d in out["depth_est_list"]],
           "finite": bool(all(torch.isfinite(d).all() for d in out["depth_est_list"]))}
    # algorithmic FLOPs (2 * k^d * Cin * Cout * outputs): the 9-layer pyramid CNN on every view and level
    # (net.py:22-51) and the refinement U-Net per level (net.py:52-89; its stride-2 level is 1/8 of the voxels)
    pyr_per_px = 2 * 9 * (3 * 64 + 64 * 64 * 2 + 64 * 32 + 32 * 32 * 2 + 32 * 16 + 16 * 16 * 2)
    reg_full = 2 * 27 * (16 * 16 * 2 + 16 * 1)                     # 16->16 x2, prob 16->1 at the level's resolution
    reg_half = 2 * 27 * (16 * 32 + 32 * 32 * 2 + 32 * 64 + 64 * 64 * 2 + 64 * 32) / 8 + 2 * 27 * 32 * 16 / 8
    flops = {"feature_pyramid": 0.0, "costreg": 0.0}
    for lvl in range(nscale):
        px = (H >> lvl) * (W >> lvl)
        flops["feature_pyramid"] += pyr_per_px * px * (nsrc + 1)
        flops["costreg"] += (reg_full + reg_half) * px * (48 if lvl == nscale - 1 else 8)
    total = sum(flops.values())
    res["algorithmic_TFLOP"] = {k: round(v / 1e12, 3) for k, v in flops.items()}
    res["achieved_TFLOPs"] = round(total / res["ms_per_ref_view"] / 1e9, 1)
    res["frac_fp32_mfma_peak"] = round(total / res["ms_per_ref_view"] / 1e9 / 157.3, 3)
    res["floor_ms_at_fp32_mfma_peak"] = round(total / 157.3e9, 1)
    if groups is not None:
        res["kernel_ms_by_group"] = {k: round(v, 2) for k, v in groups.items()}
        res["group_TFLOPs"] = {"feature_pyramid": round(flops["feature_pyramid"] / groups["feature_pyramid_2d_convs"] / 1e9, 1),
                               "costreg": round(flops["costreg"] / groups["costreg_3d_convs"] / 1e9, 1)}
    # whole-forward roofline: the pyramid CNN and the regularisers are matrix-pipe work on the two-piece fp16 kernels
    from mvs_amd import ops
    peak = 2500.0 / 3.0 if ops.split_f16_enabled() else 2500.0 / 6.0
    res["roofline"] = {"kernel": "whole forward (9-layer pyramid CNN x views x levels + refinement U-Nets)", "bound": "mfma",
                       "achieved": res["achieved_TFLOPs"], "peak": round(peak, 1), "unit": "TFLOP/s",
                       "frac": round(res["achieved_TFLOPs"] / peak, 4), "algorithmic_flops": total, "ms": res["ms_per_ref_view"],
                       "peak_note": "fp16 dense MFMA peak 2500 / 3 products per fp32 product; against the fp32 MFMA peak (157.3): "
                                    f"{res['frac_fp32_mfma_peak']}"}
    if "--golden" in sys.argv and (H, W, nsrc, nscale) == (1056, 1920, 6, 5):
        # the reference's own CVP `network` CPU forward on these inputs (tests/golden/make_golden_fullsize.py: g14; maps wider
        # than 1000 pixels on the even-row, even-column grid)
        g = dict(np.load(os.path.join(REPO, "tests", "golden", "g14_cvp_fullsize.npz")))
        res["golden"] = "tests/golden/g14_cvp_fullsize.npz: the imported reference's CPU forward on the same seeded inputs"
        res["depth_maxabs_vs_reference_mm_per_level"] = []
        for i, d in enumerate(out["depth_est_list"]):
            sub = 2 if d.shape[-1] > 1000 else 1
            res["depth_maxabs_vs_reference_mm_per_level"].append(
                float((d[:, ::sub, ::sub].cpu() - torch.from_numpy(g[f"depth_level{i}"])).abs().max()))
    if "--parity" in sys.argv:   # the checker: ATen CPU restatement on the same inputs
        from oracle import torch_ref as tr
        torch.set_num_threads(os.cpu_count())
        t0 = time.perf_counter()
        with torch.no_grad():
            ref = tr.cvp_forward(imgs[:, 0].cpu(), imgs[:, 1:].cpu(), *(cams[k].cpu() for k in (
                "ref_in", "src_in", "ref_ex", "src_ex", "depth_min", "depth_max")), sd, nscale)
        res["cpu_seconds"] = round(time.perf_counter() - t0, 1)
        res["depth_maxabs_mm_per_level"] = [float((a.cpu() - b).abs().max())
                                            for a, b in zip(out["depth_est_list"], ref["depth_est_list"])]
    print(json.dumps(res))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "bench_cvp.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
