#!/usr/bin/env python3
"""Error budget of the forward at BASELINE configs[1] (1600x1184, N=5, D=192): how far the HIP path and a float32 CPU
evaluation of the reference composition each are from the float64 evaluation, stage by stage (cumulative errors of the
pipeline up to that stage: FeatureNet -> variance volume -> conv0 -> regularised cost -> depth).

The float32 / float64 CPU evaluations are oracle/torch_ref.py (the reference's Python does not travel to the GPU box;
torch_ref's float32 mode reproduces the reference bit for bit in the build container: tests/golden/make_golden_configs.py
prints port_vs_reference = 0.0).  Writes gpurun_out/error_budget.json.   python scripts/error_budget.py [scene]"""
import json, os, sys, time
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import config_cases as cc   # noqa: E402
from mvs_amd import ops     # noqa: E402
from mvs_amd.models import MVSNet   # noqa: E402
from oracle import torch_ref as tr   # noqa: E402


def cpu_stages(c, dtype, threads):
    torch.set_num_threads(threads)
    t0 = time.time()
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in c["sd"].items()}
    imgs, proj, dv = f(c["imgs"]), f(c["proj"]), f(c["depth_values"])
    with torch.no_grad():
        V = imgs.shape[1]
        feats = [tr.feature_net(imgs[:, v], sd) for v in range(V)]
        out = {"feature": torch.stack(feats, 1).double()}
        var = tr.variance_volume(feats, [proj[:, v] for v in range(V)], dv)
        out["variance"] = var if dtype == torch.float64 else var.double()
        cap = {}
        cost = tr.cost_reg_net(var, sd, capture=cap).squeeze(1)
        del var
        out["conv0"] = cap["conv0"].double()
        out["cost"] = cost.double()
        out["depth"] = tr.regress(cost, dv)[0].double()
    print(f"cpu {dtype} forward: {time.time() - t0:.1f} s on {threads} threads", flush=True)
    return out


def hip_stages(c, fast):
    dev = torch.device("cuda:0")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    model = MVSNet(refine=False)
    model.load_state_dict(c["sd"])
    model = model.to(dev).eval()
    model.variance_fast = fast
    cap = {}
    o_var, o_c0, o_reg = ops.costvol_variance_c16, ops.conv3d_c8_f16x3, ops.softmax_regress_conf

    def w_var(*a, **k):
        r = o_var(*a, **k); cap["variance"] = r; return r

    def w_c0(*a, **k):
        r = o_c0(*a, **k); cap["conv0"] = r; return r

    def w_reg(cost, *a, **k):
        cap["cost"] = cost; return o_reg(cost, *a, **k)

    ops.costvol_variance_c16, ops.conv3d_c8_f16x3, ops.softmax_regress_conf = w_var, w_c0, w_reg      # (conv0: the two-piece fp16 kernel)
    try:
        with torch.no_grad():
            imgs = d(c["imgs"])
            B, V = imgs.shape[:2]
            f4 = model.extract_features(imgs.reshape(B * V, *imgs.shape[2:]))          # [B*V,8,h,w,4]
            ops.set_timer(ops.StageTimer())      # per-layer path (the one-call CostRegNet entry hides conv0's output)
            out = model(imgs, d(c["proj"]), d(c["depth_values"]))
    finally:
        ops.set_timer(None)
        ops.costvol_variance_c16, ops.conv3d_c8_f16x3, ops.softmax_regress_conf = o_var, o_c0, o_reg
    n, q, h, w, _ = f4.shape
    res = {"feature": f4.permute(0, 1, 4, 2, 3).reshape(B, V, q * 4, h, w).double().cpu(),
           "variance": ops.c8_to_nchw(cap["variance"]).double().cpu(),
           "conv0": cap["conv0"].permute(0, 4, 1, 2, 3).double().cpu(),
           "cost": cap["cost"].double().cpu(), "depth": out["depth"].double().cpu()}
    return res


def stats(x, truth):
    e = (x - truth).abs()
    return {"max": float(e.max()), "rms": float(e.pow(2).mean().sqrt()), "truth_rms": float(truth.pow(2).mean().sqrt())}


def main():
    scene = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    c = cc.mvsnet_fullsize_case(scene)
    threads = min(32, os.cpu_count() or 1)
    hip = {"exact": hip_stages(c, False), "fast": hip_stages(c, True)}
    f64 = cpu_stages(c, torch.float64, threads)
    f32 = cpu_stages(c, torch.float32, threads)
    table = {"workload": "MVSNet 1600x1184, N=5, D=192, scene %d (tests/config_cases.py)" % scene,
             "note": "cumulative error of the pipeline up to each stage against the float64 evaluation (oracle/torch_ref.py in "
                     "double); float32_cpu = the same composition in float32 on the host (= the reference's arithmetic)", "stages": {}}
    for k in ("feature", "variance", "conv0", "cost", "depth"):
        table["stages"][k] = {"float32_cpu": stats(f32[k], f64[k]), "hip_exact_coordinates": stats(hip["exact"][k], f64[k]),
                              "hip_fast_coordinates": stats(hip["fast"][k], f64[k])}
        print(k, json.dumps(table["stages"][k]), flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", f"error_budget_scene{scene}.json"), "w") as f:
        json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()
