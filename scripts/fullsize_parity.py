#!/usr/bin/env python3
"""Stage-by-stage parity at BASELINE configs[1] size (1600x1184, N=5, D=192):
HIP path on cuda:0 vs the ATen CPU restatement (oracle/torch_ref.py), to see
which stage the end-to-end depth difference comes from.  Test infrastructure
(imports oracle/); run on the GPU box:  python scripts/fullsize_parity.py"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import ops, synth  # noqa: E402
from mvs_amd.models import MVSNet  # noqa: E402
from oracle import torch_ref as tr  # noqa: E402


def main():
    H, W, V, D = (int(x) for x in (sys.argv[1:5] or (1184, 1600, 5, 192)))
    h, w = H // 4, W // 4
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    imgs = torch.from_numpy(synth.images(rng, 1, V, H, W))
    proj = torch.from_numpy(synth.proj_matrices(V, h, w))
    dv = torch.from_numpy(synth.depth_values(D) if D == 192 else
                          synth.depth_values(D, interval=synth.sweep_interval(D)))
    sd = synth.random_state_dict(0)
    model = MVSNet(refine=False)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    torch.set_num_threads(os.cpu_count())
    res = {}
    t0 = time.time()
    with torch.no_grad():
        f_cpu = [tr.feature_net(imgs[:, v], sd) for v in range(V)]
        var_cpu = tr.variance_volume(f_cpu, [proj[:, v] for v in range(V)], dv)
        cost_cpu = tr.cost_reg_net(var_cpu, sd).squeeze(1)
        d_cpu, c_cpu, _ = tr.regress(cost_cpu, dv)
    res["cpu_seconds"] = round(time.time() - t0, 1)

    def mx(a, b):
        return float((a.cpu() - b).abs().max())

    with torch.no_grad():
        gi, gp, gd = imgs.to(dev), proj.to(dev), dv.to(dev)
        f_gpu = [model.feature(gi[:, v]) for v in range(V)]
        res["feature_maxabs"] = max(mx(a, b) for a, b in zip(f_gpu, f_cpu))
        res["feature_scale"] = float(f_cpu[0].abs().max())
        rts = torch.stack([ops.rot_trans(gp[:, v], gp[:, 0]) for v in range(1, V)])
        # (b) variance from the CPU features: isolates the sweep kernel
        cl = [ops.nchw_to_nhwc(f.to(dev)) for f in f_cpu]
        var_b = ops.costvol_variance_cl(cl[0], torch.stack(cl[1:]), rts, gd)
        vb = var_b.permute(0, 4, 1, 2, 3)
        res["variance_from_cpu_features_maxabs"] = mx(vb, var_cpu)
        res["variance_exact_fraction"] = float((vb.cpu() == var_cpu).float().mean())
        res["variance_scale"] = float(var_cpu.abs().max())
        # (c) CostRegNet from the CPU variance: isolates the conv kernels
        cost_c = model.cost_regularization.forward_hip(ops.nchw_to_nhwc(var_cpu.to(dev)))
        res["cost_from_cpu_variance_maxabs"] = mx(cost_c, cost_cpu)
        res["cost_scale"] = float(cost_cpu.abs().max())
        dc, cc, _ = ops.softmax_regress_conf(cost_c, gd)
        res["depth_from_cpu_variance_maxabs_mm"] = mx(dc, d_cpu)
        # (d) regression from the CPU cost volume: isolates softmax/expectation
        dd, cd, _ = ops.softmax_regress_conf(cost_cpu.to(dev), gd)
        res["depth_from_cpu_cost_maxabs_mm"] = mx(dd, d_cpu)
        res["conf_from_cpu_cost_maxabs"] = mx(cd, c_cpu)
        # fp64 ground truth of the regression stage: whose rounding is it?
        p64 = torch.softmax(cost_cpu.double(), 1)
        d64 = (p64 * dv.double().reshape(1, D, 1, 1)).sum(1)
        res["regress_hip_vs_fp64_mm"] = float((dd.cpu().double() - d64).abs().max())
        res["regress_cpuref_vs_fp64_mm"] = float((d_cpu.double() - d64).abs().max())
        # (e) end to end
        out = model(gi, gp, gd)
        res["e2e_depth_maxabs_mm"] = mx(out["depth"], d_cpu)
        res["e2e_conf_maxabs"] = mx(out["photometric_confidence"], c_cpu)
        # (f) GPU features -> rest on CPU: what the feature difference alone does
        var_f = tr.variance_volume([f.cpu() for f in f_gpu], [proj[:, v] for v in range(V)], dv)
        d_f, _, _ = tr.regress(tr.cost_reg_net(var_f, sd).squeeze(1), dv)
        res["cpu_pipeline_on_gpu_features_vs_cpu_maxabs_mm"] = float((d_f - d_cpu).abs().max())
        res["e2e_vs_cpu_pipeline_on_gpu_features_mm"] = mx(out["depth"], d_f)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
