#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE's own Python
(/root/reference/MVSNet/models, CasMVSNet/models, CVP-MVSNet/models) on CPU.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

Outputs are data only (inputs + the reference's outputs) as .npz files next to
this script.  Tests load them; nothing at test/bench time reads /root/reference.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"

from mvs_amd import synth  # noqa: E402
from oracle import torch_ref  # noqa: E402

warnings.filterwarnings("ignore")
torch.set_num_threads(8)


def _import_ref(subdir, stubs=()):
    """Import `models` from one of the reference's script trees in isolation."""
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    for s in stubs:
        if s not in sys.modules:
            sys.modules[s] = types.ModuleType(s)
    sys.path.insert(0, os.path.join(REF, subdir))
    try:
        import models  # noqa: F401
        mods = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")}
    finally:
        sys.path.pop(0)
    return mods


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    mods = _import_ref("MVSNet")
    ref_module = mods["models.module"]
    ref_mvsnet = mods["models.mvsnet"]
    rng = np.random.default_rng(20260928)

    # ---- weights (one copy, shared by every case) ----------------------
    sd = torch_ref.random_state_dict(seed=0)
    save("weights_seed0", **{k: v for k, v in sd.items()})
    model = ref_mvsnet.MVSNet(refine=False)
    model.load_state_dict(sd, strict=True)
    model.eval()

    # ---- G1: homo_warping, shared depth planes -------------------------
    B, C, D, H, W = 2, 8, 6, 12, 16
    proj = synth.proj_matrices(3, H, W, batch=B)
    proj[1, 1, :3, 3] += np.array([9.0, -4.0, 0.0], np.float32) * 40  # push samples out of view
    dv = synth.depth_values(D, batch=B, interval=synth.sweep_interval(D))
    dv[1] += 13.0
    src = synth.smooth_features(rng, (B, C, H, W))
    outs = {}
    for v in (1, 2):
        outs[f"warped_v{v}"] = ref_module.homo_warping(T(src), T(proj[:, v]), T(proj[:, 0]), T(dv))
    save("g1_warp", src=src, proj=proj, depth=dv, **outs)

    # ---- G2: known-answer warps ----------------------------------------
    # identical projections: the sweep degenerates to the identity map in
    # pixel space, i.e. a resample at ix = x*W/(W-1) - 0.5 under the
    # align_corners=False default the reference call runs with.
    ident = ref_module.homo_warping(T(src), T(proj[:, 0]), T(proj[:, 0]), T(dv))
    save("g2_identity", src=src, proj=proj[:, :1], depth=dv, warped=ident)

    # ---- G3/G4/G6: end-to-end forwards with every stage captured -------
    def run_e2e(name, h, w, nv, nd, batch=1, light=False):
        fh, fw = h // 4, w // 4
        imgs = synth.images(rng, batch, nv, h, w)
        pm = synth.proj_matrices(nv, fh, fw, batch=batch)
        dvs = synth.depth_values(nd, batch=batch, interval=synth.sweep_interval(nd))
        cap = {}
        hooks = []
        cr = model.cost_regularization
        hooks.append(cr.register_forward_hook(
            lambda m, i, o: cap.update(variance=i[0].detach().clone(), cost=o.detach().clone())))
        hooks.append(model.feature.register_forward_hook(
            lambda m, i, o: cap.setdefault("features", []).append(o.detach().clone())))
        for lname in ("conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7",
                      "conv9", "conv11"):
            hooks.append(getattr(cr, lname).register_forward_hook(
                lambda m, i, o, lname=lname: cap.update({"act_" + lname: o.detach().clone()})))
        with torch.no_grad():
            out = model(T(imgs), T(pm), T(dvs))
        for hk in hooks:
            hk.remove()
        feats = torch.stack(cap.pop("features"), 1)  # [B,V,32,h/4,w/4]
        if light:  # keep the fixture small: inputs, features, cost, outputs only
            cap = {"cost": cap["cost"]}
        save(name, imgs=imgs, proj=pm, depth_values=dvs, features=feats,
             depth=out["depth"], confidence=out["photometric_confidence"], **cap)
        return imgs, pm, dvs

    run_e2e("g6_e2e_64x96_v3_d8", 64, 96, 3, 8)
    run_e2e("g6_e2e_128x160_v3_d16", 128, 160, 3, 16, light=True)
    run_e2e("g3_e2e_64x64_v2_d8", 64, 64, 2, 8)
    run_e2e("g3_e2e_64x64_v5_d8_b2", 64, 64, 5, 8, batch=2)

    # ---- G5: softmax / regression / confidence edge cases ---------------
    class _Fixed(torch.nn.Module):
        def __init__(self, vol):
            super().__init__()
            self.vol = vol

        def forward(self, x):
            return self.vol

    def run_regress(name, cost, dvs):
        h4, w4 = cost.shape[2], cost.shape[3]
        keep = model.cost_regularization
        model.cost_regularization = _Fixed(T(cost).unsqueeze(1))
        nv = 2
        imgs = synth.images(rng, cost.shape[0], nv, h4 * 4, w4 * 4)
        pm = synth.proj_matrices(nv, h4, w4, batch=cost.shape[0])
        with torch.no_grad():
            out = model(T(imgs), T(pm), T(dvs))
        model.cost_regularization = keep
        save(name, cost=cost, depth_values=dvs, depth=out["depth"],
             confidence=out["photometric_confidence"])

    Bq, Dq, Hq, Wq = 2, 8, 8, 8
    cost = rng.standard_normal((Bq, Dq, Hq, Wq)).astype(np.float32) * 3
    cost[0, 0, 0, :] += 40.0        # all mass on plane 0   (pad edge, idx-1 < 0)
    cost[0, Dq - 1, 1, :] += 40.0   # all mass on the last plane (idx+1, idx+2 >= D)
    cost[1, Dq - 2, 2, :] += 40.0
    cost[1, :, 3, :] = 0.0          # flat distribution
    run_regress("g5_regress_d8", cost,
                synth.depth_values(Dq, batch=Bq, interval=synth.sweep_interval(Dq)))
    cost = rng.standard_normal((1, 192, 8, 8)).astype(np.float32) * 4
    cost[0, 0, 0, :] += 60.0
    cost[0, 191, 1, :] += 60.0
    run_regress("g5_regress_d192", cost, synth.depth_values(192, batch=1))

    # ---- G7: training step (train-mode BN, loss, gradients) -------------
    torch.manual_seed(1)
    model.train()
    h, w, nv, nd = 64, 96, 3, 8
    imgs = synth.images(rng, 2, nv, h, w)
    pm = synth.proj_matrices(nv, h // 4, w // 4, batch=2)
    dvs = synth.depth_values(nd, batch=2, interval=synth.sweep_interval(nd))
    gt = (synth.DTU_TARGET_Z + 20 * rng.standard_normal((2, h // 4, w // 4))).astype(np.float32)
    mask = (rng.random((2, h // 4, w // 4)) > 0.2).astype(np.float32)
    model.zero_grad()
    out = model(T(imgs), T(pm), T(dvs))
    loss = ref_mvsnet.mvsnet_loss(out["depth"], T(gt), T(mask))
    loss.backward()
    grads = {"grad__" + k: p.grad for k, p in model.named_parameters()
             if k in ("feature.conv0.conv.weight", "feature.feature.bias",
                      "cost_regularization.conv0.conv.weight",
                      "cost_regularization.conv7.0.weight",
                      "cost_regularization.conv11.1.weight",
                      "cost_regularization.prob.weight", "cost_regularization.prob.bias")}
    save("g7_train_step", imgs=imgs, proj=pm, depth_values=dvs, gt=gt, mask=mask,
         depth=out["depth"], loss=loss.detach(), **grads)
    model.load_state_dict(sd, strict=True)  # undo running-stat updates
    model.eval()

    # gradient of homo_warping alone w.r.t. the source feature map
    Bw, Cw, Dw, Hw, Ww = 1, 4, 5, 10, 12
    pw = synth.proj_matrices(2, Hw, Ww, batch=Bw)
    dw = synth.depth_values(Dw, batch=Bw, interval=synth.sweep_interval(Dw))
    sw = T(synth.smooth_features(rng, (Bw, Cw, Hw, Ww))).requires_grad_(True)
    gout = rng.standard_normal((Bw, Cw, Dw, Hw, Ww)).astype(np.float32)
    wv = ref_module.homo_warping(sw, T(pw[:, 1]), T(pw[:, 0]), T(dw))
    wv.backward(T(gout))
    save("g7_warp_grad", src=sw.detach(), proj=pw, depth=dw, grad_out=gout, warped=wv.detach(),
         grad_src=sw.grad)

    # variance-volume gradient (train branch of mvsnet.py:159-161) w.r.t. all features
    Vv = 3
    pv = synth.proj_matrices(Vv, Hw, Ww, batch=Bw)
    feats = [T(synth.smooth_features(rng, (Bw, Cw, Hw, Ww))).requires_grad_(True) for _ in range(Vv)]
    refv = feats[0].unsqueeze(2).repeat(1, 1, Dw, 1, 1)
    s, q = refv, refv ** 2
    for i in range(1, Vv):
        wv = ref_module.homo_warping(feats[i], T(pv[:, i]), T(pv[:, 0]), T(dw))
        s = s + wv
        q = q + wv ** 2
    var = q.div(Vv).sub(s.div(Vv).pow(2))
    var.backward(T(gout))
    save("g7_variance_grad", feats=torch.stack([f.detach() for f in feats]), proj=pv, depth=dw,
         grad_out=gout, variance=var.detach(), grad_feats=torch.stack([f.grad for f in feats]))

    # ---- G8: CasMVSNet per-pixel hypotheses; CVP alias quirk ------------
    cmods = _import_ref("CasMVSNet", stubs=("torchvision", "torchvision.utils", "cv2"))
    cas_module = cmods["models.module"]
    B, C, D, H, W = 1, 8, 4, 12, 16
    proj = synth.proj_matrices(3, H, W, batch=B)
    base = synth.depth_values(D, batch=B, interval=synth.sweep_interval(D))
    pp = (base[:, :, None, None] + 15 * rng.standard_normal((B, D, H, W))).astype(np.float32)
    src = synth.smooth_features(rng, (B, C, H, W))
    wpp = cas_module.homo_warping(T(src), T(proj[:, 1]), T(proj[:, 0]), T(pp))
    cost = rng.standard_normal((B, D, H, W)).astype(np.float32) * 2
    prob = torch.softmax(T(cost), 1)
    dpp = cas_module.depth_regression(prob, T(pp))
    save("g8_cas_perpixel", src=src, proj=proj, depth=pp, warped=wpp, cost=cost, regressed=dpp)

    # Cas DepthNet stage (cas_mvsnet.py:12-66): per-pixel depth + clamp
    cas_net = cmods["models.cas_mvsnet"]
    dn = cas_net.DepthNet()
    creg = cas_module.CostRegNet(in_channels=8, base_channels=8)
    torch.manual_seed(3)
    for p in creg.parameters():
        torch.nn.init.normal_(p, 0, 0.15)
    creg.eval()
    Hc, Wc, Dc = 16, 24, 8
    cproj = synth.cas_proj_matrices(3, Hc, Wc, batch=1)
    feats = [T(synth.smooth_features(rng, (1, 8, Hc, Wc))) for _ in range(3)]
    basec = synth.depth_values(Dc, batch=1, interval=synth.sweep_interval(Dc))
    ppc = (basec[:, :, None, None] + 6 * rng.standard_normal((1, Dc, Hc, Wc))).astype(np.float32)
    capc = {}
    hk = creg.register_forward_hook(lambda m, i, o: capc.update(variance=i[0].clone(), cost=o.clone()))
    dn.eval()
    with torch.no_grad():
        oc = dn(feats, T(cproj), T(ppc), Dc, creg)
    hk.remove()
    save("g8_cas_depthnet", feats=torch.stack(feats), cas_proj=cproj, depth=ppc,
         variance=capc["variance"], cost=capc["cost"], out_depth=oc["depth"],
         out_conf=oc["photometric_confidence"],
         **{"creg__" + k: v for k, v in creg.state_dict().items()})

    # CVP alias quirk (CVP-MVSNet/models/modules.py:228-229): restated from the
    # reference lines with the reference's own homo_warping (the CVP module
    # hard-codes .cuda(); its warp arithmetic is identical to module.py:46-87)
    Vq = 3
    pq = synth.proj_matrices(Vq, H, W, batch=1)
    fq = [T(synth.smooth_features(rng, (1, C, H, W))) for _ in range(Vq)]
    vs = fq[0].unsqueeze(2).repeat(1, 1, D, 1, 1)
    vq2 = vs.pow_(2)  # aliases vs, exactly as modules.py:228-229
    for i in range(1, Vq):
        wq = cas_module.homo_warping(fq[i], T(pq[:, i]), T(pq[:, 0]), T(pp))
        vs = vs + wq
        vq2 = vq2 + wq.pow_(2)
    cq = vq2.div_(Vq).sub_(vs.div_(Vq).pow_(2))
    save("g8_cvp_alias", feats=torch.stack(fq), proj=pq, depth=pp, variance=cq)

    print("done")


if __name__ == "__main__":
    main()
