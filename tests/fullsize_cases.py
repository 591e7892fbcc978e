"""The three full-size cases of BASELINE.json run through the HIP path and compared with the
reference's own CPU outputs (fixtures g12..g14).  Shared by tests/test_gpu_fullsize_reference.py
and scripts/fullsize_reference_parity.py."""
import os
import types

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class ProbCapture:
    """Wraps ops.softmax_regress_conf: keeps the probability volume of every call so that a
    confidence mismatch can be traced to a flip of the truncated index (mvsnet.py:189-191)."""

    def __init__(self, ops):
        self.ops, self.orig, self.calls = ops, ops.softmax_regress_conf, []

    def __enter__(self):
        def wrapped(cost, dv, clamp_idx=False, want_prob=False):
            d, c, p = self.orig(cost, dv, clamp_idx, True)
            self.calls.append((p, bool(clamp_idx)))
            return d, c, (p if want_prob else None)
        self.ops.softmax_regress_conf = wrapped
        return self

    def __exit__(self, *exc):
        self.ops.softmax_regress_conf = self.orig
        return False


def conf_report(prob, clamp, conf, conf_ref, tol=2e-4, sub=1):
    """Compare photometric confidences.  conf = sum of the four probabilities around
    idx = trunc(sum_d p_d d) (mvsnet.py:187-191): where that expectation lies within rounding of
    an integer, the two implementations may truncate to different indices and the confidence
    jumps by a whole probability -- legitimate, and explicitly accounted for here: a mismatch
    is `explained` when the reference's value equals the window sum at the neighbouring index."""
    D = prob.shape[1]
    dev = prob.device
    p = prob[:, :, ::sub, ::sub]
    bad = (conf - conf_ref).abs() > tol
    out = {"maxabs": float((conf - conf_ref).abs().max()), "mismatches": int(bad.sum()), "pixels": conf.numel()}
    if out["mismatches"] == 0:
        out["unexplained"] = 0
        return out
    idx_f = (p.double() * torch.arange(D, dtype=torch.float64, device=dev).view(1, D, 1, 1)).sum(1)
    pad = torch.nn.functional.pad(p, (0, 0, 0, 0, 1, 2))                       # zeros at d = -1, D, D+1
    s4 = pad[:, 0:D] + pad[:, 1:D + 1] + pad[:, 2:D + 2] + pad[:, 3:D + 3]     # p[d-1] + p[d] + p[d+1] + p[d+2]
    r = idx_f.round()
    near = (idx_f - r).abs() < 2e-3
    lo = (r.long() - 1).clamp(0, D - 1)
    hi = r.long().clamp(0, D - 1)
    c_lo = s4.gather(1, lo[:, None])[:, 0]
    c_hi = s4.gather(1, hi[:, None])[:, 0]
    explained = near & (torch.minimum((c_lo - conf_ref).abs(), (c_hi - conf_ref).abs()) < tol)
    out["unexplained"] = int((bad & ~explained).sum())
    out["maxabs_without_flips"] = float(((conf - conf_ref).abs() * (~bad)).max())
    return out


def depth_report(d, ref, truth=None):
    """max |d - ref| (the gate); with `truth` (the float64 evaluation of the same composition, fixtures *64 keys)
    also the error budget: how far the HIP result and the reference's float32 result each are from it."""
    e = (d - ref).abs()
    out = {"maxabs_mm": float(e.max()), "p999_mm": float(e.flatten().kthvalue(max(1, int(e.numel() * 0.999))).values)}
    if truth is not None:
        eh, er = (d.double() - truth).abs(), (ref.double() - truth).abs()
        out.update(hip_vs_f64_mm=float(eh.max()), ref_vs_f64_mm=float(er.max()),
                   hip_vs_f64_rms=float(eh.pow(2).mean().sqrt()), ref_vs_f64_rms=float(er.pow(2).mean().sqrt()))
    return out


def _dev(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _truth(t, key, ref32, dev):
    """float64 answer stored as int16 steps from the reference's float32 map (make_golden_configs._delta16)."""
    return _dev(np.asarray(ref32, dtype=np.float64) + t[key + "_d16"].astype(np.float64) * float(t["delta_step_mm"]), dev)


def run_mvsnet_case(case, gold, fast, truth_file=None):
    """One MVSNet eval forward on a config_cases recipe against fixture `gold` (depth / confidence from the
    reference; depth64 from the float64 evaluation, possibly in a separate fixture)."""
    from mvs_amd import ops
    from mvs_amd.models import MVSNet
    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(GOLDEN, gold + ".npz")))
    t = dict(np.load(os.path.join(GOLDEN, truth_file + ".npz"))) if truth_file else g
    model = MVSNet(refine=False)
    model.load_state_dict(case["sd"])
    model = model.to(dev).eval()
    model.variance_fast = fast
    with ProbCapture(ops) as cap:
        out = model(_dev(case["imgs"], dev), _dev(case["proj"], dev), _dev(case["depth_values"], dev))
    res = depth_report(out["depth"], _dev(g["depth"], dev), _dev(t["depth64"], dev) if "depth64" in t else None)
    res["conf"] = conf_report(cap.calls[-1][0], cap.calls[-1][1], out["photometric_confidence"], _dev(g["confidence"], dev))
    return res


def run_mvsnet(fast, scene=0):
    import config_cases as cc
    if scene == 0:
        return run_mvsnet_case(cc.mvsnet_fullsize_case(0), "g12_mvsnet_fullsize", fast, "g20_mvsnet_fullsize_fp64")
    return run_mvsnet_case(cc.mvsnet_fullsize_case(scene), f"g19_mvsnet_fullsize_scene{scene}", fast)


def run_eval_small(fast):
    """BASELINE configs[0]: 640x512, N=3, D=48."""
    import config_cases as cc
    return run_mvsnet_case(cc.eval_small_case(), "g18_eval_640x512_v3_d48", fast)


def run_train_step():
    """BASELINE configs[4], one GPU's share of a step (640x512, V=3, D=192, B=1): forward(train) -> mvsnet_loss ->
    backward on the HIP training path against the reference's own step (g17: depth, loss, every parameter gradient,
    BatchNorm running statistics) and against the float64 evaluation of the same step."""
    import config_cases as cc
    from mvs_amd.models import MVSNet, mvsnet_loss
    dev = torch.device("cuda:0")
    c = cc.train_case()
    g = dict(np.load(os.path.join(GOLDEN, "g17_train_640x512_v3_d192.npz")))
    model = MVSNet(refine=False)
    model.load_state_dict(c["sd"])
    model = model.to(dev).train()
    out = model(_dev(c["imgs"], dev), _dev(c["proj"], dev), _dev(c["depth_values"], dev))
    loss = mvsnet_loss(out["depth"], _dev(c["gt"], dev), _dev(c["mask"], dev))
    loss.backward()
    params = dict(model.named_parameters())
    names = [str(k) for k in g["grad_names"]]
    got = torch.cat([params[k].grad.reshape(-1) for k in names]).double().cpu()
    ref, truth = torch.from_numpy(g["grads"]).double(), torch.from_numpy(g["grads64"])
    res = {"depth": depth_report(out["depth"].detach(), _dev(g["depth"], dev), _dev(g["depth64"], dev)),
           "loss": float(loss.detach()), "loss_ref": float(g["loss"]), "loss64": float(g["loss64"]), "grads": {}, "stats": {}}
    off = 0
    gmax = float(truth.abs().max())
    res["grads_all"] = {"hip_vs_f64_rms": float((got - truth).pow(2).mean().sqrt()), "ref_vs_f64_rms": float((ref - truth).pow(2).mean().sqrt()),
                        "hip_vs_ref_rms": float((got - ref).pow(2).mean().sqrt()), "truth_rms": float(truth.pow(2).mean().sqrt())}
    for k, nel in zip(names, g["grad_sizes"]):
        a, r, t = got[off:off + nel], ref[off:off + nel], truth[off:off + nel]
        off += int(nel)
        # relative to the tensor's largest true gradient (prob.bias: the softmax is shift-invariant, its true
        # gradient is ~0 -- measured against the largest gradient of the network instead)
        scale = max(float(t.abs().max()), 1e-6 * gmax)
        res["grads"][k] = {"hip_vs_f64": float((a - t).abs().max()) / scale, "ref_vs_f64": float((r - t).abs().max()) / scale,
                           "hip_vs_ref": float((a - r).abs().max()) / scale}
    sd = model.state_dict()
    for k in g:
        if k.startswith("stat__"):
            res["stats"][k[6:]] = float((sd[k[6:]].cpu() - torch.from_numpy(g[k])).abs().max())
    return res


def run_cas(scene=0):
    import config_cases as cc
    from mvs_amd import ops
    from mvs_amd.models.cas_mvsnet import CascadeMVSNet
    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(GOLDEN, "g13_cas_fullsize.npz" if scene == 0 else f"g21_cas_fullsize_scene{scene}.npz")))
    t = dict(np.load(os.path.join(GOLDEN, "g23_cas_fullsize_fp64.npz")))
    c = cc.cas_fullsize_case(scene)
    net = CascadeMVSNet()
    net.load_state_dict(c["sd"])
    net.eval().to(dev)
    with ProbCapture(ops) as cap:
        out = net(_dev(c["imgs"], dev), {k: _dev(v, dev) for k, v in c["proj"].items()}, _dev(c["depth_values"], dev))
    res = {}
    for i, s in enumerate(("stage1", "stage2", "stage3")):
        sub = 2 if s == "stage3" else 1
        d, cf = out[s]["depth"][:, ::sub, ::sub], out[s]["photometric_confidence"][:, ::sub, ::sub]
        res[s] = depth_report(d, _dev(g[s + "_depth"], dev), _truth(t, f"s{scene}_{s}_depth64", g[s + "_depth"], dev))
        res[s]["conf"] = conf_report(cap.calls[i][0], cap.calls[i][1], cf, _dev(g[s + "_conf"], dev), sub=sub)
    return res


def run_cvp(scene=0):
    import config_cases as cc
    from mvs_amd import ops
    from mvs_amd.models.cvp_mvsnet import network
    dev = torch.device("cuda:0")
    # scene 2 (round 4): reference outputs and float64 answers in one file
    g = dict(np.load(os.path.join(GOLDEN, {0: "g14_cvp_fullsize.npz", 1: "g22_cvp_fullsize_scene1.npz", 2: "g25_cvp_fullsize_scene2.npz"}[scene])))
    t = g if scene == 2 else dict(np.load(os.path.join(GOLDEN, "g24_cvp_fullsize_fp64.npz")))
    c = cc.cvp_fullsize_case(scene)
    net = network(types.SimpleNamespace(nscale=c["nscale"], nsrc=c["nsrc"], mode="test"))
    net.load_state_dict(c["sd"])
    net.eval().to(dev)
    imgs = _dev(c["imgs"], dev)
    cams = {k: _dev(v, dev) for k, v in c["cams"].items()}
    with ProbCapture(ops) as cap:
        out = net(imgs[:, 0], imgs[:, 1:], cams["ref_in"], cams["src_in"], cams["ref_ex"], cams["src_ex"],
                  cams["depth_min"], cams["depth_max"])
    res = {}
    for i, d in enumerate(out["depth_est_list"]):
        ref = _dev(g[f"depth_level{i}"], dev)
        sub = 2 if d.shape[-1] > 1000 else 1
        res[f"level{i}"] = depth_report(d[:, ::sub, ::sub], ref, _truth(t, f"s{scene}_depth_level{i}_64", g[f"depth_level{i}"], dev))
    cf = out["prob_confidence"]
    sub = 2 if cf.shape[-1] > 1000 else 1
    cref = _dev(g["prob_confidence"], dev).reshape(cf[..., ::sub, ::sub].shape)
    res["conf"] = conf_report(cap.calls[-1][0], cap.calls[-1][1], cf[..., ::sub, ::sub], cref, sub=sub)
    return res
