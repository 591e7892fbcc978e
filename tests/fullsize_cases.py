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


def depth_report(d, ref):
    e = (d - ref).abs()
    return {"maxabs_mm": float(e.max()), "p999_mm": float(e.flatten().kthvalue(max(1, int(e.numel() * 0.999))).values)}


def run_mvsnet(fast):
    from mvs_amd import ops, synth
    from mvs_amd.models import MVSNet
    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(GOLDEN, "g12_mvsnet_fullsize.npz")))
    H, W, V, D = (int(x) for x in g["shape"])
    imgs = torch.from_numpy(synth.images(np.random.default_rng(0), 1, V, H, W)).to(dev)
    proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D)).to(dev)
    model = MVSNet(refine=False)
    model.load_state_dict(synth.random_state_dict(0))
    model = model.to(dev).eval()
    model.variance_fast = fast
    with ProbCapture(ops) as cap:
        out = model(imgs, proj, dv)
    res = depth_report(out["depth"], torch.from_numpy(g["depth"]).to(dev))
    res["conf"] = conf_report(cap.calls[-1][0], cap.calls[-1][1], out["photometric_confidence"],
                              torch.from_numpy(g["confidence"]).to(dev))
    return res


def run_cas():
    from mvs_amd import ops, synth
    from mvs_amd.models.cas_mvsnet import CascadeMVSNet
    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(GOLDEN, "g13_cas_fullsize.npz")))
    H, W, V = (int(x) for x in g["shape"])
    imgs = torch.from_numpy(synth.images(np.random.default_rng(0), 1, V, H, W)).to(dev)
    projs = {f"stage{s + 1}": torch.from_numpy(synth.cas_proj_matrices(V, H // sc, W // sc)).to(dev)
             for s, sc in enumerate((4, 2, 1))}
    dv = torch.from_numpy(synth.depth_values(192)).to(dev)
    net = CascadeMVSNet()
    net.load_state_dict(synth.cas_random_state_dict(0))
    net.eval().to(dev)
    with ProbCapture(ops) as cap:
        out = net(imgs, projs, dv)
    res = {}
    for i, s in enumerate(("stage1", "stage2", "stage3")):
        sub = 2 if s == "stage3" else 1
        d, c = out[s]["depth"][:, ::sub, ::sub], out[s]["photometric_confidence"][:, ::sub, ::sub]
        res[s] = depth_report(d, torch.from_numpy(g[s + "_depth"]).to(dev))
        res[s]["conf"] = conf_report(cap.calls[i][0], cap.calls[i][1], c, torch.from_numpy(g[s + "_conf"]).to(dev), sub=sub)
    return res


def run_cvp():
    from mvs_amd import ops, synth
    from mvs_amd.models.cvp_mvsnet import network
    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(GOLDEN, "g14_cvp_fullsize.npz")))
    H, W, nsrc, nscale = (int(x) for x in g["shape"])
    imgs = torch.from_numpy(synth.images(np.random.default_rng(0), 1, nsrc + 1, H, W)).to(dev)
    cams = {k: torch.from_numpy(v).to(dev) for k, v in synth.cvp_cameras(nsrc, H, W).items()}
    net = network(types.SimpleNamespace(nscale=nscale, nsrc=nsrc, mode="test"))
    net.load_state_dict(synth.cvp_random_state_dict(0))
    net.eval().to(dev)
    with ProbCapture(ops) as cap:
        out = net(imgs[:, 0], imgs[:, 1:], cams["ref_in"], cams["src_in"], cams["ref_ex"], cams["src_ex"],
                  cams["depth_min"], cams["depth_max"])
    res = {}
    for i, d in enumerate(out["depth_est_list"]):
        ref = torch.from_numpy(g[f"depth_level{i}"]).to(dev)
        sub = 2 if d.shape[-1] > 1000 else 1
        res[f"level{i}"] = depth_report(d[:, ::sub, ::sub], ref)
    c = out["prob_confidence"]
    sub = 2 if c.shape[-1] > 1000 else 1
    cref = torch.from_numpy(g["prob_confidence"]).to(dev).reshape(c[..., ::sub, ::sub].shape)
    res["conf"] = conf_report(cap.calls[-1][0], cap.calls[-1][1], c[..., ::sub, ::sub], cref, sub=sub)
    return res
