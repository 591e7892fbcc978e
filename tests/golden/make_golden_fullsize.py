"""Full-size golden vectors: the REFERENCE's own CPU forward at the sizes BASELINE.json names.

Run in the build container (needs /root/reference):
    python tests/golden/make_golden_fullsize.py [mvsnet] [cas] [cvp]

  g12_mvsnet_fullsize   configs[1]: MVSNet/models (mvsnet.py:124-198), 1600x1184, 5 views, D=192
  g13_cas_fullsize      configs[2]: CasMVSNet/models CascadeMVSNet (cas_mvsnet.py:109-164), 1600x1184,
                        5 views, 48/32/8 hypotheses
  g14_cvp_fullsize      configs[3]: CVP-MVSNet/models network (net.py:99-209), 1920x1056, 7 views, 5 levels

Inputs are the seeded synthetic recipe of mvs_amd.synth (numpy default_rng(0) images, arc cameras,
DTU depth sweep; weights synth.*_random_state_dict(0)), i.e. exactly what bench.py /
scripts/bench_cascade.py / scripts/bench_cvp.py feed the HIP path -- so only OUTPUTS are stored:
depth and confidence maps (full resolution up to 592x800; the 1184x1600 / 1056x1920 maps on the
even-row, even-column grid to keep the fixtures at a couple of MB).  Only data is stored.
"""
import os
import pdb
import sys
import time
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import T, _import_ref, save, synth  # noqa: E402


def mvsnet():
    mods = _import_ref("MVSNet")
    net = mods["models.mvsnet"].MVSNet(refine=False)
    net.load_state_dict(synth.random_state_dict(0))
    net.eval()
    H, W, V, D = 1184, 1600, 5, 192
    rng = np.random.default_rng(0)
    imgs = synth.images(rng, 1, V, H, W)
    proj = synth.proj_matrices(V, H // 4, W // 4)
    dv = synth.depth_values(D)
    t0 = time.time()
    with torch.no_grad():
        out = net(T(imgs), T(proj), T(dv))
    print("mvsnet reference forward", round(time.time() - t0, 1), "s")
    save("g12_mvsnet_fullsize", depth=out["depth"], confidence=out["photometric_confidence"],
         shape=np.array([H, W, V, D], dtype=np.int64), seed=np.int64(0))


def cas():
    mods = _import_ref("CasMVSNet", stubs=("torchvision", "torchvision.utils", "cv2"))
    net = mods["models.cas_mvsnet"].CascadeMVSNet(refine=False)
    net.load_state_dict(synth.cas_random_state_dict(0))
    net.eval()
    H, W, V = 1184, 1600, 5
    rng = np.random.default_rng(0)
    imgs = synth.images(rng, 1, V, H, W)
    proj = {f"stage{s + 1}": synth.cas_proj_matrices(V, H // sc, W // sc) for s, sc in enumerate((4, 2, 1))}
    dv = synth.depth_values(192)
    t0 = time.time()
    with torch.no_grad():
        out = net(T(imgs), {k: T(v) for k, v in proj.items()}, T(dv))
    print("cascade reference forward", round(time.time() - t0, 1), "s")
    arrs = dict(shape=np.array([H, W, V], dtype=np.int64), seed=np.int64(0))
    for s in ("stage1", "stage2", "stage3"):
        d, c = out[s]["depth"], out[s]["photometric_confidence"]
        if s == "stage3":
            d, c = d[:, ::2, ::2], c[:, ::2, ::2]
        arrs[s + "_depth"] = d.contiguous()
        arrs[s + "_conf"] = c.contiguous()
    save("g13_cas_fullsize", **arrs)


def cvp():
    warnings.filterwarnings("ignore")
    for s in ("torchvision", "torchvision.utils", "cv2"):
        sys.modules.setdefault(s, types.ModuleType(s))
    for k in [k for k in sys.modules if k in ("models", "utils") or k.startswith("models.")]:
        del sys.modules[k]
    torch.Tensor.cuda = lambda self, *a, **k: self      # the reference hard-codes .cuda()
    pdb.set_trace = lambda *a, **k: None                # ... and a breakpoint in its forward
    sys.path.insert(0, "/root/reference/CVP-MVSNet")
    from models import net as refnet
    sys.path.pop(0)
    np.seterr(all="warn")
    H, W, nsrc, nscale = 1056, 1920, 6, 5
    net = refnet.network(types.SimpleNamespace(nscale=nscale, nsrc=nsrc, mode="test"))
    net.load_state_dict(synth.cvp_random_state_dict(0))
    net.eval()
    rng = np.random.default_rng(0)
    imgs = synth.images(rng, 1, nsrc + 1, H, W)
    cams = synth.cvp_cameras(nsrc, H, W)
    t0 = time.time()
    with torch.no_grad():
        out = net(T(imgs[:, 0]), T(imgs[:, 1:]), T(cams["ref_in"]), T(cams["src_in"]), T(cams["ref_ex"]),
                  T(cams["src_ex"]), T(cams["depth_min"]), T(cams["depth_max"]))
    print("cvp reference forward", round(time.time() - t0, 1), "s")
    arrs = dict(shape=np.array([H, W, nsrc, nscale], dtype=np.int64), seed=np.int64(0))
    for i, d in enumerate(out["depth_est_list"]):
        arrs[f"depth_level{i}"] = (d[:, ::2, ::2] if d.shape[-1] > 1000 else d).contiguous()
    c = out["prob_confidence"]
    arrs["prob_confidence"] = (c[..., ::2, ::2] if c.shape[-1] > 1000 else c).contiguous()
    save("g14_cvp_fullsize", **arrs)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["mvsnet", "cas", "cvp"]
    for w in which:
        {"mvsnet": mvsnet, "cas": cas, "cvp": cvp}[w]()
