"""Golden vectors at the workloads BASELINE.json names that rounds 1-2 had not pinned, a second full-size
scene per model, and float64 "true answers" beside the reference's float32 outputs.

Run in the build container (needs /root/reference):
    python tests/golden/make_golden_configs.py [g17] [g18] [g19] [g20] [g21] [g22]

  g17_train_640x512_v3_d192   configs[4], one GPU's share of a step: the reference MVSNet in train() mode,
                              640x512, V=3, D=192, B=1 (MVSNet/train.py:204-248 up to loss.backward()):
                              depth, loss, EVERY parameter gradient (338,129 floats)
  g18_eval_640x512_v3_d48     configs[0]: MVSNet/eval.py's forward at 640x512, N=3, D=48 (eval.py:96-131)
  g19_mvsnet_fullsize_scene1  configs[1] again on a second scene: image seed 1, weight seed 1, camera rig 1
                              (wider baselines, roll, unequal distances: mvs_amd/synth.py)
  g20_mvsnet_fullsize_fp64    float64 depth of scene 0 (the scene of g12)
  g21_cas_fullsize_scene1     configs[2] on scene 1
  g22_cvp_fullsize_scene1     configs[3] on scene 1
  g23_cas_fullsize_fp64       float64 depths of configs[2], scenes 0 and 1 (all three stages)
  g24_cvp_fullsize_fp64       float64 depths of configs[3], scenes 0 and 1 (all five levels)
  g25_cvp_fullsize_scene2     configs[3] on a third scene (image / weight seed 2, rig 2): reference float32 + float64

Every fixture holds the REFERENCE's float32 outputs (imported from /root/reference, as in
make_golden_fullsize.py) and, under *64 keys, the same composition evaluated in float64 by
oracle/torch_ref.py (the reference itself cannot run in float64: module.py:67-68 hard-codes float32 grids;
torch_ref's float32 mode is pinned to the reference by tests/test_oracle_golden.py and again here:
`port_vs_reference` is printed for every case).  The float64 answer turns the parity gate from "within
1e-3 mm of one float32 evaluation" into "no farther from the true value than the reference is".
Inputs are the seeded recipe of mvs_amd.synth, so only outputs are stored.
"""
import gc
import os
import sys
import time
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import T, _import_ref, save, synth, torch_ref  # noqa: E402

sys.path.insert(0, os.path.dirname(HERE))
import config_cases as cc  # noqa: E402  (tests/config_cases.py: the input recipes, shared with the GPU tests)


DELTA_STEP_MM = 5e-8


def _delta16(d64, ref32):
    """float64 answer as its difference from the reference's float32 map in steps of 5e-8 mm (int16: 2 bytes per pixel
    instead of 8; reconstructed within 2.5e-8 mm -- the differences are ~1e-3 mm at most)."""
    q = np.round((d64.numpy() - np.asarray(ref32, dtype=np.float64)) / DELTA_STEP_MM)
    assert np.abs(q).max() < 32767, np.abs(q).max()
    return q.astype(np.int16)


def _dbl(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def _mvsnet_ref(sd, train=False):
    mods = _import_ref("MVSNet")
    net = mods["models.mvsnet"].MVSNet(refine=False)
    net.load_state_dict(sd)
    net.train(train)
    return net, mods["models.mvsnet"].mvsnet_loss


def g17():
    c = cc.train_case()
    net, ref_loss = _mvsnet_ref(c["sd"], train=True)
    t0 = time.time()
    net.zero_grad()
    out = net(T(c["imgs"]), T(c["proj"]), T(c["depth_values"]))
    loss = ref_loss(out["depth"], T(c["gt"]), T(c["mask"]))
    loss.backward()
    print("g17 reference train step", round(time.time() - t0, 1), "s; loss", float(loss))
    names = [k for k, _ in net.named_parameters()]
    grads = torch.cat([p.grad.reshape(-1) for _, p in net.named_parameters()])
    depth = out["depth"].detach().clone()
    # running statistics after the step (momentum 0.1): pins the fused BatchNorm's buffer update
    stats = {k: v.clone() for k, v in net.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    del net, out
    gc.collect()
    # float64 answer of the same step
    t0 = time.time()
    sd64 = {k: (v.double().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v)
            for k, v in c["sd"].items()}
    o64 = torch_ref.mvsnet_forward(T(c["imgs"]).double(), T(c["proj"]).double(), T(c["depth_values"]).double(), sd64, train=True)
    l64 = torch_ref.masked_smooth_l1(o64["depth"], T(c["gt"]).double(), T(c["mask"]))
    l64.backward()
    print("g17 float64 train step", round(time.time() - t0, 1), "s; loss", float(l64))
    g64 = torch.cat([sd64[k].grad.reshape(-1) for k in names])
    print("  |depth32 - depth64| max", float((depth.double() - o64["depth"].detach()).abs().max()),
          " grads: max|g32-g64| / max|g64| =", float((grads.double() - g64).abs().max() / g64.abs().max()))
    save("g17_train_640x512_v3_d192", depth=depth, loss=loss.detach(), grads=grads,
         grad_names=np.array(names), grad_sizes=np.array([c["sd"][k].numel() for k in names], dtype=np.int64),
         depth64=o64["depth"].detach(), loss64=l64.detach(), grads64=g64,
         **{"stat__" + k: v for k, v in stats.items()
            if k in ("feature.conv0.bn.running_mean", "feature.conv6.bn.running_var",
                     "cost_regularization.conv0.bn.running_var", "cost_regularization.conv11.1.running_mean")})


G27_PARAMS = ("feature.conv0.conv.weight", "cost_regularization.conv6.conv.weight", "cost_regularization.prob.weight")
G27_STATS = ("feature.conv0.bn.running_mean", "feature.conv6.bn.running_var", "cost_regularization.conv0.bn.running_var",
             "cost_regularization.conv11.1.running_mean")


def g27():
    """SURVEY 8(c)'s last row: THREE consecutive steps of the reference's train_sample composition (MVSNet/train.py:98,204-248:
    zero_grad -> forward(train) -> mvsnet_loss -> backward -> Adam(lr 1e-3, betas 0.9 / 0.999, wd 0).step()) on ONE fixed batch
    (configs[4]'s per-GPU workload, tests/config_cases.py::train_case): the three losses, three named parameters and four
    BatchNorm running statistics after step 3 -- and the same trajectory in float64 (oracle/torch_ref.py) beside them."""
    c = cc.train_case()
    net, ref_loss = _mvsnet_ref(c["sd"], train=True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0)
    losses = []
    t0 = time.time()
    for step in range(3):
        opt.zero_grad()
        out = net(T(c["imgs"]), T(c["proj"]), T(c["depth_values"]))
        loss = ref_loss(out["depth"], T(c["gt"]), T(c["mask"]))
        loss.backward()
        opt.step()
        losses.append(float(loss))
        print("g27 reference step", step, "loss", float(loss), round(time.time() - t0, 1), "s", flush=True)
    sd3 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    del net, out, opt
    gc.collect()
    sd64 = {k: (v.double().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else
                (v.double() if v.is_floating_point() else v.clone())) for k, v in c["sd"].items()}
    params = [v for v in sd64.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0)
    losses64 = []
    for step in range(3):
        opt.zero_grad()
        o64 = torch_ref.mvsnet_forward(T(c["imgs"]).double(), T(c["proj"]).double(), T(c["depth_values"]).double(), sd64, train=True)
        l64 = torch_ref.masked_smooth_l1(o64["depth"], T(c["gt"]).double(), T(c["mask"]))
        l64.backward()
        opt.step()
        losses64.append(float(l64))
        print("g27 float64 step", step, "loss", float(l64), round(time.time() - t0, 1), "s", flush=True)
        del o64, l64
        gc.collect()
    print("  losses32", losses, "losses64", losses64)
    for k in G27_PARAMS:
        print("  ", k, "max |p32 - p64| after step 3:", float((sd3[k].double() - sd64[k].detach()).abs().max()),
              " max |p3 - p0|:", float((sd3[k] - c["sd"][k]).abs().max()))
    save("g27_train_3steps", losses=np.array(losses, dtype=np.float64), losses64=np.array(losses64, dtype=np.float64),
         **{"param__" + k: sd3[k] for k in G27_PARAMS}, **{"param64__" + k: sd64[k].detach() for k in G27_PARAMS},
         **{"stat__" + k: sd3[k] for k in G27_STATS})      # (torch_ref's functional BatchNorm keeps no running statistics: float32 only)


def _eval_case(name, c, tag):
    net, _ = _mvsnet_ref(c["sd"])
    t0 = time.time()
    with torch.no_grad():
        out = net(T(c["imgs"]), T(c["proj"]), T(c["depth_values"]))
    print(tag, "reference forward", round(time.time() - t0, 1), "s")
    del net
    gc.collect()
    with torch.no_grad():
        port = torch_ref.mvsnet_forward(T(c["imgs"]), T(c["proj"]), T(c["depth_values"]), c["sd"])
        print("  port_vs_reference (float32 torch_ref vs reference) max", float((port["depth"] - out["depth"]).abs().max()))
        del port
        gc.collect()
        t0 = time.time()
        o64 = torch_ref.mvsnet_forward(T(c["imgs"]).double(), T(c["proj"]).double(), T(c["depth_values"]).double(), _dbl(c["sd"]))
        print(tag, "float64 forward", round(time.time() - t0, 1), "s; max|ref32 - f64| =",
              float((out["depth"].double() - o64["depth"]).abs().max()), "mm")
    return out, o64


def g18():
    c = cc.eval_small_case()
    out, o64 = _eval_case("g18", c, "g18")
    save("g18_eval_640x512_v3_d48", depth=out["depth"], confidence=out["photometric_confidence"],
         depth64=o64["depth"], confidence64=o64["photometric_confidence"])


def g19():
    c = cc.mvsnet_fullsize_case(scene=1)
    out, o64 = _eval_case("g19", c, "g19")
    save("g19_mvsnet_fullsize_scene1", depth=out["depth"], confidence=out["photometric_confidence"],
         depth64=o64["depth"], confidence64=o64["photometric_confidence"])


def g20():
    c = cc.mvsnet_fullsize_case(scene=0)
    with torch.no_grad():
        o64 = torch_ref.mvsnet_forward(T(c["imgs"]).double(), T(c["proj"]).double(), T(c["depth_values"]).double(), _dbl(c["sd"]))
    g12 = dict(np.load(os.path.join(HERE, "g12_mvsnet_fullsize.npz")))
    print("g20: max|g12 depth - f64| =", float(np.abs(g12["depth"].astype(np.float64) - o64["depth"].numpy()).max()), "mm")
    save("g20_mvsnet_fullsize_fp64", depth64=o64["depth"], confidence64=o64["photometric_confidence"])


def g21():
    c = cc.cas_fullsize_case(scene=1)
    mods = _import_ref("CasMVSNet", stubs=("torchvision", "torchvision.utils", "cv2"))
    net = mods["models.cas_mvsnet"].CascadeMVSNet(refine=False)
    net.load_state_dict(c["sd"])
    net.eval()
    t0 = time.time()
    with torch.no_grad():
        out = net(T(c["imgs"]), {k: T(v) for k, v in c["proj"].items()}, T(c["depth_values"]))
    print("g21 cascade reference forward", round(time.time() - t0, 1), "s")
    arrs = {}
    for s in ("stage1", "stage2", "stage3"):
        d, cf = out[s]["depth"], out[s]["photometric_confidence"]
        if s == "stage3":
            d, cf = d[:, ::2, ::2], cf[:, ::2, ::2]
        arrs[s + "_depth"] = d.contiguous()
        arrs[s + "_conf"] = cf.contiguous()
    save("g21_cas_fullsize_scene1", **arrs)


def g22():
    import pdb
    warnings.filterwarnings("ignore")
    c = cc.cvp_fullsize_case(scene=1)
    for s in ("torchvision", "torchvision.utils", "cv2"):
        sys.modules.setdefault(s, types.ModuleType(s))
    for k in [k for k in sys.modules if k in ("models", "utils") or k.startswith("models.")]:
        del sys.modules[k]
    torch.Tensor.cuda = lambda self, *a, **k: self      # the reference hard-codes .cuda()
    pdb.set_trace = lambda *a, **k: None                # ... and a breakpoint in its forward
    sys.path.insert(0, "/root/reference/CVP-MVSNet")
    from models import net as refnet
    sys.path.pop(0)
    net = refnet.network(types.SimpleNamespace(nscale=c["nscale"], nsrc=c["nsrc"], mode="test"))
    net.load_state_dict(c["sd"])
    net.eval()
    cams, imgs = c["cams"], c["imgs"]
    t0 = time.time()
    with torch.no_grad():
        out = net(T(imgs[:, 0]), T(imgs[:, 1:]), T(cams["ref_in"]), T(cams["src_in"]), T(cams["ref_ex"]),
                  T(cams["src_ex"]), T(cams["depth_min"]), T(cams["depth_max"]))
    print("g22 cvp reference forward", round(time.time() - t0, 1), "s")
    arrs = {}
    for i, d in enumerate(out["depth_est_list"]):
        arrs[f"depth_level{i}"] = (d[:, ::2, ::2] if d.shape[-1] > 1000 else d).contiguous()
    cf = out["prob_confidence"]
    arrs["prob_confidence"] = (cf[..., ::2, ::2] if cf.shape[-1] > 1000 else cf).contiguous()
    save("g22_cvp_fullsize_scene1", **arrs)


def g23():
    arrs = {}
    for scene in (0, 1):
        c = cc.cas_fullsize_case(scene)
        t0 = time.time()
        with torch.no_grad():
            o = torch_ref.cascade_forward(T(c["imgs"]).double(), {k: T(v).double() for k, v in c["proj"].items()},
                                          T(c["depth_values"]).double(), _dbl(c["sd"]))
        print("g23 cascade float64 scene", scene, round(time.time() - t0, 1), "s")
        ref = dict(np.load(os.path.join(HERE, "g13_cas_fullsize.npz" if scene == 0 else "g21_cas_fullsize_scene1.npz")))
        for s_ in ("stage1", "stage2", "stage3"):
            d = o[s_]["depth"]
            if s_ == "stage3":
                d = d[:, ::2, ::2]
            arrs[f"s{scene}_{s_}_depth64_d16"] = _delta16(d.contiguous(), ref[s_ + "_depth"])
            print("   ", s_, "max|ref32 - f64| =", float(np.abs(ref[s_ + "_depth"].astype(np.float64) - d.numpy()).max()), "mm")
        del o
        gc.collect()
    save("g23_cas_fullsize_fp64", delta_step_mm=np.float64(DELTA_STEP_MM), **arrs)


def g24():
    arrs = {}
    for scene in (0, 1):
        c = cc.cvp_fullsize_case(scene)
        cams, imgs = {k: T(v).double() for k, v in c["cams"].items()}, T(c["imgs"]).double()
        t0 = time.time()
        with torch.no_grad():
            o = torch_ref.cvp_forward(imgs[:, 0], imgs[:, 1:], cams["ref_in"], cams["src_in"], cams["ref_ex"], cams["src_ex"],
                                      cams["depth_min"], cams["depth_max"], _dbl(c["sd"]), c["nscale"])
        print("g24 cvp float64 scene", scene, round(time.time() - t0, 1), "s")
        ref = dict(np.load(os.path.join(HERE, "g14_cvp_fullsize.npz" if scene == 0 else "g22_cvp_fullsize_scene1.npz")))
        for i, d in enumerate(o["depth_est_list"]):
            d = (d[:, ::2, ::2] if d.shape[-1] > 1000 else d).contiguous()
            arrs[f"s{scene}_depth_level{i}_64_d16"] = _delta16(d, ref[f"depth_level{i}"])
            print("    level", i, "max|ref32 - f64| =", float(np.abs(ref[f"depth_level{i}"].astype(np.float64) - d.numpy()).max()), "mm")
        del o
        gc.collect()
    save("g24_cvp_fullsize_fp64", delta_step_mm=np.float64(DELTA_STEP_MM), **arrs)


def g25():
    """configs[3] on a THIRD scene (image seed 2, weight seed 2, camera rig 2: VERDICT r03 item 6): the reference's float32
    outputs and, in the same file, the float64 evaluation of the composition (delta-coded like g24)."""
    import pdb
    warnings.filterwarnings("ignore")
    c = cc.cvp_fullsize_case(scene=2)
    for s in ("torchvision", "torchvision.utils", "cv2"):
        sys.modules.setdefault(s, types.ModuleType(s))
    for k in [k for k in sys.modules if k in ("models", "utils") or k.startswith("models.")]:
        del sys.modules[k]
    torch.Tensor.cuda = lambda self, *a, **k: self      # the reference hard-codes .cuda()
    pdb.set_trace = lambda *a, **k: None                # ... and a breakpoint in its forward
    sys.path.insert(0, "/root/reference/CVP-MVSNet")
    from models import net as refnet
    sys.path.pop(0)
    net = refnet.network(types.SimpleNamespace(nscale=c["nscale"], nsrc=c["nsrc"], mode="test"))
    net.load_state_dict(c["sd"])
    net.eval()
    cams, imgs = c["cams"], c["imgs"]
    t0 = time.time()
    with torch.no_grad():
        out = net(T(imgs[:, 0]), T(imgs[:, 1:]), T(cams["ref_in"]), T(cams["src_in"]), T(cams["ref_ex"]),
                  T(cams["src_ex"]), T(cams["depth_min"]), T(cams["depth_max"]))
        port = torch_ref.cvp_forward(T(imgs[:, 0]), T(imgs[:, 1:]), T(cams["ref_in"]), T(cams["src_in"]), T(cams["ref_ex"]),
                                     T(cams["src_ex"]), T(cams["depth_min"]), T(cams["depth_max"]), c["sd"], c["nscale"])
    print("g25 cvp reference forward (+ port)", round(time.time() - t0, 1), "s; port_vs_reference =",
          max(float((a - b).abs().max()) for a, b in zip(out["depth_est_list"], port["depth_est_list"])))
    arrs = {}
    for i, d in enumerate(out["depth_est_list"]):
        arrs[f"depth_level{i}"] = (d[:, ::2, ::2] if d.shape[-1] > 1000 else d).contiguous()
    cf = out["prob_confidence"]
    arrs["prob_confidence"] = (cf[..., ::2, ::2] if cf.shape[-1] > 1000 else cf).contiguous()
    del out, port
    gc.collect()
    cams64, imgs64 = {k: T(v).double() for k, v in c["cams"].items()}, T(c["imgs"]).double()
    t0 = time.time()
    with torch.no_grad():
        o = torch_ref.cvp_forward(imgs64[:, 0], imgs64[:, 1:], cams64["ref_in"], cams64["src_in"], cams64["ref_ex"], cams64["src_ex"],
                                  cams64["depth_min"], cams64["depth_max"], _dbl(c["sd"]), c["nscale"])
    print("g25 cvp float64", round(time.time() - t0, 1), "s")
    for i, d in enumerate(o["depth_est_list"]):
        d = (d[:, ::2, ::2] if d.shape[-1] > 1000 else d).contiguous()
        arrs[f"s2_depth_level{i}_64_d16"] = _delta16(d, arrs[f"depth_level{i}"].numpy())
        print("    level", i, "max|ref32 - f64| =", float(np.abs(arrs[f"depth_level{i}"].numpy().astype(np.float64) - d.numpy()).max()), "mm")
    save("g25_cvp_fullsize_scene2", delta_step_mm=np.float64(DELTA_STEP_MM), **arrs)


if __name__ == "__main__":
    torch.set_num_threads(8)
    warnings.filterwarnings("ignore")
    which = sys.argv[1:] or ["g17", "g18", "g19", "g20", "g21", "g22"]
    for w in which:
        globals()[w]()
        gc.collect()
