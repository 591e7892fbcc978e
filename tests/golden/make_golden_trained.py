"""Trained-weights parity scene (no DTU data and no checkpoint exists in either container).

Run in the build container (needs /root/reference):

    python tests/golden/make_golden_trained.py train [steps]     # -> weights_trained.npz (+ the loss log)
    python tests/golden/make_golden_trained.py g26               # -> g26_trained_640x512_v3_d48.npz, g26_trained_fullsize.npz

`train` runs the REFERENCE's own model, loss and optimiser composition (MVSNet/train.py:98,204-248: model.train(),
zero_grad, forward, mvsnet_loss, backward, Adam(lr 1e-3, betas 0.9 / 0.999, wd 0) step) on the CPU over freshly
rendered scenes of mvs_amd.synth_scene (random tilt, relief, texture, camera rig; 320x256 images, V=3, D=48 over the
DTU depth range), and keeps the state_dict: learned BatchNorm scales and running statistics, a softmax as peaked as the
`prob` layer learned to make it.  `g26` then runs the reference's eval forward (MVSNet/eval.py:104-116) with those
weights on two rendered scenes at configs[0] and configs[1] size, float32 (the imported reference) and float64
(oracle/torch_ref.py), as g12 / g18 / g20 do for random weights.  Fixtures hold outputs only; the inputs are the seeded
recipe `trained_case()` below, which the GPU tests re-run (tests/trained_cases.py imports it from here? no: it lives in
mvs_amd.synth_scene so that nothing under tests/golden is needed at test time).
"""
import gc
import json
import math
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import T, _import_ref, save, synth, torch_ref  # noqa: E402
from make_golden_configs import _dbl  # noqa: E402

from mvs_amd import synth_scene  # noqa: E402

WEIGHTS = os.path.join(HERE, "weights_trained.npz")


def _random_rig(rng, V):
    """World->camera matrices of V cameras looking at the target from a jittered arc (view 0 = identity)."""
    Es = [np.eye(4)]
    target = np.array([0.0, 0.0, synth.DTU_TARGET_Z])
    for i in range(1, V):
        sign = 1.0 if i % 2 else -1.0
        th = sign * rng.uniform(4.0, 13.0)
        ph = rng.uniform(-8.0, 8.0)
        ps = rng.uniform(-3.0, 3.0)
        dist = rng.uniform(0.92, 1.08)
        R = synth._rot_z(math.radians(ps)) @ synth._rot_y(math.radians(th)) @ synth._rot_x(math.radians(ph))
        C = target - R.T @ np.array([0.0, 0.0, synth.DTU_TARGET_Z * dist])
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = -R @ C
        Es.append(E)
    return np.stack(Es)


def train_sample(rng, H=256, W=320, V=3, D=48):
    fh, fw = H // 4, W // 4
    K_img = synth_scene.image_intrinsics(fh, fw)
    K_feat = synth.feature_intrinsics(fh, fw)
    scene = synth_scene.Scene(int(rng.integers(1 << 31)), synth.DTU_TARGET_Z / K_img[0, 0],
                              z0=rng.uniform(560.0, 800.0), tilt=(rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3)),
                              relief_mm=rng.uniform(4.0, 30.0), relief_wavelength=rng.uniform(60.0, 200.0))
    Es = _random_rig(rng, V)
    gains = [(1.0, 0.0)] + [(rng.uniform(0.9, 1.1), rng.uniform(-0.03, 0.03)) for _ in range(V - 1)]
    imgs, _ = synth_scene.render(scene, K_img, Es, H, W, gains)
    _, gt = synth_scene.render(scene, K_feat, Es[:1], fh, fw)
    P = Es.copy()
    for i in range(V):
        P[i, :3, :4] = K_feat @ Es[i, :3, :4]
    dv = synth.depth_values(D, interval=synth.sweep_interval(D))
    mask = ((gt > dv[0, 0]) & (gt < dv[0, -1])).astype(np.float32)
    return imgs[None], P.astype(np.float32)[None], dv, gt, mask


def train(steps=400, batch=2):
    torch.manual_seed(26)
    mods = _import_ref("MVSNet")
    net = mods["models.mvsnet"].MVSNet(refine=False)
    loss_fn = mods["models.mvsnet"].mvsnet_loss
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0)   # train.py:98
    rng = np.random.default_rng(26)
    log = []
    t0 = time.time()
    for step in range(steps):
        samples = [train_sample(rng) for _ in range(batch)]
        imgs, proj, dv, gt, mask = (T(np.concatenate([s[i] for s in samples])) for i in range(5))
        opt.zero_grad()
        out = net(imgs, proj, dv)
        loss = loss_fn(out["depth"], gt, mask)
        loss.backward()
        opt.step()
        err = float(((out["depth"].detach() - gt).abs() * mask).sum() / mask.sum())
        log.append((float(loss), err))
        if step % 10 == 0 or step == steps - 1:
            print(f"step {step:4d}  loss {float(loss):8.3f}  abs err {err:7.3f} mm  {time.time() - t0:6.0f} s", flush=True)
        if step % 50 == 49 or step == steps - 1:
            np.savez_compressed(WEIGHTS, **{k: v.detach().numpy() for k, v in net.state_dict().items()})
    with open(os.path.join(HERE, "weights_trained_log.json"), "w") as f:
        json.dump({"steps": steps, "batch": batch, "loss_abs_err_mm": log}, f)


def _case(tag, seed, H, W, V, D, rig, interval=None):
    c = synth_scene.eval_case(seed, H, W, V, D, rig=rig, interval=interval)
    sd = {k: torch.from_numpy(v) for k, v in np.load(WEIGHTS).items()}
    mods = _import_ref("MVSNet")
    net = mods["models.mvsnet"].MVSNet(refine=False)
    net.load_state_dict(sd)
    net.eval()
    t0 = time.time()
    with torch.no_grad():
        out = net(T(c["imgs"]), T(c["proj"]), T(c["depth_values"]))
    print(tag, "reference forward", round(time.time() - t0, 1), "s; mean |depth - ground truth| =",
          float((out["depth"][0] - T(c["gt"])[0]).abs().mean()), "mm")
    del net
    gc.collect()
    with torch.no_grad():
        port = torch_ref.mvsnet_forward(T(c["imgs"]), T(c["proj"]), T(c["depth_values"]), sd)
        print("  port_vs_reference (float32 torch_ref vs reference) max", float((port["depth"] - out["depth"]).abs().max()))
        del port
        gc.collect()
        t0 = time.time()
        o64 = torch_ref.mvsnet_forward(T(c["imgs"]).double(), T(c["proj"]).double(), T(c["depth_values"]).double(), _dbl(sd))
        print(tag, "float64 forward", round(time.time() - t0, 1), "s; max|ref32 - f64| =",
              float((out["depth"].double() - o64["depth"]).abs().max()), "mm")
    return c, out, o64


def g26():
    c, out, o64 = _case("g26 small", 260, 512, 640, 3, 48, rig=0, interval=synth.sweep_interval(48))
    save("g26_trained_640x512_v3_d48", depth=out["depth"], confidence=out["photometric_confidence"],
         depth64=o64["depth"], confidence64=o64["photometric_confidence"], gt=c["gt"])
    c, out, o64 = _case("g26 full", 261, 1184, 1600, 5, 192, rig=1)
    save("g26_trained_fullsize", depth=out["depth"], confidence=out["photometric_confidence"],
         depth64_delta32=(o64["depth"].numpy() - out["depth"].numpy().astype(np.float64)).astype(np.float32),
         confidence64=o64["photometric_confidence"].float(), gt=c["gt"])
    # (a trained network's peaked softmax puts the reference's float32 forward up to ~1e-2 mm from the float64 answer: the
    # int16 delta coding of g20 -- steps of 5e-8 mm -- does not hold it; a float32 difference does, to 1e-9 mm)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "g26"
    if what == "train":
        train(int(sys.argv[2]) if len(sys.argv) > 2 else 400)
    else:
        g26()
