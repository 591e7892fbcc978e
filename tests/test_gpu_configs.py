"""BASELINE configs[0] and configs[4] at their own workloads, single GPU (VERDICT r02 items 1 and `configs_untested`).

Fixtures (tests/golden/make_golden_configs.py, build container): the imported reference's float32 outputs AND the same
composition evaluated in float64 (oracle/torch_ref.py in double) -- the "true" answer both float32 evaluations
approximate.  Two statements per case:
  * the north-star gate, |HIP - reference| < 1e-3 mm, for the eval forward;
  * the error budget: HIP is no farther from the float64 answer than the reference itself is.
For the train() step (batch-statistics BatchNorm, a softmax sharpened x30: the reference's own float32 step is 0.23 mm and
1-3 % of the largest gradient away from the float64 step) the second statement IS the gate -- it replaces the hand-set
atol=5e-2 / 2e-3 * max of round 2's small-fixture test with bounds measured against the truth."""
import pytest
import torch

from fullsize_cases import run_eval_small, run_train_step

pytestmark = pytest.mark.gpu
GATE_MM = 1e-3


@pytest.mark.parametrize("fast", [False, True], ids=["exact_coordinates", "fast_coordinates"])
def test_mvsnet_config0_eval_640x512_v3_d48(fast):
    """configs[0]: MVSNet/eval.py's forward at 640x512, N=3, D=48 (eval.py:96-131) against g18."""
    with torch.no_grad():
        r = run_eval_small(fast)
    assert r["maxabs_mm"] < GATE_MM, r
    assert r["hip_vs_f64_mm"] < GATE_MM and r["hip_vs_f64_mm"] <= 1.1 * r["ref_vs_f64_mm"], r
    assert r["hip_vs_f64_rms"] <= 1.1 * r["ref_vs_f64_rms"], r
    c = r["conf"]
    assert c["unexplained"] == 0 and c.get("maxabs_without_flips", c["maxabs"]) < 2e-4, c


def test_mvsnet_config4_train_step_640x512_v3_d192():
    """configs[4], one GPU's share of a step (MVSNet/train.py:204-248 up to loss.backward()): 640x512, V=3, D=192, B=1
    on the HIP training path against the reference's own step (g17: depth, loss, all 338,129 gradient elements, BatchNorm
    running statistics) and against the float64 step."""
    r = run_train_step()
    d = r["depth"]
    # forward: as close to the float64 depth as the reference's train() forward is (max and rms)
    assert d["hip_vs_f64_mm"] <= 1.25 * d["ref_vs_f64_mm"], d
    assert d["hip_vs_f64_rms"] <= 1.1 * d["ref_vs_f64_rms"], d
    # against the reference's own train-mode depth: a FIXED bound (the sum of the two distances to float64 is the
    # triangle inequality and cannot fail).  Measured 0.218 mm, with the reference itself 0.234 mm from the float64 step (batch-
    # statistics BatchNorm and a x30 sharper softmax amplify float32 rounding ~300x over the eval forward's 7e-4 mm)
    assert d["maxabs_mm"] <= 0.30, d
    # loss
    assert abs(r["loss"] - r["loss64"]) <= 2.0 * abs(r["loss_ref"] - r["loss64"]) + 1e-6 * abs(r["loss64"]), r
    assert abs(r["loss"] - r["loss_ref"]) <= 1e-5 * abs(r["loss_ref"]), r
    # gradients: over all elements no worse (rms) than the reference's float32 backward; per tensor within 2x of it
    ga = r["grads_all"]
    assert ga["hip_vs_f64_rms"] <= 1.1 * ga["ref_vs_f64_rms"], ga
    # (the training path takes rot_trans from mvs_rot_trans_f32 -- float64 internally -- not from the reference's float32
    # LAPACK inverse, whose rounding is most of the reference's own distance from the float64 step: the two float32
    # results are therefore about as far from each other as the reference is from the truth)
    assert ga["hip_vs_ref_rms"] <= 1.5 * ga["ref_vs_f64_rms"], ga
    for k, v in r["grads"].items():
        assert v["hip_vs_f64"] <= 2.0 * v["ref_vs_f64"] + 1e-4, (k, v)
    # BatchNorm running statistics after the step (momentum 0.1, unbiased variance)
    for k, v in r["stats"].items():
        assert v < 5e-6, (k, v)


def test_three_training_steps_follow_the_reference_trajectory():
    """SURVEY 8(c)'s last row: THREE consecutive steps of the reference's train_sample composition (MVSNet/train.py:98,204-248:
    zero_grad -> forward(train) -> mvsnet_loss -> backward -> Adam(1e-3, 0.9 / 0.999, wd 0).step()) on one fixed batch -- g27 holds
    the reference's own three losses, three named parameters and four BatchNorm running statistics after step 3, and the float64
    trajectory of the same composition.

    What can be asserted: step 1 starts from the same weights -- its loss is the reference's to 1e-5 (as g17).  From then on the
    trajectory is ILL-CONDITIONED in float32 and the fixture shows it: Adam divides every gradient element by its own running
    magnitude, so an element whose float32 gradient is off by a few per cent of the tensor's largest (g17) moves by a different
    +-1e-3 -- the reference's own float32 parameters end 9e-4 rms from the float64 ones after three steps, of a 2e-3 rms total
    movement.  Three float32 evaluations of this trajectory that differ only in rounding (this path; the same with the reference's
    float32 LAPACK inverse, train_proj_where="host"; the same with ATen's convolutions, train_impl="torch") end 5-9e-4 rms from the
    reference's parameters -- the reference's own distance from float64 -- and their third losses spread over 89.14 ... 89.56
    around the reference's 89.50 and float64's 89.54 (scripts/exp_train3.py, profiles/r06_train3_switches.json).  So: losses 2
    and 3 within 1e-2 relative of the reference's and of float64's; per parameter tensor the rms distance to the reference's AND
    to the float64 parameters no larger than 1.5x the reference's own distance to float64; running statistics within 5 % of
    their largest entry of the reference's (they average three batches of activations that already differ)."""
    import numpy as np
    import config_cases as cc
    from fullsize_cases import GOLDEN, _dev
    from mvs_amd.models import MVSNet, mvsnet_loss
    import os
    dev = torch.device("cuda:0")
    c = cc.train_case()
    g = dict(np.load(os.path.join(GOLDEN, "g27_train_3steps.npz")))
    model = MVSNet(refine=False)
    model.load_state_dict(c["sd"])
    model = model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0)
    imgs, proj, dv, gt, mask = (_dev(c[k], dev) for k in ("imgs", "proj", "depth_values", "gt", "mask"))
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = model(imgs, proj, dv)
        loss = mvsnet_loss(out["depth"], gt, mask)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    ref, l64 = g["losses"], g["losses64"]
    assert abs(losses[0] - ref[0]) <= 1e-5 * ref[0], (losses, ref)
    for i in (1, 2):
        assert abs(losses[i] - ref[i]) <= 1e-2 * ref[i] and abs(losses[i] - l64[i]) <= 1e-2 * l64[i], (losses, ref, l64)
    assert losses[2] < losses[1] < losses[0]
    sd = {k: v.detach().double().cpu().numpy() for k, v in model.state_dict().items()}
    rms = lambda a: float(np.sqrt((a ** 2).mean()))   # noqa: E731
    report = {}
    for k in (x[7:] for x in g if x.startswith("param__")):
        p32, p64 = g["param__" + k].astype(np.float64), g["param64__" + k]
        own = rms(p32 - p64)
        report[k] = (rms(sd[k] - p32), rms(sd[k] - p64), own)
        assert rms(sd[k] - p32) <= 1.5 * own and rms(sd[k] - p64) <= 1.5 * own, (k, report[k])
    for k in (x[6:] for x in g if x.startswith("stat__")):
        s32 = g["stat__" + k].astype(np.float64)
        assert np.abs(sd[k] - s32).max() <= 5e-2 * np.abs(s32).max(), (k, float(np.abs(sd[k] - s32).max()), float(np.abs(s32).max()))
