"""Pins the CPU oracles (oracle/mvs_oracle.c and oracle/torch_ref.py) against
golden vectors captured from the imported reference (tests/golden/make_golden.py).
CPU only.  The warp / variance restatements are bit-exact; convolution and
softmax sums differ from ATen/MKL-DNN only in summation order."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rot_trans_torch
from oracle import c_oracle as co
from oracle import torch_ref as tr


def _rts(proj):
    return np.stack([rot_trans_torch(proj, v) for v in range(1, proj.shape[1])])


@pytest.mark.parametrize("v", [1, 2])
def test_warp_bit_exact(v):
    g = load_golden("g1_warp")
    out = co.warp(g["src"], rot_trans_torch(g["proj"], v), g["depth"])
    ref = g[f"warped_v{v}"]
    assert (ref == 0).mean() > 0.05, "fixture must exercise zero padding"
    np.testing.assert_array_equal(out, ref)


def test_warp_identity_known_answer():
    g = load_golden("g2_identity")
    out = co.warp(g["src"], rot_trans_torch(g["proj"], 0), g["depth"])
    np.testing.assert_allclose(out, g["warped"], atol=1e-6, rtol=0)
    # independent closed form: identical projections => ix = x*W/(W-1) - 0.5
    src = g["src"].astype(np.float64)
    B, C, H, W = src.shape
    x = np.arange(W) * W / (W - 1) - 0.5
    y = np.arange(H) * H / (H - 1) - 0.5
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)

    def tap(yy, xx):
        ok = ((yy >= 0) & (yy < H))[:, None] & ((xx >= 0) & (xx < W))[None, :]
        return np.where(ok, src[:, :, np.clip(yy, 0, H - 1)][:, :, :, np.clip(xx, 0, W - 1)], 0)

    fx, fy = (x - x0)[None, :], (y - y0)[:, None]
    want = (tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) +
            tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy)
    np.testing.assert_allclose(out[:, :, 0], want, atol=2e-5)


def test_warp_per_pixel_depth_cas():
    g = load_golden("g8_cas_perpixel")
    out = co.warp(g["src"], rot_trans_torch(g["proj"], 1), g["depth"])
    np.testing.assert_array_equal(out, g["warped"])
    prob = torch.softmax(torch.from_numpy(g["cost"]), 1).numpy()
    dep, _ = co.softmax_regress_conf(g["cost"], g["depth"], clamp_idx=True)
    np.testing.assert_allclose(dep, g["regressed"], atol=2e-4)
    assert prob.shape == g["cost"].shape


@pytest.mark.parametrize("name", ["g6_e2e_64x96_v3_d8", "g3_e2e_64x64_v2_d8",
                                  "g3_e2e_64x64_v5_d8_b2"])
def test_variance_bit_exact(name):
    g = load_golden(name)
    f = g["features"]
    V = f.shape[1]
    var = co.costvol_variance(f[:, 0], np.stack([f[:, v] for v in range(1, V)]),
                              _rts(g["proj"]), g["depth_values"])
    np.testing.assert_array_equal(var, g["variance"])


def test_variance_cvp_alias_quirk():
    g = load_golden("g8_cvp_alias")
    f = g["feats"]
    var = co.costvol_variance(f[0], f[1:], _rts(g["proj"]), g["depth"], alias_quirk=True)
    np.testing.assert_array_equal(var, g["variance"])
    plain = co.costvol_variance(f[0], f[1:], _rts(g["proj"]), g["depth"], alias_quirk=False)
    assert np.abs(plain - var).max() > 1e-3


def test_variance_backward():
    g = load_golden("g7_variance_grad")
    f = g["feats"]
    gr, gs = co.costvol_variance_bwd(g["grad_out"], f[0], f[1:], _rts(g["proj"]), g["depth"])
    np.testing.assert_allclose(gr, g["grad_feats"][0], atol=2e-6)
    np.testing.assert_allclose(gs, g["grad_feats"][1:], atol=2e-6)


def test_costregnet_layers_and_whole(weights):
    g = load_golden("g6_e2e_64x96_v3_d8")
    pre = "cost_regularization."
    s, b = co.bn_fold(*(weights[f"{pre}conv0.bn.{k}"] for k in
                        ("weight", "bias", "running_mean", "running_var")))
    c0 = co.conv3d(g["variance"], weights[pre + "conv0.conv.weight"], s, b, None, True, 1)
    np.testing.assert_allclose(c0, g["act_conv0"], atol=1e-5)
    s, b = co.bn_fold(*(weights[f"{pre}conv1.bn.{k}"] for k in
                        ("weight", "bias", "running_mean", "running_var")))
    c1 = co.conv3d(g["act_conv0"], weights[pre + "conv1.conv.weight"], s, b, None, True, 2)
    np.testing.assert_allclose(c1, g["act_conv1"], atol=1e-5)
    s, b = co.bn_fold(*(weights[f"{pre}conv7.1.{k}"] for k in
                        ("weight", "bias", "running_mean", "running_var")))
    c7 = co.deconv3d(g["act_conv6"], weights[pre + "conv7.0.weight"], s, b, None, True, 2)
    np.testing.assert_allclose(c7, g["act_conv7"], atol=1e-5)
    cost = co.costregnet(g["variance"], weights)
    np.testing.assert_allclose(cost, g["cost"], atol=3e-5)


@pytest.mark.parametrize("name", ["g5_regress_d8", "g5_regress_d192"])
def test_regress_confidence_edges(name):
    g = load_golden(name)
    dep, conf = co.softmax_regress_conf(g["cost"], g["depth_values"])
    np.testing.assert_allclose(dep, g["depth"], atol=5e-4)
    np.testing.assert_allclose(conf, g["confidence"], atol=1e-6)
    d2, c2, _ = tr.regress(torch.from_numpy(g["cost"]), torch.from_numpy(g["depth_values"]))
    np.testing.assert_array_equal(d2.numpy(), g["depth"])
    np.testing.assert_array_equal(c2.numpy(), g["confidence"])


@pytest.mark.parametrize("name", ["g6_e2e_64x96_v3_d8", "g6_e2e_128x160_v3_d16",
                                  "g3_e2e_64x64_v5_d8_b2"])
def test_torch_restatement_end_to_end(name, weights):
    g = load_golden(name)
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    out = tr.mvsnet_forward(torch.from_numpy(g["imgs"]), torch.from_numpy(g["proj"]),
                            torch.from_numpy(g["depth_values"]), sd)
    np.testing.assert_allclose(out["depth"].numpy(), g["depth"], atol=1e-4)
    np.testing.assert_allclose(out["photometric_confidence"].numpy(), g["confidence"], atol=1e-6)
    spread = g["depth"].max() - g["depth"].min()
    assert spread > 50, "weights must give a non-trivial depth map"


def test_c_oracle_end_to_end(weights):
    g = load_golden("g6_e2e_64x96_v3_d8")
    f = g["features"]
    var = co.costvol_variance(f[:, 0], np.stack([f[:, v] for v in (1, 2)]), _rts(g["proj"]),
                              g["depth_values"])
    cost = co.costregnet(var, weights)
    dep, conf = co.softmax_regress_conf(cost[:, 0], g["depth_values"])
    np.testing.assert_allclose(dep, g["depth"], atol=1e-3)
    np.testing.assert_allclose(conf, g["confidence"], atol=1e-4)


def test_train_step_restatement(weights):
    g = load_golden("g7_train_step")
    sd = {k: torch.from_numpy(v).clone().requires_grad_(v.dtype == np.float32 and
                                                        "running" not in k)
          for k, v in weights.items()}
    out = tr.mvsnet_forward(torch.from_numpy(g["imgs"]), torch.from_numpy(g["proj"]),
                            torch.from_numpy(g["depth_values"]), sd, train=True)
    loss = tr.masked_smooth_l1(out["depth"], torch.from_numpy(g["gt"]), torch.from_numpy(g["mask"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    assert co.masked_smooth_l1(g["depth"], g["gt"], g["mask"]) == pytest.approx(float(g["loss"]),
                                                                                rel=1e-5)
    for k in g:
        if k.startswith("grad__"):
            ref = g[k]
            got = sd[k[6:]].grad.numpy()
            np.testing.assert_allclose(got, ref, atol=1e-4 * max(1.0, np.abs(ref).max()))


def test_torch_ref_cascade_matches_reference():
    """oracle/torch_ref.cascade_forward (3-stage CasMVSNet) vs the imported reference's
    CascadeMVSNet forward (golden g9, made by tests/golden/make_golden_cas.py)."""
    from mvs_amd import synth
    from oracle import torch_ref
    g = load_golden("g9_cas_cascade")
    sd = synth.cas_random_state_dict(int(g["seed"]))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    with torch.no_grad():
        feat = torch_ref.cas_feature_net(T(g["imgs"][:, 0]), sd)
        for k in ("stage1", "stage2", "stage3"):
            np.testing.assert_allclose(feat[k].numpy(), g[k + "_feat_ref"], atol=1e-5, rtol=1e-5)
        out = torch_ref.cascade_forward(T(g["imgs"]), {k: T(g["proj_" + k]) for k in ("stage1", "stage2", "stage3")},
                                        T(g["depth_values"]), sd)
    for k in ("stage1", "stage2", "stage3"):
        assert np.abs(out[k]["depth"].numpy() - g[k + "_depth"]).max() < 1e-3, k
        np.testing.assert_allclose(out[k]["photometric_confidence"].numpy(), g[k + "_conf"], atol=1e-4)


def test_torch_ref_cvp_matches_reference():
    """oracle/torch_ref.cvp_forward (CVP-MVSNet coarse-to-fine) vs the imported reference's
    `network` forward (golden g11, made by tests/golden/make_golden_cvp.py)."""
    from mvs_amd import synth
    from oracle import torch_ref
    g = load_golden("g11_cvp")
    sd = synth.cvp_random_state_dict(int(g["seed"]))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    imgs = T(g["imgs"])
    with torch.no_grad():
        out = torch_ref.cvp_forward(imgs[:, 0], imgs[:, 1:], *(T(g[k]) for k in (
            "ref_in", "src_in", "ref_ex", "src_ex", "depth_min", "depth_max")), sd, int(g["nscale"]))
    for i, d in enumerate(out["depth_est_list"]):
        assert np.abs(d.numpy() - g[f"depth_level{i}"]).max() < 1e-3, i
    np.testing.assert_allclose(out["prob_confidence"].numpy(), g["prob_confidence"], atol=1e-4)


def test_geo_filter_restatement_known_answers():
    """oracle/geo_filter.py: the cv2.remap restatement on known answers (integer coordinates return
    the pixel, halves average, 1/32-pixel rounding, constant border 0, NaN -> 0) and the consistency
    check on exact depth maps of a plane (every visible pixel passes; a 5 % error fails)."""
    from mvs_amd import synth
    from oracle import geo_filter as gf
    img = np.arange(20, dtype=np.float32).reshape(4, 5) ** 2
    xs, ys = np.meshgrid(np.arange(5, dtype=np.float32), np.arange(4, dtype=np.float32))
    assert np.array_equal(gf.remap_linear(img, xs, ys), img)
    half = gf.remap_linear(img, xs[:, :-1] + 0.5, ys[:, :-1])
    assert np.array_equal(half, (img[:, :-1] + img[:, 1:]) * np.float32(0.5))
    q = gf.remap_linear(img, np.float32([[1.01, 1.02]]), np.float32([[2.0, 2.0]]))     # 1.01 -> 32/32.., 1.02 -> 33/32
    assert q[0, 0] == img[2, 1] and q[0, 1] == img[2, 1] * np.float32(31 / 32) + img[2, 2] * np.float32(1 / 32)
    edge = gf.remap_linear(img, np.float32([[-0.5, 4.5, np.nan, 100.0]]), np.float32([[0.0, 3.0, 1.0, 1.0]]))
    assert edge[0, 0] == img[0, 0] * np.float32(0.5) and edge[0, 1] == img[3, 4] * np.float32(0.5)
    assert edge[0, 2] == 0 and edge[0, 3] == 0
    depths, K, E = synth.plane_depth_maps(3, 40, 56)
    mask, dep, x_src, y_src = gf.check_geometric_consistency(depths[0], K, E[0], depths[1], K, E[1])
    inside = (x_src > 1) & (x_src < 54) & (y_src > 1) & (y_src < 38)
    assert mask[inside].all() and np.abs(dep[mask] - depths[0][mask]).max() < 0.05
    bad = depths[0] * np.float32(1.05)
    assert not gf.check_geometric_consistency(bad, K, E[0], depths[1], K, E[1])[0].any()


def test_cvp_refine_mirror_matches_reference_caldepthhypo():
    """The fp64 torch restatement of calDepthHypo (mvs_amd.models.cvp_mvsnet.refine_hypotheses, the CPU
    checker of the CVP tests) against the reference's own function (g16)."""
    import torch
    from conftest import load_golden
    from mvs_amd.models.cvp_mvsnet import refine_hypotheses
    g = load_golden("g16_glue")
    T = torch.from_numpy
    got = refine_hypotheses(T(g["cvp_depth_up"]), T(g["cvp_K_ref"]), T(g["cvp_K_src"][:, 0]), T(g["cvp_ref_ex"]),
                            T(g["cvp_src_ex"][:, 0])).numpy()
    import numpy as np
    np.testing.assert_allclose(got, g["cvp_hypos"], atol=2e-4, rtol=0)
