"""The range guard of the two-piece fp16 convolutions (mvs_amd/csrc/conv_guard.h) against the reference's own operator:
ATen's float32 convolution on the CPU (what MVSNet/models/mvsnet.py:83-93 runs on the volume module.py:83-84 sampled).

The reference lets a non-finite sample poison its receptive field and nothing else; one global operand scale per tensor would
flush every other voxel to zero.  So: with an Inf, a NaN, or a finite outlier 2^30 above the rest in the input, every element the
reference leaves finite must match it to float32 tolerance -- against a float64 evaluation, |error| <= 2e-6 x the local sum of
|w| |x| -- and every element the reference makes NaN / +Inf / -Inf must be exactly that; a slab at 2^-20 of the maximum keeps the
two-piece arithmetic and still meets the bound.  VERDICT r03 "next round" item 1, ADVICE r03 (conv_split_common.h:93)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = ("clean", "inf", "neg_inf", "nan", "outlier_2p30", "slab_2m20")
TAKES_FALLBACK = {"clean": False, "inf": True, "neg_inf": True, "nan": True, "outlier_2p30": True, "slab_2m20": False}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mvs_amd import _lib
    _lib.load()  # must exist: no fallback
    return torch.device("cuda:0")


def _poison(x, case, pos):
    """x [B,C,*dims] -> a copy with the case's damage at voxel `pos` (channel 1).  The slab case replaces the data by values
    within a factor 4 of their maximum first: "2^-20 of the maximum" is then 2^-20 of everything else too, and the two-piece
    bound (2^-40 of the maximum, absolute) is 2^-20 relative there -- on heavy-tailed data the same slab would sit 2^-30 below
    the maximum and keep 10 bits (the documented limit of one scale per tensor, include/mvs_hip.h "RANGE GUARD")."""
    x = x.clone()
    idx = (0, 1) + tuple(pos)
    if case == "slab_2m20":
        x = 0.25 + 0.75 * torch.rand(x.shape, generator=torch.Generator().manual_seed(77))
    if case == "inf":
        x[idx] = float("inf")
    elif case == "neg_inf":
        x[idx] = float("-inf")
    elif case == "nan":
        x[idx] = float("nan")
    elif case == "outlier_2p30":
        x[idx] = x.abs().max() * 2.0 ** 30
    elif case == "slab_2m20":
        sl = [slice(None)] * x.dim()
        sl[2] = slice(0, max(1, x.shape[2] // 3))     # the first third along the first spatial axis
        x[tuple(sl)] *= 2.0 ** -20
    return x


def _check(got, ref32, ref64, bound, what, reach=None):
    """got (device) against ref32 = the reference's float32 result and ref64 = float64 of the input with its non-finite
    elements zeroed.  reach: the outputs a non-finite input element contributes to -- there the reference's value (NaN, +-Inf,
    or the finite relu(-Inf) = 0 [+ skip]) must be reproduced exactly; everywhere else the float32 bound holds.  Returns the
    number of non-finite outputs."""
    got = got.detach().cpu()
    nan_r, inf_r = torch.isnan(ref32), torch.isinf(ref32)
    assert torch.equal(torch.isnan(got), nan_r), f"{what}: NaN pattern differs ({int(torch.isnan(got).sum())} vs {int(nan_r.sum())})"
    assert torch.equal(torch.isinf(got), inf_r) and torch.equal(got[inf_r], ref32[inf_r]), f"{what}: Inf pattern / signs differ"
    if reach is None:       # (no poisoned input element: whatever the reference makes non-finite)
        reach = nan_r | inf_r
    assert not (nan_r | inf_r)[~reach].any(), f"{what}: the reference is non-finite outside the poisoned element's receptive field?"
    inside = reach & ~nan_r
    assert torch.equal(got[inside], ref32[inside]), f"{what}: differs from the reference inside the poisoned receptive field"
    err = (got.double() - ref64)[~reach].abs()
    worst = (err / bound[~reach]).max().item() if err.numel() else 0.0
    assert worst <= 1.0, f"{what}: error {worst:.2f} x the float32 bound at the worst element outside the poisoned receptive field"
    return int((nan_r | inf_r).sum())


def _fallbacks():
    from mvs_amd import ops
    return ops.guard_fallback_count()


def _layer_reference(x, w, scale, shift, res, relu, conv, **kw):
    """The layer as the reference computes it (float32 ATen) + in float64 + the per-element float32 error bound."""
    cout = scale.numel()
    v = (1, cout) + (1,) * (x.dim() - 2)
    act = (lambda t: torch.relu(t)) if relu == 1 else ((lambda t: F.leaky_relu(t, 0.1)) if relu == 2 else (lambda t: t))
    y32 = act(conv(x, w, **kw) * scale.view(v) + shift.view(v))
    xf = torch.where(torch.isfinite(x), x, torch.zeros_like(x))
    y64 = act(conv(xf.double(), w.double(), **kw) * scale.double().view(v) + shift.double().view(v))
    mag = conv(xf.double().abs(), w.double().abs(), **kw) * scale.double().abs().view(v)
    reach = conv((~torch.isfinite(x)).double(), torch.ones_like(w).double(), **kw) > 0     # outputs with a non-finite term
    perm = (0,) + tuple(range(2, x.dim())) + (1,)
    y32, y64, mag, reach = y32.permute(perm), y64.permute(perm), mag.permute(perm), reach.permute(perm)
    if res is not None:
        y32, y64 = y32 + res, y64 + res.double()
    bound = 2e-6 * mag + 3e-7 * (y64.abs() + (res.double().abs() if res is not None else 0)) + 1e-30
    return y32.contiguous(), y64.contiguous(), bound.contiguous(), reach.contiguous()


@pytest.mark.parametrize("case", CASES)
def test_conv0_two_piece_keeps_damage_local(dev, case):
    """mvs_conv3d_c8_f16x3_f32 (conv0, mvsnet.py:66,83) on a variance-like volume with one poisoned voxel."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(11)
    B, cin, D, H, W = 1, 32, 12, 20, 70
    x = (torch.randn(B, cin, D, H, W, generator=g) * torch.rand(B, cin, D, H, W, generator=g) ** 4).square()
    x = _poison(x, case, (5, 9, 33))
    w = torch.randn(8, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
    res = torch.randn(B, D, H, W, 8, generator=g)
    y32, y64, bound, reach = _layer_reference(x, w, scale, shift, res, 1, F.conv3d, padding=1)
    n0 = _fallbacks()
    om = ops.absmax_block(dev, zero=True)
    got = ops.conv3d_c8_f16x3(ops.nchw_to_c8(x.to(dev)), ops.pack_conv3d_weight_f16x3(w.to(dev)), None, scale.to(dev), shift.to(dev),
                              res.to(dev), True, out_absmax=om)
    nbad = _check(got, y32, y64, bound, f"conv0 / {case}", reach)
    assert (_fallbacks() - n0 > 0) == TAKES_FALLBACK[case], (case, _fallbacks() - n0)
    if case in ("inf", "neg_inf", "nan"):
        assert 0 < nbad <= 27 * 8           # inside the 3x3x3 receptive field only
        assert not np.isfinite(ops.absmax_value(om))      # the next layer's guard sees it
    else:
        assert ops.absmax_value(om) == got.abs().max().item()


@pytest.mark.parametrize("case", ("inf", "nan", "outlier_2p30", "slab_2m20"))
@pytest.mark.parametrize("kd,cin,cout,shape,k,stride,relu", [
    (3, 16, 16, (1, 6, 9, 21), 3, 1, 1), (3, 8, 16, (1, 9, 20, 37), 3, 2, 1), (3, 32, 64, (1, 6, 9, 18), 3, 2, 1),
    (3, 64, 64, (1, 4, 10, 18), 3, 1, 0), (1, 16, 16, (3, 21, 45), 3, 1, 2), (1, 8, 16, (2, 37, 70), 5, 2, 1)])
def test_conv_split_two_piece_keeps_damage_local(dev, case, kd, cin, cout, shape, k, stride, relu):
    """mvs_conv_split_f16_f32 (conv1 .. conv6, mvsnet.py:67-72; FeatureNet's layers, mvsnet.py:13-27): 3D / 2D, both strides, 5x5,
    two launches per layer, every activation."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(kd * 1000 + cin * 10 + cout + shape[-1])
    dims = shape[1:]
    x = torch.randn(shape[0], cin, *dims, generator=g).clamp_min(0) * torch.rand(shape[0], cin, *dims, generator=g) ** 2 * 3.0 + 1e-3
    x = _poison(x, case, tuple(d // 2 for d in dims))
    w = torch.randn(cout, cin, *([3] * (kd == 3)), k, k, generator=g) / (k * k * (3 if kd == 3 else 1) * cin) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    conv = F.conv3d if kd == 3 else F.conv2d
    y32, y64, bound, reach = _layer_reference(x, w, scale, shift, None, relu, conv, stride=stride, padding=k // 2)
    res = torch.randn(y32.shape, generator=g) if relu != 2 else None
    if res is not None:
        y32, y64 = y32 + res, y64 + res.double()
        bound = bound + 3e-7 * res.double().abs()
    pin = (0, 2, 3, 4, 1) if kd == 3 else (0, 2, 3, 1)
    n0 = _fallbacks()
    om = ops.absmax_block(dev, zero=True)
    got = ops.conv_split_f16(x.permute(pin).contiguous().to(dev), ops.pack_conv_weight_split_f16(w.to(dev), stride), cout, None,
                             scale.to(dev), shift.to(dev), res.to(dev) if res is not None else None, relu, kd=kd, stride=stride, out_absmax=om)
    _check(got, y32, y64, bound, f"conv_split {kd}D {cin}->{cout} s{stride} / {case}", reach)
    assert (_fallbacks() - n0 > 0) == TAKES_FALLBACK[case]


@pytest.mark.parametrize("case", ("neg_inf", "nan", "outlier_2p30"))
@pytest.mark.parametrize("cin,cout,shape", [(16, 8, (1, 5, 9, 21)), (64, 32, (1, 4, 5, 16))])
def test_deconv_split_two_piece_keeps_damage_local(dev, case, cin, cout, shape):
    """mvs_deconv_split_f16_f32 (conv7 / conv9 / conv11, mvsnet.py:76-81,89-91): the skip is added after the ReLU."""
    from mvs_amd import ops
    B, D, H, W = shape
    g = torch.Generator().manual_seed(cin * 10 + cout + W)
    x = _poison(torch.randn(B, cin, D, H, W, generator=g).clamp_min(0) * 2.0 + 1e-3, case, (D // 2, H // 2, W // 2))
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) / (27 * cin / 8) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, 2 * D, 2 * H, 2 * W, cout, generator=g)
    y32, y64, bound, reach = _layer_reference(x, w, scale, shift, res, 1, F.conv_transpose3d, stride=2, padding=1, output_padding=1)
    n0 = _fallbacks()
    got = ops.deconv_split_f16(x.permute(0, 2, 3, 4, 1).contiguous().to(dev), ops.pack_deconv_weight_split_f16(w.to(dev)), cout, None,
                               scale.to(dev), shift.to(dev), res.to(dev), True)
    _check(got, y32, y64, bound, f"deconv {cin}->{cout} / {case}", reach)
    assert _fallbacks() - n0 > 0


def test_non_finite_weights_take_the_fp32_path(dev):
    """A NaN weight poisons every output of its channel in the reference; the packed two-piece weights would have been scaled by
    2^-113 instead."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 16, 4, 6, 18, generator=g)
    w = torch.randn(16, 16, 3, 3, 3, generator=g) / 20
    w[5, 2, 1, 1, 1] = float("nan")
    one, zero = torch.ones(16), torch.zeros(16)
    y32, y64, bound, reach = _layer_reference(x, w, one, zero, None, 1, F.conv3d, padding=1)
    n0 = _fallbacks()
    got = ops.conv_split_f16(x.permute(0, 2, 3, 4, 1).contiguous().to(dev), ops.pack_conv_weight_split_f16(w.to(dev), 1), 16, None,
                             one.to(dev), zero.to(dev), None, 1)
    assert _check(got, y32, y64, bound, "NaN weight") == x[0, 0].numel()       # the whole channel 5 (no input is non-finite: reach is empty)
    assert _fallbacks() - n0 > 0


def test_absmax_collectors_agree_on_non_finite(dev):
    """The block an epilogue collects and the one mvs_absmax_f32 collects by a pass over the same tensor judge it alike (ADVICE
    r03: fmaxf dropped a NaN in the epilogues, the pass let it win)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.rand(1, 16, 4, 8, 20, generator=g)
    x[0, 3, 2, 4, 10] = float("nan")
    w = torch.randn(16, 16, 3, 3, 3, generator=g) / 20
    om = ops.absmax_block(dev, zero=True)
    got = ops.conv_split_f16(x.permute(0, 2, 3, 4, 1).contiguous().to(dev), ops.pack_conv_weight_split_f16(w.to(dev), 1), 16, None,
                             None, None, None, 1, out_absmax=om)
    assert torch.isnan(got).any()
    assert np.isnan(ops.absmax_value(om)) and np.isnan(ops.absmax_value(ops.absmax(got)))
    # ... and so do the fp32 MFMA layers that sit between two-piece layers (conv3, conv5): NaN in, NaN through the ReLU, NaN in the block
    w2 = torch.randn(32, 16, 3, 3, 3, generator=g) / 20
    om2 = ops.absmax_block(dev, zero=True)
    y = ops.conv3d(got, w2.to(dev), None, None, None, True, False, 2, channels_last=True, packed=ops.pack_conv3d_weight(w2.to(dev), False, 2),
                   impl=ops.IMPL_MFMA, out_absmax=om2)
    ref = torch.relu(F.conv3d(got.cpu().permute(0, 4, 1, 2, 3), w2, stride=2, padding=1)).permute(0, 2, 3, 4, 1)
    assert torch.equal(torch.isnan(y.cpu()), torch.isnan(ref)) and torch.isnan(ref).any()
    assert np.isnan(ops.absmax_value(om2))


@pytest.mark.parametrize("case", ("nan", "neg_inf", "outlier_2p30"))
def test_costregnet_keeps_damage_where_the_reference_does(dev, weights, case):
    """The whole U-Net through mvs_costreg_fwd2_f32 (one C call, absmax blocks chained through its workspace) on a variance volume
    with one poisoned voxel, against the reference's CostRegNet composition in ATen float32 (oracle/torch_ref.cost_reg_net): the
    regularised cost is non-finite exactly where the reference's is and matches it elsewhere."""
    from mvs_amd import ops
    from mvs_amd.models import MVSNet
    from oracle import torch_ref
    if not (ops.split_f16_enabled() and ops.conv0_f16_enabled()):
        pytest.skip("the guarded path is the DEFAULT arithmetic; MVS_CONV0_F16=0 / MVS_SPLIT_F16=0 select the unguarded exact kernels, "
                    "which turn a non-finite voxel into NaN over a slightly wider field (DESIGN section 3)")
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    model = MVSNet(refine=False)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(21)
    B, D, H, W = 1, 16, 24, 136       # wide enough for voxels the poison cannot reach (the U-Net's receptive field: ~37 voxels)
    var = (torch.randn(B, 32, D, H, W, generator=g) * torch.rand(B, 32, D, H, W, generator=g) ** 3).square() * 0.05
    var = _poison(var, case, (9, 13, 6))
    with torch.no_grad():
        ref = torch_ref.cost_reg_net(var, sd)[:, 0]
        varf = torch.where(torch.isfinite(var), var, torch.zeros_like(var))
        ref64 = torch_ref.cost_reg_net(varf.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})[:, 0]
        n0 = _fallbacks()
        got = model.cost_regularization.forward_hip(ops.nchw_to_c8(var.to(dev)), in_c8=True).cpu()
        assert _fallbacks() - n0 > 0
    bad = ~torch.isfinite(ref)
    assert torch.equal(torch.isnan(got), torch.isnan(ref)), (int(torch.isnan(got).sum()), int(torch.isnan(ref).sum()))
    assert torch.equal(torch.isinf(got), torch.isinf(ref)) and torch.equal(got[torch.isinf(ref)], ref[torch.isinf(ref)])
    if case != "outlier_2p30":
        assert bad.any() and not bad.all()      # the damage is there, and it is local
        fin = ~bad
        # where the reference is finite the float64 evaluation of the CLEANED volume is the same composition
        err = (got.double() - ref64)[fin].abs().max().item()
        err_ref = (ref.double() - ref64)[fin].abs().max().item()
        assert err <= max(2.0 * err_ref, 2e-5), (err, err_ref)
    else:
        # a finite outlier: every output is finite in the reference; the guarded path must be as close to float64 as it is
        ref64 = torch_ref.cost_reg_net(var.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})[:, 0]
        err = (got.double() - ref64).abs()
        err_ref = (ref.double() - ref64).abs()
        scale = ref64.abs().max().item()
        assert err.max().item() <= 2.0 * err_ref.max().item() + 1e-6 * scale, (err.max().item(), err_ref.max().item(), scale)
