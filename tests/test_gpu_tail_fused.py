"""conv11 + prob as one kernel (mvs_amd/csrc/tail_fused.hip) against the reference's composition -- MVSNet/models/mvsnet.py:
79-81, 89-93: x = conv0 + relu(bn(ConvTranspose3d(x))), then prob = Conv3d(8 -> 1) with bias -- evaluated in float64 by ATen on
the CPU, and against the two unfused layers of the same library."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _case(seed, B, Di, Hi, Wi, mag=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Di, Hi, Wi, 16, generator=g).clamp_min(0) * mag                    # conv9's output: after a ReLU + skip
    skip = torch.randn(B, 2 * Di, 2 * Hi, 2 * Wi, 8, generator=g).clamp_min(0) * 0.7 * mag
    w11 = torch.randn(16, 8, 3, 3, 3, generator=g) * 0.08
    sc = 0.5 + torch.rand(8, generator=g)
    sh = torch.randn(8, generator=g) * 0.1 * mag
    pw = torch.randn(1, 8, 3, 3, 3, generator=g) * 0.1
    pb = torch.randn(1, generator=g) * 0.05
    return x, skip, w11, sc, sh, pw, pb


def _ref64(x, skip, w11, sc, sh, pw, pb):
    xd = x.double().permute(0, 4, 1, 2, 3)
    y = F.conv_transpose3d(xd, w11.double(), stride=2, padding=1, output_padding=1)
    y = torch.relu(y * sc.double().view(1, 8, 1, 1, 1) + sh.double().view(1, 8, 1, 1, 1)) + skip.double().permute(0, 4, 1, 2, 3)
    return F.conv3d(y, pw.double(), pb.double(), padding=1)[:, 0], y


def _run(dev, case, unfused=True):
    from mvs_amd import ops
    x, skip, w11, sc, sh, pw, pb = (t.to(dev) for t in case)
    xa, sa = ops.absmax(x), ops.absmax(skip)
    tail = ops.pack_costreg_tail(w11)
    cost, flag = ops.costreg_tail(x, xa, skip, sa, tail, sc, sh, pw, None, pb)
    if not unfused:
        torch.cuda.synchronize()
        return cost.cpu(), int(flag.item()), None
    # the unfused layers of the same library: deconv_split (two-piece) then the fp32 prob kernel
    pk = ops.pack_deconv_weight_split_f16(w11)
    d11 = ops.deconv_split_f16(x, pk, 8, xa, sc, sh, skip, True)
    unf = ops.conv3d(d11, pw, None, pb, None, False, False, 1, channels_last=True, packed=ops.pack_conv3d_weight(pw, False, 1))
    torch.cuda.synchronize()
    return cost.cpu(), int(flag.item()), unf.cpu().reshape(cost.shape)


@pytest.mark.parametrize("shape", [(1, 4, 8, 16), (2, 3, 11, 19), (1, 12, 20, 40), (1, 40, 30, 60), (1, 1, 7, 15), (1, 24, 37, 50)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fused_tail_matches_float64_and_the_unfused_layers(dev, shape):
    """One column and partial tiles; B = 2; more (column, step) pairs than CUs -- ranges that begin inside a column and recompute a
    step; a single input plane; odd sizes in every dimension."""
    case = _case(hash(shape) % 1000, *shape)
    ref, d11 = _ref64(*case)
    cost, flag, unf = _run(dev, case)
    assert flag == 0
    assert cost.shape == ref.shape and torch.isfinite(cost).all()
    scale = float(ref.abs().max())
    e_f, e_u = float((cost.double() - ref).abs().max()), float((unf.double() - ref).abs().max())
    # both are float32 accumulations of 216 + 128 products of a float32 intermediate: errors of a few 1e-7 of the largest value
    assert e_f <= max(2.5 * e_u, 2e-6 * scale), (e_f, e_u, scale)
    assert e_f <= 5e-6 * scale, (e_f, scale)


@pytest.mark.parametrize("mag", [1e-12, 1e-4, 1e6, 1e15], ids=lambda m: f"x{m:g}")
def test_fused_tail_scales_follow_the_data(dev, mag):
    """Nothing is tuned to one magnitude: the operand scales come from the absmax blocks and the weights."""
    case = _case(7, 1, 6, 12, 24, mag=mag)
    ref, _ = _ref64(*case)
    cost, flag, unf = _run(dev, case)
    assert flag == 0
    scale = float(ref.abs().max())
    assert float((cost.double() - ref).abs().max()) <= 5e-6 * scale


@pytest.mark.parametrize("poison", ["nan_in", "inf_skip", "nan_prob_weight", "outlier"])
def test_fused_tail_declines_what_the_guard_declines(dev, poison):
    """A non-finite voxel in either input, non-finite prob weights, an outlier-dominated input: the launch sets the flag and
    leaves the work to the unfused layers (whose own guard reproduces the reference's Inf / NaN pattern)."""
    from mvs_amd import ops
    x, skip, w11, sc, sh, pw, pb = _case(3, 1, 8, 16, 32)
    if poison == "nan_in":
        x[0, 2, 3, 4, 5] = float("nan")
    elif poison == "inf_skip":
        skip[0, 5, 6, 7, 1] = float("inf")
    elif poison == "nan_prob_weight":
        pw[0, 3, 1, 1, 1] = float("nan")
    else:
        x[0, 1, 1, 1, 1] = 2.0 ** 40
    # (the fused launch alone: what the unfused layers then do with such inputs is tests/test_gpu_range_guard.py's subject)
    _, flag, _ = _run(dev, (x, skip, w11, sc, sh, pw, pb), unfused=False)
    assert flag == 1


def _guarded_params(dev, case):
    from mvs_amd import ops
    x, skip, w11, sc, sh, pw, pb = (t.to(dev) for t in case)
    pk11 = ops.pack_conv3d_weight(w11, True, 2, split=True)
    p11 = dict(weight=w11, packed=pk11, scale=sc, shift=sh, packed_tail=ops.pack_costreg_tail(w11))
    assert ops.f16_companion(pk11) is not None
    pprob = dict(weight=pw, packed=ops.pack_conv3d_weight(pw, False, 1), shift=pb)
    return x, skip, p11, pprob


def test_guarded_tail_leaves_the_cost_alone_when_the_fused_kernel_ran(dev):
    """the two unfused layers enqueued behind the fused kernel must return at once when its flag stays 0 -- with the
    scratch volume poisoned, the cost is the fused kernel's bit for bit and the scratch is untouched."""
    from mvs_amd import ops
    case = _case(11, 1, 6, 12, 24)
    x, skip, p11, pprob = _guarded_params(dev, case)
    xa, sa = ops.absmax(x), ops.absmax(skip)
    alone, flag = ops.costreg_tail(x, xa, skip, sa, p11["packed_tail"], p11["scale"], p11["shift"], pprob["weight"], None, pprob["shift"])
    d11 = torch.full((1, 12, 24, 48, 8), float("nan"), device=dev)
    flag2 = torch.zeros(1, device=dev, dtype=torch.int32)
    cost = ops.costreg_tail_guarded(x, xa, skip, sa, p11, pprob, flag=flag2, d11=d11)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0 and int(flag2.item()) == 0
    assert torch.equal(cost, alone) and torch.isfinite(cost).all()
    assert torch.isnan(d11).all()


def test_guarded_tail_refuses_a_fallback_that_would_ignore_the_flag(dev):
    """Without prob's packed weights the unfused `prob` would take the direct kernel, which does not read the run-only-if word:
    the call must fail before anything is enqueued (MVS_EUNSUPPORTED), and so must any non-flag-aware launch under a flag."""
    from mvs_amd import ops
    case = _case(12, 1, 4, 8, 16)
    x, skip, p11, pprob = _guarded_params(dev, case)
    xa, sa = ops.absmax(x), ops.absmax(skip)
    bad = dict(pprob)
    bad.pop("packed")
    with pytest.raises(ops.MvsHipError, match="packed"):
        ops.costreg_tail_guarded(x, xa, skip, sa, p11, bad)


@pytest.mark.poisoned_inputs
def test_guarded_tail_falls_back_when_the_fused_kernel_declines(dev):
    """A NaN voxel in conv11's input: the fused kernel declines, the unfused layers run behind the flag and the cost carries the
    reference's NaN pattern (ATen float32 on the CPU is the checker, as in tests/test_gpu_range_guard.py)."""
    from mvs_amd import ops
    case = list(_case(13, 1, 4, 8, 16))
    case[0][0, 1, 2, 3, 4] = float("nan")
    x, skip, p11, pprob = _guarded_params(dev, case)
    xa, sa = ops.absmax(x), ops.absmax(skip)
    flag = torch.zeros(1, device=dev, dtype=torch.int32)
    cost = ops.costreg_tail_guarded(x, xa, skip, sa, p11, pprob, flag=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    xc, sk, w11, sc, sh, pw, pb = case
    y = F.conv_transpose3d(xc.permute(0, 4, 1, 2, 3), w11, stride=2, padding=1, output_padding=1)
    y = torch.relu(y * sc.view(1, 8, 1, 1, 1) + sh.view(1, 8, 1, 1, 1)) + sk.permute(0, 4, 1, 2, 3)
    ref = F.conv3d(y, pw, pb, padding=1)[:, 0]
    got = cost.cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(ref)) and torch.isnan(ref).any()
    ok = ~torch.isnan(ref)
    assert float((got[ok] - ref[ok]).abs().max()) <= 5e-6 * float(ref[ok].abs().max())


def test_costregnet_through_the_fused_tail_equals_the_unfused_net(dev, weights, monkeypatch):
    """mvs_costreg_fwd3_f32 with and without the fused tail (MVS_TAIL_FUSED) on a variance-like volume: same regularised cost to
    float32 rounding, and the fused call launches neither conv11's nor prob's full kernel (their run flag stays 0)."""
    from mvs_amd import ops
    from mvs_amd.models import MVSNet
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    model = MVSNet(refine=False)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    B, D, H, W = 1, 16, 40, 72
    var = ((torch.randn(B, 32, D, H, W, generator=g) * torch.rand(B, 32, D, H, W, generator=g) ** 3).square() * 0.05)
    vc8 = ops.nchw_to_c8(var.to(dev))
    with torch.no_grad():
        monkeypatch.setenv("MVS_TAIL_FUSED", "1")
        a = model.cost_regularization.forward_hip(vc8, in_c8=True).cpu()
        monkeypatch.setenv("MVS_TAIL_FUSED", "0")
        b = model.cost_regularization.forward_hip(vc8, in_c8=True).cpu()
    assert ops.guard_fallback_count() >= 0
    scale = float(b.abs().max())
    assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 4e-6 * scale, (float((a - b).abs().max()), scale)
    assert not torch.equal(a, b)      # (two different summation orders: bit equality would mean the switch did nothing)
