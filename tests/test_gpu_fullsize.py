"""Size-independent properties at BASELINE.json's full configs[1] shapes
(feature 296x400, C=32, D=192, 5 views), where the CPU oracle would take a
minute per call: exact scaling laws, agreement of independent HIP implementations,
analytic answers.  Run on the GPU box (-m gpu)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W, C, D, V = 296, 400, 32, 192, 5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from mvs_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def scene(dev):
    from mvs_amd import ops, synth
    g = torch.Generator(device=dev).manual_seed(5)
    feats = torch.randn(V, 1, C, H, W, device=dev, generator=g) * 0.1
    proj = torch.from_numpy(synth.proj_matrices(V, H, W)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D)).to(dev)
    rts = ops.rot_trans_all(proj)
    return feats, rts, dv


def test_variance_two_kernels_agree_bit_for_bit_and_scale_exactly(dev, scene):
    """gather kernel == LDS-staged kernel (independent implementations), and
    var(2 f) == 4 var(f) exactly (power-of-two scaling commutes with every rounding)."""
    from mvs_amd import ops
    feats, rts, dv = scene
    cl = ops.nchw_to_nhwc(feats.reshape(V, C, H, W)).reshape(V, 1, H, W, C)
    v_gather = ops.costvol_variance_cl(cl[0], cl[1:], rts, dv)
    f16 = ops.nchw_to_c16(feats)
    v_lds = ops.costvol_variance_c16(f16[0], f16[1:], rts, dv)       # persistent kernel, 16-channel blocks
    assert torch.equal(v_gather, v_lds)
    f4 = ops.nchw_to_c4(feats)
    assert torch.equal(ops.costvol_variance_c16(f4[0], f4[1:], rts, dv), v_lds)   # 4-channel blocks
    v_fast = ops.costvol_variance_c16(f4[0], f4[1:], rts, dv, fast=True)
    assert float((v_fast - v_lds).abs().max()) < 2e-5     # features ~N(0, 0.1^2): variances ~1e-2
    f16x2 = ops.nchw_to_c16(feats * 2)
    v_lds2 = ops.costvol_variance_c16(f16x2[0], f16x2[1:], rts, dv)
    assert torch.equal(v_lds2, v_lds * 4)
    v8 = ops.costvol_variance_c16(f16[0], f16[1:], rts, dv, out_c8=True)
    assert torch.equal(ops.c8_to_nchw(v8), v_lds.permute(0, 4, 1, 2, 3))
    assert torch.isfinite(v_lds).all() and float(v_lds.min()) > -1e-6
    # identical views under torch-1.2 sampling (align_corners=True) => zero variance
    same = f16[:1].expand(V, -1, -1, -1, -1, -1).contiguous()
    ident = torch.zeros_like(rts)
    ident[:, :, 0] = ident[:, :, 5] = ident[:, :, 10] = 1.0
    v0 = ops.costvol_variance_c16(same[0], same[1:], ident, dv, align_corners=True)
    assert float(v0.abs().max()) < 1e-6


def test_conv0_mfma_matches_direct_kernel_fullsize(dev):
    """MFMA implicit GEMM (shifted Cout=8 form, blocked input) vs the direct VALU
    kernel on the full 192x296x400 volume: tile/edge logic at the real dims."""
    from mvs_amd import ops
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(1, C, 48, H, W, device=dev, generator=g) * 0.1      # a 48-plane slab
    w = torch.randn(8, C, 3, 3, 3, device=dev, generator=g) * 0.05
    sc = torch.rand(8, device=dev, generator=g) + 0.5
    sh = torch.randn(8, device=dev, generator=g) * 0.1
    want = ops.conv3d(ops.nchw_to_nhwc(x), w, sc, sh, None, True, False, 1, channels_last=True,
                      impl=ops.IMPL_DIRECT)
    got = ops.conv3d(ops.nchw_to_c8(x), w, sc, sh, None, True, False, 1,
                     packed=ops.pack_conv3d_weight(w, False, 1), impl=ops.IMPL_MFMA, in_c8=True)
    assert float((got - want).abs().max()) < 2e-5


def test_regress_analytic_answers_fullsize(dev):
    from mvs_amd import ops, synth
    dv = torch.from_numpy(synth.depth_values(D)).to(dev)
    g = torch.Generator(device=dev).manual_seed(3)
    cost = torch.randn(1, D, H, W, device=dev, generator=g)
    k = torch.randint(0, D, (1, 1, H, W), device=dev, generator=g)
    cost.scatter_(1, k, 200.0)                       # all mass on plane k(y,x)
    depth, conf, _ = ops.softmax_regress_conf(cost, dv)
    assert torch.equal(depth, dv[0][k[0, 0]].unsqueeze(0))
    assert float((conf - 1).abs().max()) < 1e-6
    # shift invariance of the softmax
    base = torch.randn(1, D, H, W, device=dev, generator=g) * 3
    d1, c1, _ = ops.softmax_regress_conf(base, dv)
    d2, c2, _ = ops.softmax_regress_conf(base + 17.0, dv)
    assert float((d1 - d2).abs().max()) < 2e-3
    # the confidence window index truncates sum(p*d): it may flip where that sum is
    # within rounding of an integer, so compare the bulk, not the max
    assert float((c1 - c2).abs().median()) < 1e-6
    assert float(d1.min()) >= float(dv.min()) - 1e-3 and float(d1.max()) <= float(dv.max()) + 1e-3
