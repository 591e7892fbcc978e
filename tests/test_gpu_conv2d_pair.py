"""Two consecutive 3x3 stride-1 16 -> 16 layers as one kernel (mvs_amd/csrc/conv2d_pair.hip: conv3 + conv4 of FeatureNet,
MVSNet/models/mvsnet.py:15-16,37-38: ConvBnReLU twice) against the composition in float64 (ATen on the CPU) and against the
two launches of the same library."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _case(seed, N, H, W, mag=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, H, W, 16, generator=g).clamp_min(0) * mag
    w1, w2 = torch.randn(16, 16, 3, 3, generator=g) * 0.1, torch.randn(16, 16, 3, 3, generator=g) * 0.1
    s1, s2 = 0.5 + torch.rand(16, generator=g), 0.5 + torch.rand(16, generator=g)
    b1, b2 = torch.randn(16, generator=g) * 0.1 * mag, torch.randn(16, generator=g) * 0.1 * mag
    return x, w1, s1, b1, w2, s2, b2


def _ref64(x, w1, s1, b1, w2, s2, b2, relu2=True):
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w1.double(), padding=1)
    y = torch.relu(y * s1.double().view(1, 16, 1, 1) + b1.double().view(1, 16, 1, 1))
    z = F.conv2d(y, w2.double(), padding=1) * s2.double().view(1, 16, 1, 1) + b2.double().view(1, 16, 1, 1)
    return (torch.relu(z) if relu2 else z).permute(0, 2, 3, 1)


def _run(dev, case, relu2=True, out_c4=False):
    from mvs_amd import ops
    x, w1, s1, b1, w2, s2, b2 = (t.to(dev) for t in case)
    p1 = dict(packed=ops.pack_conv2d_weight(w1, 1, split=True), scale=s1, shift=b1, relu=True)
    p2 = dict(packed=ops.pack_conv2d_weight(w2, 1, split=True), scale=s2, shift=b2, relu=relu2)
    pair = ops.pack_conv2d_pair(w1, w2)
    xa = ops.absmax(x)
    oa = ops.absmax_block(dev, zero=True)
    got = ops.conv2d_pair(x, xa, pair, p1, p2, out_c4=out_c4, out_absmax=oa)
    ma = ops.absmax_block(dev, zero=True)
    mid = ops.conv2d(x, p1["packed"], 16, 16, 3, 1, s1, b1, True, x_absmax=xa, out_absmax=ma)
    unf = ops.conv2d(mid, p2["packed"], 16, 16, 3, 1, s2, b2, relu2, x_absmax=ma)
    torch.cuda.synchronize()
    if out_c4:
        got = got.permute(0, 2, 3, 1, 4).reshape(unf.shape)
    return got.cpu(), unf.cpu(), ops.absmax_value(oa)


@pytest.mark.parametrize("shape", [(1, 14, 30), (1, 16, 32), (2, 37, 61), (1, 100, 200), (3, 9, 11), (5, 148, 200)],
                         ids=lambda s: "x".join(map(str, s)))
def test_pair_matches_float64_and_the_two_launches(dev, shape):
    """One tile exactly; one pixel more than a tile in both directions; B > 1 with ragged tiles; more tiles than CUs; an image
    smaller than a tile; FeatureNet's own shape at a quarter of config 2."""
    case = _case(sum(shape), *shape)
    ref = _ref64(*case)
    got, unf, amax = _run(dev, case)
    scale = float(ref.abs().max())
    e_f, e_u = float((got.double() - ref).abs().max()), float((unf.double() - ref).abs().max())
    assert torch.isfinite(got).all()
    assert e_f <= max(2.5 * e_u, 2e-6 * scale) and e_f <= 5e-6 * scale, (e_f, e_u, scale)
    assert abs(amax - float(got.abs().max())) <= 1e-6 * scale          # the block the next layer scales by


@pytest.mark.parametrize("mag", [1e-10, 1e-3, 1e5, 1e12], ids=lambda m: f"x{m:g}")
def test_pair_scales_follow_the_data(dev, mag):
    case = _case(11, 1, 40, 70, mag=mag)
    ref = _ref64(*case)
    got, _, _ = _run(dev, case)
    assert float((got.double() - ref).abs().max()) <= 5e-6 * float(ref.abs().max())


def test_pair_without_final_relu_and_blocked_output(dev):
    case = _case(5, 2, 33, 47)
    ref = _ref64(*case, relu2=False)
    got, unf, _ = _run(dev, case, relu2=False, out_c4=True)
    assert (ref < 0).any()
    assert float((got.double() - ref).abs().max()) <= 5e-6 * float(ref.abs().max())


def test_pair_declines_what_the_guard_declines(dev):
    """A NaN pixel in the input: the fused launch sets its flag and the two enqueued layers produce what they always did."""
    from mvs_amd import ops
    case = list(_case(3, 1, 30, 40))
    case[0][0, 7, 9, 3] = float("nan")
    x, w1, s1, b1, w2, s2, b2 = (t.to(dev) for t in case)
    pair = ops.pack_conv2d_pair(w1, w2)
    xa = ops.absmax(x)
    out = torch.empty(1, 30, 40, 16, device=dev)
    flag = torch.zeros(ops.ABSMAX_WORDS + 64, device=dev, dtype=torch.int32)
    import ctypes
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    ops.check(ops._lib.load().mvs_conv2d_pair_f16_f32(ops.ptr(x), vp(xa), vp(pair), ops.ptr(s1), ops.ptr(b1), ops.ptr(s2), ops.ptr(b2), 1,
                                                     1, 16, 30, 40, 0, ops.ptr(out), None, vp(flag), ops.stream()), "pair")
    assert int(flag[0].item()) == 1
