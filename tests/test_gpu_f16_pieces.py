"""The two-piece fp16 convolution kernels (three products per fp32 product; mvs_amd/csrc/conv_f16x3.hip, conv_split.hip and
deconv_split.hip with NP = 2) and the absmax blocks that carry their operand scales from layer to layer.

What they replace is arithmetic of the reference in float32 (CostRegNet, mvsnet.py:48-93; FeatureNet, mvsnet.py:8-45), so the
bar is the distance from a float64 evaluation of the same layer: no farther than the three-piece bf16 kernels and than ATen's
float32 convolution (the reference's own arithmetic) within a small factor, whatever the magnitude of the data."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mvs_amd import _lib
    _lib.load()  # must exist: no fallback
    return torch.device("cuda:0")


def _err(y, ref):
    d = (y.detach().double().cpu() - ref).abs()
    return d.max().item(), d.pow(2).mean().sqrt().item()


@pytest.mark.parametrize("shape,cin,mag", [((1, 5, 9, 21), 32, 1.0), ((2, 7, 6, 37), 16, 1.0), ((1, 17, 30, 70), 8, 1.0),
                                           ((1, 12, 20, 70), 32, 1e-20), ((1, 12, 20, 70), 32, 3e9), ((1, 4, 4, 32), 32, 1.0),
                                           ((1, 40, 12, 33), 32, 1.0)])
def test_conv3d_f16x3_vs_float64(dev, shape, cin, mag):
    """mvs_conv3d_c8_f16x3_f32 against a float64 convolution: as close as the bf16 six-product kernel and ATen's float32
    convolution, on ragged tiles of every axis, a last group of one to three tiles, batch 2, all supported channel counts,
    magnitudes from 1e-20 to 3e9 (the operand scale follows the data), with affine, ReLU and skip add; the absmax block
    of the output holds exactly the largest magnitude written."""
    from mvs_amd import ops
    B, D, H, W = shape
    g = torch.Generator().manual_seed(cin * 1000 + D * 10 + W)
    x = (torch.randn(B, cin, D, H, W, generator=g) * torch.rand(B, cin, D, H, W, generator=g) ** 4).square() * mag
    w = torch.randn(8, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1 * mag
    res = torch.randn(B, D, H, W, 8, generator=g) * mag
    v = lambda t: t.view(1, 8, 1, 1, 1)
    ref = (torch.relu(F.conv3d(x.double(), w.double(), padding=1) * v(scale.double()) + v(shift.double()))).permute(0, 2, 3, 4, 1) + res.double()
    aten = (torch.relu(F.conv3d(x, w, padding=1) * v(scale) + v(shift))).permute(0, 2, 3, 4, 1) + res
    wt, x8 = w.to(dev), ops.nchw_to_c8(x.to(dev))
    om = ops.absmax_block(dev, zero=True)
    got = ops.conv3d_c8_f16x3(x8, ops.pack_conv3d_weight_f16x3(wt), None, scale.to(dev), shift.to(dev), res.to(dev), True, out_absmax=om)
    six = ops.conv3d_c8_split(x8, ops.pack_conv3d_weight_split(wt), scale.to(dev), shift.to(dev), res.to(dev), True)
    (e2, r2), (e3, r3), (ea, ra) = _err(got, ref), _err(six, ref), _err(aten, ref)
    assert r2 <= 1.2 * r3 + 1e-9 * mag and r2 <= 2.0 * ra + 1e-9 * mag, (r2, r3, ra)
    assert e2 <= 2.0 * max(e3, ea) + 1e-8 * mag, (e2, e3, ea)
    assert ops.absmax_value(om) == got.abs().max().item()


def test_conv3d_f16x3_scale_follows_the_producer_block(dev):
    """The operand scale comes from the absmax block alone: a block that overstates the maximum by 2^6 (a loose bound) costs
    nothing visible, one collected by mvs_absmax_f32 and one handed over by the variance op give the same bits; zeros in,
    the shift out."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(1, 32, 6, 10, 40, generator=g) * torch.rand(1, 32, 6, 10, 40, generator=g) ** 4).square()
    w = (torch.randn(8, 32, 3, 3, 3, generator=g) / 30).to(dev)
    x8 = ops.nchw_to_c8(x.to(dev))
    pf = ops.pack_conv3d_weight_f16x3(w)
    ref = F.conv3d(x.double(), w.double().cpu(), padding=1).permute(0, 2, 3, 4, 1)
    blk = ops.absmax(x8)
    assert ops.absmax_value(blk) == x.max().item()
    a = ops.conv3d_c8_f16x3(x8, pf, blk)
    loose = ops.absmax_block(dev, zero=True)
    loose[3] = torch.tensor(x.max().item() * 64.0).view(torch.int32)
    b = ops.conv3d_c8_f16x3(x8, pf, loose)
    assert _err(b, ref)[1] <= 1.3 * _err(a, ref)[1]
    assert torch.equal(a, ops.conv3d_c8_f16x3(x8, pf, None))
    z = ops.conv3d_c8_f16x3(torch.zeros_like(x8), pf, None, None, torch.arange(8.0, device=dev), None, False)
    assert torch.equal(z, torch.arange(8.0, device=dev).expand_as(z))


@pytest.mark.parametrize("kd,cin,cout,shape,k,stride,relu", [
    (3, 16, 16, (1, 5, 9, 21), 3, 1, 1), (3, 32, 32, (2, 6, 7, 33), 3, 1, 1), (3, 64, 64, (1, 4, 10, 18), 3, 1, 0),
    (3, 16, 32, (1, 9, 17, 40), 3, 1, 1), (3, 64, 32, (1, 3, 5, 16), 3, 1, 0), (3, 8, 16, (1, 9, 20, 37), 3, 2, 1),
    (3, 8, 32, (2, 6, 9, 35), 3, 1, 0),
    (3, 16, 32, (1, 8, 11, 35), 3, 2, 1), (3, 32, 64, (1, 6, 9, 18), 3, 2, 1),
    (1, 16, 16, (3, 21, 45), 3, 1, 2), (1, 32, 32, (2, 40, 70), 3, 1, 1), (1, 64, 64, (1, 18, 50), 3, 1, 2),
    (1, 8, 16, (2, 37, 70), 5, 2, 1), (1, 16, 32, (2, 37, 70), 5, 2, 1)])
def test_conv_split_f16_vs_float64(dev, kd, cin, cout, shape, k, stride, relu):
    """mvs_conv_split_f16_f32 on every shape class of mvs_conv_split_f32 (3D and 2D, stride 1 and 2, 5x5, one and two
    launches per layer, all activations, skip add): no farther from float64 than the bf16 form; out_absmax exact."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(kd * 1000 + cin * 10 + cout + shape[-1])
    dims = shape[1:]
    x = torch.randn(shape[0], cin, *dims, generator=g).clamp_min(0) * torch.rand(shape[0], cin, *dims, generator=g) ** 2 * 3.0
    w = torch.randn(cout, cin, *([3] * (kd == 3)), k, k, generator=g) / (k * k * (3 if kd == 3 else 1) * cin) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    conv = F.conv3d if kd == 3 else F.conv2d
    pin, pout = ((0, 2, 3, 4, 1), (1, cout, 1, 1, 1)) if kd == 3 else ((0, 2, 3, 1), (1, cout, 1, 1))
    y = conv(x.double(), w.double(), stride=stride, padding=k // 2) * scale.double().view(pout) + shift.double().view(pout)
    y = torch.relu(y) if relu == 1 else (F.leaky_relu(y, 0.1) if relu == 2 else y)
    res = torch.randn(y.permute(pin).shape, generator=g) if (kd == 3 or stride == 1) and relu != 2 else None
    ref = y.permute(pin) + (res.double() if res is not None else 0)
    xd, wd = x.permute(pin).contiguous().to(dev), w.to(dev)
    rd = res.to(dev) if res is not None else None
    om = ops.absmax_block(dev, zero=True)
    got = ops.conv_split_f16(xd, ops.pack_conv_weight_split_f16(wd, stride), cout, None, scale.to(dev), shift.to(dev), rd, relu,
                             kd=kd, stride=stride, out_absmax=om)
    six = ops.conv_split(xd, ops.pack_conv_weight_split(wd, stride), cout, scale.to(dev), shift.to(dev), rd, relu, kd=kd, stride=stride)
    (e2, r2), (e3, r3) = _err(got, ref), _err(six, ref)
    assert r2 <= 1.2 * r3 + 1e-9 and e2 <= 2.0 * e3 + 1e-8, (e2, e3, r2, r3)
    assert ops.absmax_value(om) == got.abs().max().item()


@pytest.mark.parametrize("cin,cout,shape", [(16, 8, (1, 5, 9, 21)), (32, 16, (2, 3, 6, 17)), (64, 32, (1, 4, 5, 16)), (16, 8, (1, 8, 12, 40))])
def test_deconv_split_f16_vs_float64(dev, cin, cout, shape):
    """mvs_deconv_split_f16_f32 (conv7 / conv9 / conv11 classes) against a float64 transposed convolution with the skip added
    after the ReLU (mvsnet.py:89-91): no farther than the bf16 form; out_absmax exact."""
    from mvs_amd import ops
    B, D, H, W = shape
    g = torch.Generator().manual_seed(cin * 10 + cout + W)
    x = torch.randn(B, cin, D, H, W, generator=g).clamp_min(0) * 2.0
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) / (27 * cin / 8) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, 2 * D, 2 * H, 2 * W, cout, generator=g)
    v = lambda t: t.view(1, cout, 1, 1, 1)
    ref = torch.relu(F.conv_transpose3d(x.double(), w.double(), stride=2, padding=1, output_padding=1) * v(scale.double())
                     + v(shift.double())).permute(0, 2, 3, 4, 1) + res.double()
    xd, wd = x.permute(0, 2, 3, 4, 1).contiguous().to(dev), w.to(dev)
    om = ops.absmax_block(dev, zero=True)
    got = ops.deconv_split_f16(xd, ops.pack_deconv_weight_split_f16(wd), cout, None, scale.to(dev), shift.to(dev), res.to(dev), True, out_absmax=om)
    six = ops.deconv_split(xd, ops.pack_deconv_weight_split(wd), cout, scale.to(dev), shift.to(dev), res.to(dev), True)
    (e2, r2), (e3, r3) = _err(got, ref), _err(six, ref)
    assert r2 <= 1.2 * r3 + 1e-9 and e2 <= 2.0 * e3 + 1e-8, (e2, e3, r2, r3)
    assert ops.absmax_value(om) == got.abs().max().item()


@pytest.mark.parametrize("n", [1, 3, 4, 1000, 4099, 1 << 20])
def test_absmax_block_of_an_array(dev, n):
    """mvs_absmax_f32: the largest magnitude of n floats (negative values, ragged tails, a NaN wins)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n + 4, generator=g).to(dev)[:n] * 7
    x = x.clone()
    assert ops.absmax_value(ops.absmax(x)) == x.abs().max().item()
    if n > 3:
        x[n // 2] = float("nan")
        assert np.isnan(ops.absmax_value(ops.absmax(x)))


@pytest.mark.parametrize("forced", ["", "16", "8", "0"])
def test_variance_collects_its_absmax_block(forced):
    """mvs_costvol_variance_fwd_ws2_f32: the block holds exactly max |volume| whichever sweep kernel runs (device-selected
    and each forced choice: own process, the switch is read once), and the volume is bit-identical to the call without it."""
    import os, subprocess, sys
    code = r'''
import torch, numpy as np
from mvs_amd import ops, synth
dev = torch.device("cuda:0")
V, h, w, D = 4, 40, 64, 24
g = torch.Generator().manual_seed(1)
f = (torch.randn(V, 1, 8, h, w, 4, generator=g) * 2.5).to(dev)
rts = ops.rot_trans_all(torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev), "device")
dv = torch.from_numpy(synth.depth_values(D, interval=synth.sweep_interval(D))).to(dev)
blk = ops.absmax_block(dev)
blk.fill_(0x7f000000)          # stale contents: the call resets the block
f16 = f.reshape(V, 1, 2, 4, h, w, 4).permute(0, 1, 2, 4, 5, 3, 6).reshape(V, 1, 2, h, w, 16).contiguous()
import os
fea = f16 if os.environ.get("MVS_SWEEP_PERSIST") == "0" else f      # (4-channel blocks are the persistent kernel's layout)
a = ops.costvol_variance_c16(fea[0], fea[1:], rts, dv, out_c8=True, fast=True, absmax_out=blk)
b = ops.costvol_variance_c16(fea[0], fea[1:], rts, dv, out_c8=True, fast=True)
assert torch.equal(a, b)
assert ops.absmax_value(blk) == a.abs().max().item() and a.abs().max().item() > 0
c = ops.costvol_variance_c16(f16[0], f16[1:], rts, dv, out_c8=False, absmax_out=blk)      # 16-channel blocks: exact coordinates
assert ops.absmax_value(blk) == c.abs().max().item()
dvp = dv.view(1, D, 1, 1).expand(1, D, h, w).contiguous()           # per-pixel planes: the per-tile kernel
d = ops.costvol_variance_c16(f16[0], f16[1:], rts, dvp, out_c8=True, absmax_out=blk)
assert ops.absmax_value(blk) == d.abs().max().item()
print("OK")
'''
    env = dict(os.environ)
    if forced:
        env["MVS_SWEEP_PERSIST"] = forced
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_costreg_one_call_equals_the_layer_chain(dev):
    """mvs_costreg_fwd2_f32 (absmax blocks chained through the workspace) returns the bits of the per-layer chain of
    CostRegNet.forward_hip under a stage timer (ops.conv3d with x_absmax / out_absmax), with and without the block of the
    input handed over; and lies within the float32 layers' distance of the reference modules' own forward."""
    from mvs_amd import ops
    from mvs_amd.models.mvsnet import CostRegNet
    torch.manual_seed(3)
    net = CostRegNet().to(dev).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.running_var.uniform_(0.5, 1.5); m.running_mean.normal_(0, 0.1); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    g = torch.Generator(device=dev).manual_seed(2)
    x = (torch.randn(1, 32, 16, 24, 40, device=dev, generator=g) * torch.rand(1, 32, 16, 24, 40, device=dev, generator=g) ** 3).square()
    x8 = ops.nchw_to_c8(x)
    with torch.no_grad():
        one = net.forward_hip(x8, in_c8=True)
        one_blk = net.forward_hip(x8, in_c8=True, x_absmax=ops.absmax(x8))
        t = ops.StageTimer()
        ops.set_timer(t)
        try:
            chain = net.forward_hip(x8, in_c8=True)
        finally:
            ops.set_timer(None)
        ref = net(x).squeeze(1)
    assert torch.equal(one, one_blk) and torch.equal(one, chain)
    assert (one - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape,cin", [((1, 12, 20, 70), 32), ((2, 7, 9, 37), 16), ((1, 5, 30, 33), 8), ((1, 17, 6, 64), 32), ((1, 4, 4, 32), 32),
                                       ((1, 40, 12, 31), 32)])
@pytest.mark.skipif(__import__("os").environ.get("MVS_HIP_TUNING") != "1", reason="the pre-split conv0 experiment lives in the tuning build "
                    "(python -m mvs_amd.build --tuning; MVS_HIP_TUNING=1 python -m pytest ...)")
def test_conv0_on_presplit_pairs_is_bit_identical(dev, shape, cin):
    """mvs_conv3d_c8h_f16x3_f32 (the volume arrives as scaled fp16 pairs, MVS_LAYOUT_C8H: no staging buffer, no split pass) returns
    the bits of mvs_conv3d_c8_f16x3_f32 on the fp32 volume under the same absmax block: odd and even widths, single-tile groups
    (the 18-slot ring's worst case), batch 2, all channel counts, affine + ReLU + skip add."""
    from mvs_amd import ops
    B, D, H, W = shape
    g = torch.Generator().manual_seed(D * 7 + W)
    x = ((torch.randn(B, D, H, cin // 8, W, 8, generator=g) * torch.rand(B, D, H, cin // 8, W, 8, generator=g) ** 4).square()).to(dev)
    w = (torch.randn(8, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).to(dev)
    sc, sh = (torch.rand(8, generator=g) + 0.5).to(dev), (torch.randn(8, generator=g) * 0.1).to(dev)
    r = torch.randn(B, D, H, W, 8, generator=g).to(dev)
    pf = ops.pack_conv3d_weight_f16x3(w)
    blk = ops.absmax(x)
    a = ops.conv3d_c8_f16x3(x, pf, blk, sc, sh, r, True)
    om = ops.absmax_block(dev, zero=True)
    b = ops.conv3d_c8h_f16x3(ops.c8_to_c8h(x, blk), (B, cin, D, H, W), pf, blk, sc, sh, r, True, out_absmax=om)
    assert torch.equal(a, b)
    assert ops.absmax_value(om) == b.abs().max().item()


@pytest.mark.parametrize("env", [{"MVS_CONV0_F16": "0"}, {"MVS_SPLIT_F16": "0"}, {"MVS_CONV_SPLIT": "0"}])
def test_arithmetic_switches_stay_within_the_gate(env):
    """The A/B switches of the convolution arithmetic (three-piece bf16 conv0; three-piece bf16 everywhere; fp32 MFMA kernels)
    run the same network: an MVSNet eval forward under each differs from the default build's by far less than the 1e-3 mm gate
    (own process: the switches are read when the weights are packed)."""
    import os, subprocess, sys
    code = r'''
import sys, numpy as np, torch
from mvs_amd import synth
from mvs_amd.models import MVSNet
dev = torch.device("cuda:0")
H, W, V, D = 128, 160, 3, 16
rng = np.random.default_rng(3)
model = MVSNet(refine=False); model.load_state_dict(synth.random_state_dict(2)); model = model.to(dev).eval()
g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
with torch.no_grad():
    out = model(g(synth.images(rng, 1, V, H, W)), g(synth.proj_matrices(V, H // 4, W // 4)),
                g(synth.depth_values(D, interval=synth.sweep_interval(D))))
np.save(sys.argv[1], out["depth"].cpu().numpy())
'''
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        outs = []
        for e in ({}, env):
            f = os.path.join(tmp, "d%d.npy" % len(outs))
            r = subprocess.run([sys.executable, "-c", code, f], env={**os.environ, **e}, capture_output=True, text=True, cwd=root)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
            outs.append(np.load(f))
    assert np.isfinite(outs[1]).all()
    assert np.abs(outs[0] - outs[1]).max() < 5e-4, np.abs(outs[0] - outs[1]).max()
