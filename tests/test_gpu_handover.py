"""The variance volume handed from the sweep to conv0 as two fp16 pieces per value (round 6; include/mvs_hip.h:
mvs_costvol_variance_fwd_ws3_f32, mvs_conv3d_c8p_f16x3_f32, mvs_conv3d_c8_handed_f16x3_f32, mvs_costreg_fwd4_f32).

The operation is the reference's `volume_variance` feeding `conv0` (MVSNet/models/mvsnet.py:152-181, 57, 83): what is checked
here is that the hand-over changes NOTHING but the storage format -- the pieces decode to the fp32 sweep's volume within the
two-piece bound, conv0 on them returns the bits of the fp32-volume kernel under the same scale, and every case the pieces
cannot carry (per-tile kernel chosen, non-finite maps, non-finite conv0 weights, a bound far above the true maximum, an
outlier-dominated volume) comes out as the fp32 volume on the device's own decision, with no host synchronisation."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _to_pairs(x, blk, layout):
    from mvs_amd import _lib
    lib = _lib.load()
    B, D, H, G, W, _ = x.shape
    out = torch.zeros(lib.mvs_c8p_bytes(B, G * 8, D, H, W, layout), device=x.device, dtype=torch.uint8)
    _lib.check(lib.mvs_c8_to_c8p_f32(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(blk.data_ptr()), B, G * 8, D, H, W, layout,
                                     ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "mvs_c8_to_c8p_f32")
    return out


def _conv_pairs(xp, blk, redo, pf, sc, sh, r, relu, shape, layout, om=None):
    from mvs_amd import _lib
    B, cin, D, H, W = shape
    out = torch.full((B, D, H, W, 8), -7.0, device=xp.device)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    _lib.check(_lib.load().mvs_conv3d_c8p_f16x3_f32(vp(xp), vp(blk), vp(redo), vp(pf), vp(sc), vp(sh), vp(r), int(relu), B, cin, D, H, W,
                                                    layout, 4, vp(out), vp(om), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "mvs_conv3d_c8p_f16x3_f32")
    return out


@pytest.mark.parametrize("shape,cin", [((1, 12, 20, 70), 32), ((2, 7, 9, 37), 16), ((1, 5, 30, 33), 8), ((1, 33, 17, 64), 32), ((1, 2, 8, 32), 32),
                                       ((1, 1, 3, 5), 8), ((2, 19, 41, 100), 32)])
@pytest.mark.parametrize("layout", [6, 7, 8], ids=["rows", "xtiled", "strips"])
def test_conv0_on_pieces_is_bit_identical(dev, shape, cin, layout):
    """mvs_conv3d_c8p_f16x3_f32 (eight-row tiles, copies straight into the plane ring: no staging buffer, no split pass, one barrier
    per step) returns the bits of mvs_conv3d_c8_f16x3_f32 on the fp32 volume under the same absmax block: ragged sizes in every
    dimension, one-plane and one-tile volumes, batch 2, all channel counts, affine + ReLU + skip add, all three piece layouts; and does
    nothing when its redo word is set."""
    from mvs_amd import ops
    B, D, H, W = shape
    g = torch.Generator().manual_seed(D * 7 + W)
    x = ((torch.randn(B, D, H, cin // 8, W, 8, generator=g) * torch.rand(B, D, H, cin // 8, W, 8, generator=g) ** 4).square()).to(dev)
    w = (torch.randn(8, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).to(dev)
    sc, sh = (torch.rand(8, generator=g) + 0.5).to(dev), (torch.randn(8, generator=g) * 0.1).to(dev)
    r = torch.randn(B, D, H, W, 8, generator=g).to(dev)
    pf = ops.pack_conv3d_weight_f16x3(w)
    blk = ops.absmax(x)
    blk[:] = blk.max() + (3 << 23)          # a scale three binary orders above the true maximum, as a bound would be
    a = ops.conv3d_c8_f16x3(x, pf, blk, sc, sh, r, True)
    om = ops.absmax_block(dev, zero=True)
    xp = _to_pairs(x, blk, layout)
    zero, one = torch.zeros(1, device=dev, dtype=torch.int32), torch.ones(1, device=dev, dtype=torch.int32)
    b = _conv_pairs(xp, blk, zero, pf, sc, sh, r, True, (B, cin, D, H, W), layout, om)
    assert torch.equal(a, b)
    assert ops.absmax_value(om) == b.abs().max().item()
    c = _conv_pairs(xp, blk, one, pf, sc, sh, r, True, (B, cin, D, H, W), layout)
    assert bool((c == -7.0).all())          # *redo != 0: the launch returns at once


def _scene(dev, V=4, h=40, w=64, D=24, seed=1, mag=2.5, batch=1):
    from mvs_amd import ops, synth
    g = torch.Generator().manual_seed(seed)
    f = (torch.randn(V, batch, 8, h, w, 4, generator=g) * mag).to(dev)
    P = torch.from_numpy(np.repeat(synth.proj_matrices(V, h, w), batch, 0)).to(dev)
    rts = ops.rot_trans_all(P, "device")
    dv = torch.from_numpy(np.repeat(synth.depth_values(D), batch, 0)).to(dev)     # DTU's interval: footprints the persistent kernel takes
    return f, rts, dv


def _decode_check(hv, ref, bound):
    """pieces -> fp32: within 2^-22 relative of the value, or 2^-39 of the scale's top absolute (fp16 subnormal pieces)"""
    got = hv.to_c8().double()
    top = 2.0 ** (np.floor(np.log2(bound)) + 1)
    err = (got - ref.double()).abs()
    lim = ref.double().abs() * 2.0 ** -21.5 + top * 2.0 ** -39
    assert bool((err <= lim).all()), (float(err.max()), float((err / lim).max()))


@pytest.mark.parametrize("case", [dict(), dict(V=2, h=33, w=47, D=9), dict(V=5, h=24, w=100, D=40, batch=2), dict(V=3, h=16, w=32, D=192)],
                         ids=["4v", "ragged", "batch2", "d192"])
@pytest.mark.parametrize("fast", [True, False], ids=["fast", "exact"])
def test_sweep_hands_the_volume_over_as_pieces(dev, case, fast):
    """mvs_costvol_variance_fwd_ws3_f32 against the fp32 sweep (mvs_costvol_variance_fwd_ws2_f32) on the same maps: redo = 0, the
    hand-over block = max|f|^2, the absmax block = the fp32 volume's true maximum exactly, the pieces decode to the fp32 volume,
    and conv0 on whatever was left = mvs_conv3d_c8_f16x3_f32 on the fp32 volume under the bound as its scale, bit for bit."""
    from mvs_amd import ops
    f, rts, dv = _scene(dev, **case)
    fa = ops.absmax(f)
    blk = ops.absmax_block(dev)
    ref = ops.costvol_variance_c16(f[0], f[1:], rts, dv, out_c8=True, fast=fast, absmax_out=blk)
    hv = ops.costvol_variance_handover(f[0], f[1:], rts, dv, fa, fast=fast)
    assert hv is not None
    torch.cuda.synchronize()
    assert int(hv.redo[0].item()) == 0
    fmax = float(f.abs().max())
    bound = np.float32(fmax) * np.float32(fmax)
    assert ops.absmax_value(hv.hand) == float(bound)
    assert ops.absmax_value(hv.absmax) == ops.absmax_value(blk) == float(ref.abs().max())
    _decode_check(hv, ref, float(bound))
    # the halo strips repeat the tile-border columns: x = 32 t - 1 (side 0) and x = 32 t + 32 (side 1), the same piece bits
    main, halo = hv.pieces()
    W = ref.shape[4]
    for t in range((W + 31) // 32):
        for side, x in ((0, 32 * t - 1), (1, 32 * t + 32)):
            if 0 <= x < W:
                assert torch.equal(halo[:, :, :, :, t, side], main[:, :, :, :, x & 1, :, x >> 1]), (t, side)
    # conv0
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(8, 32, 3, 3, 3, generator=g) / (27 * 32) ** 0.5).to(dev)
    sc, sh = (torch.rand(8, generator=g) + 0.5).to(dev), (torch.randn(8, generator=g) * 0.1).to(dev)
    pf = ops.pack_conv3d_weight_f16x3(w)
    om = ops.absmax_block(dev, zero=True)
    got = ops.conv3d_c8_handed(hv, pf, sc, sh, True, out_absmax=om)
    # (test_conv0_on_pieces_is_bit_identical: conv0 on pieces(x, block) = the fp32-volume kernel on x under that block)
    want = ops.conv3d_c8_f16x3(ref, pf, hv.hand, sc, sh, None, True)
    assert torch.equal(got, want)
    assert ops.absmax_value(om) == float(got.abs().max())
    # against the fp32 volume under its own scale: two roundings of the same operands
    own = ops.conv3d_c8_f16x3(ref, pf, blk, sc, sh, None, True)
    assert float((got - own).abs().max()) <= 3e-6 * float(own.abs().max())


def _referee(dev, f, rts, dv, fa=None, veto=None, fast=True):
    from mvs_amd import ops
    fa = ops.absmax(f) if fa is None else fa
    blk = ops.absmax_block(dev)
    ref = ops.costvol_variance_c16(f[0], f[1:], rts, dv, out_c8=True, fast=fast, absmax_out=blk)
    hv = ops.costvol_variance_handover(f[0], f[1:], rts, dv, fa, fast=fast, veto=veto)
    torch.cuda.synchronize()
    return hv, ref, blk


def _same(a, b):
    return torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, nan=0.0), torch.nan_to_num(b, nan=0.0))


@pytest.mark.poisoned_inputs
@pytest.mark.parametrize("poison", ["nan_map", "inf_map", "veto", "loose_bound", "outlier"])
def test_referee_keeps_the_volume_fp32_when_pieces_cannot_carry_it(dev, poison):
    """The cases the hand-over declines, each decided on the device: non-finite feature maps (no finite bound), non-finite conv0
    weights (the reader's veto word), a bound more than 8 binary orders above the volume's true maximum, a volume whose maximum is
    carried by a few outliers (conv_guard.h's rule).  redo = 1, the buffer holds the fp32 volume -- the fp32 sweep's own bits --
    and conv0 on it is mvs_conv3d_c8_f16x3_f32's result with ITS guard's semantics (the reference's NaN / Inf pattern)."""
    from mvs_amd import ops
    f, rts, dv = _scene(dev)
    fa, veto = None, None
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(8, 32, 3, 3, 3, generator=g) / (27 * 32) ** 0.5).to(dev)
    if poison == "nan_map":
        f[2, 0, 3, 11, 17, 2] = float("nan")
    elif poison == "inf_map":
        f[0, 0, 1, 5, 5, 0] = float("inf")
    elif poison == "veto":
        w[3, 7, 1, 1, 1] = float("nan")
    elif poison == "loose_bound":
        fa = ops.absmax(f * 40.0)            # bound 1600x the honest one: > 2^8 above the true maximum
    else:
        f[1, 0, 0, 20, 30, 1] = 3.0e5        # one texel: the volume's maximum sits in a handful of voxels
    pf = ops.pack_conv3d_weight_f16x3(w)
    if poison == "veto":
        veto = ops.conv0_veto_word(pf, 32)
    before = ops.guard_fallback_count()
    hv, ref, blk = _referee(dev, f, rts, dv, fa, veto)
    assert hv is not None and int(hv.redo[0].item()) == 1
    vol = hv.to_c8()
    assert _same(vol, ref)
    assert hv.absmax.max().item() == blk.max().item()
    got = ops.conv3d_c8_handed(hv, pf, None, None, True)
    want = ops.conv3d_c8_f16x3(ref, pf, blk, None, None, None, True)
    assert _same(got, want)
    if poison in ("nan_map", "inf_map", "veto", "outlier"):
        assert ops.guard_fallback_count() > before        # (the fp32 conv0's own guard took it from there)


def test_per_tile_choice_leaves_fp32(dev):
    """A geometry whose footprints outgrow LDS (x4 depth interval on a wide rig): the chooser takes the per-tile kernel, which
    writes fp32 -- redo = 1 without any recomputation, the volume is the fp32 sweep's bits."""
    from mvs_amd import ops, synth
    V, h, w, D = 5, 128, 160, 48
    g = torch.Generator().manual_seed(3)
    f = torch.randn(V, 1, 8, h, w, 4, generator=g).to(dev)
    rts = ops.rot_trans_all(torch.from_numpy(synth.proj_matrices(V, h, w, rig=1)).to(dev), "device")
    dv = torch.from_numpy(synth.depth_values(D, interval=synth.sweep_interval(D) * 4)).to(dev)
    hv, ref, blk = _referee(dev, f, rts, dv)
    assert hv is not None
    from mvs_amd.ops import _variance_ws
    ws = next(iter(_variance_ws.values()))
    choice = int(ws[4:8].view(torch.int32).item())
    if choice != 0:
        pytest.skip(f"the chooser took {choice}-plane tiles for this geometry")
    assert int(hv.redo[0].item()) == 1 and torch.equal(hv.to_c8(), ref)


def test_costregnet_and_mvsnet_through_the_hand_over(dev):
    """mvs_costreg_fwd4_f32 = the per-layer chain on the handed volume (stage timer) bit for bit; MVSNet's eval forward with and
    without the hand-over (MVS_HANDOVER, read per call) agrees far inside the 1e-3 mm gate and is not the same bits (the switch
    does something); the guard stayed silent."""
    import os
    from mvs_amd import ops, synth
    from mvs_amd.models import MVSNet
    model = MVSNet(refine=False)
    model.load_state_dict(synth.random_state_dict(3), strict=False)
    model = model.to(dev).eval()
    V, H, W, D = 4, 256, 320, 192
    g = torch.Generator(device=dev).manual_seed(2)
    imgs = torch.rand(1, V, 3, H, W, device=dev, generator=g)
    proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D)).to(dev)
    before = ops.guard_fallback_count()
    with torch.no_grad():
        a = model(imgs, proj, dv)
        t = ops.StageTimer()
        ops.set_timer(t)
        try:
            chain = model(imgs, proj, dv)
        finally:
            ops.set_timer(None)
        os.environ["MVS_HANDOVER"] = "0"
        try:
            b = model(imgs, proj, dv)
        finally:
            del os.environ["MVS_HANDOVER"]
    assert torch.equal(a["depth"], chain["depth"]) and torch.equal(a["photometric_confidence"], chain["photometric_confidence"])
    d = float((a["depth"] - b["depth"]).abs().max())
    assert 0.0 < d < 2e-4, d
    assert ops.guard_fallback_count() == before


def test_entry_points_reject_what_they_cannot_take(dev):
    """Per-pixel hypotheses and a forced kernel have no hand-over (the Python wrapper returns None, the C entry MVS_EUNSUPPORTED with
    nothing launched); mvs_costreg_fwd4_f32 needs conv0's two-piece pack."""
    from mvs_amd import _lib, ops
    f, rts, dv = _scene(dev)
    fa = ops.absmax(f)
    dvp = dv.view(1, -1, 1, 1).expand(1, dv.shape[1], 40, 64).contiguous()
    assert ops.costvol_variance_handover(f[0], f[1:], rts, dvp, fa) is None
    lib = _lib.load()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    buf = torch.empty(lib.mvs_costvol_variance_handover_bytes(1, 32, 24, 40, 64), device=dev, dtype=torch.uint8)
    words = torch.zeros(3 * 256, device=dev, dtype=torch.int32)
    rc = lib.mvs_costvol_variance_fwd_ws3_f32(vp(f[0]), vp(f[1:]), vp(rts), vp(dv), 1, 4, 32, 24, 40, 64, 0, _lib.MVS_LAYOUT_C4, 1, vp(fa), None,
                                              vp(buf), None, 0, vp(words[:256]), vp(words[256:512]), vp(words[512:]),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -4      # MVS_EWORKSPACE: the hand-over rides on the workspace entry
    rc = lib.mvs_costvol_variance_fwd_ws3_f32(vp(f[0]), vp(f[1:]), vp(rts), vp(dv), 1, 4, 32, 24, 40, 64, 0, _lib.MVS_LAYOUT_C4, 1, None, None,
                                              vp(buf), None, 0, vp(words[:256]), vp(words[256:512]), vp(words[512:]),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -1 and b"absmax" in lib.mvs_last_error_string()


@pytest.mark.parametrize("geom", [dict(h=40, w=64, D=24, rig=0, ac=0), dict(h=74, w=100, D=48, rig=1, ac=0), dict(h=128, w=160, D=192, rig=2, ac=0),
                                  dict(h=74, w=100, D=48, rig=1, ac=1)],
                         ids=["small", "rig1", "rig2_d192", "rig1_align_corners"])
def test_fast_mode_samples_at_the_reference_coordinates(dev, geom):
    """(round 6) The persistent sweep's FAST mode keeps the reference's sampling coordinates bit for bit (module.py:66-84: its four
    divisions through shared / precomputed refined reciprocals); only the variance arithmetic differs (FMA accumulation of the
    squares, multiplication by 1/V).  With two views and a zero reference map the variance is w^2 / 4 in both arithmetics
    EXACTLY (halving and quartering are exact), so the two modes must agree to the bit -- which they do iff every warped value
    w, i.e. every tap and weight, is the same.  Until round 5 FAST had its own coordinate arithmetic (~3e-5 texel away) and a
    trained network turned that into 7e-3 mm of depth (profiles/r06_trained_budget_switches.json)."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from mvs_amd import ops, synth
dev = torch.device("cuda:0")
h, w, D, rig = %(h)d, %(w)d, %(D)d, %(rig)d
g = torch.Generator().manual_seed(7)
f = (torch.randn(2, 1, 8, h, w, 4, generator=g) * 3).to(dev)
f[0] = 0
rts = ops.rot_trans_all(torch.from_numpy(synth.proj_matrices(3, h, w, rig=rig)[:, [0, 2]]).to(dev), "device")
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
a = ops.costvol_variance_c16(f[0], f[1:], rts, dv, align_corners=bool(%(ac)d), out_c8=True, fast=True)
b = ops.costvol_variance_c16(f[0], f[1:], rts, dv, align_corners=bool(%(ac)d), out_c8=True, fast=False)
assert float(a.abs().max()) > 0 and bool((a == 0).any())        # in-image and out-of-image samples both occur
assert torch.equal(a, b), float((a - b).abs().max())
print("OK")
''' % geom
    # (own process with the persistent kernel forced: the comparison must not depend on which kernel the chooser takes)
    env = dict(os.environ, MVS_SWEEP_PERSIST="16")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_eval_forward_captures_into_a_hip_graph_in_the_default_capture_mode(dev):
    """A user's `torch.cuda.graph(g)` around the eval forward (capture_error_mode='global', torch's default): nothing in the path --
    the hand-over's blocks, the sweep chooser's asynchronous verdict read-back (an event query during capture), the
    workspaces -- makes an illegal call while capturing, and a replay returns the eager forward's bits.  `proj_where="device"`:
    the reference's float32 LAPACK inverse is a host hop, which no graph can hold."""
    from mvs_amd import synth
    from mvs_amd.models import MVSNet
    model = MVSNet(refine=False)
    model.load_state_dict(synth.random_state_dict(5), strict=False)
    model = model.to(dev).eval()
    model.proj_where = "device"
    V, H, W, D = 3, 256, 320, 192
    g = torch.Generator(device=dev).manual_seed(4)
    imgs = torch.rand(1, V, 3, H, W, device=dev, generator=g)
    proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D)).to(dev)
    with torch.no_grad():
        for _ in range(9):                 # (past the first read-back of the chooser's verdict: one is pending or done)
            eager = model(imgs, proj, dv)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(imgs, proj, dv)          # the side stream's workspaces exist before the capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = model(imgs, proj, dv)
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(out["depth"], eager["depth"]) and torch.equal(out["photometric_confidence"], eager["photometric_confidence"])
    imgs.copy_(torch.rand(1, V, 3, H, W, device=dev, generator=g))      # new pixels in the static input: the replay follows them
    with torch.no_grad():
        graph.replay()
        torch.cuda.synchronize()
        again = model(imgs, proj, dv)
    assert torch.equal(out["depth"], again["depth"])
