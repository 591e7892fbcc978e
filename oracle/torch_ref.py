"""PyTorch (ATen, CPU) restatement of the reference MVSNet forward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  This is the `cpu_baseline`
that bench.py times on the GPU box's host cores (kind "port": the reference's
Python cannot travel there) and a second, ATen-level checker beside the plain-C
oracle.  It is a functional re-expression -- weights come in as a state_dict
with the reference's key names -- of:

  FeatureNet            MVSNet/models/mvsnet.py:8-45
  homo_warping          MVSNet/models/module.py:46-87
  variance aggregation  MVSNet/models/mvsnet.py:152-170
  CostRegNet            MVSNet/models/mvsnet.py:48-93
  softmax/regression    MVSNet/models/mvsnet.py:183-185, module.py:91-103
  photometric conf.     MVSNet/models/mvsnet.py:187-191
  CasMVSNet cascade     CasMVSNet/models/cas_mvsnet.py:12-66,108-164; module.py:304-438,485-524
  CVP-MVSNet            CVP-MVSNet/models/net.py:22-207; modules.py:29-78,122-275,338-355

Parity status: PINNED -- tests/test_oracle_golden.py checks every stage against
golden vectors captured from the imported reference.
"""
import torch
import torch.nn.functional as F

EPS = 1e-5


def _bn_eval(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, EPS)


def _bn_train(x, sd, p):
    return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, EPS)


def feature_net(img, sd, prefix="feature.", train=False):
    """[B,3,H,W] -> [B,32,H/4,W/4]  (mvsnet.py:8-45)"""
    bn = _bn_train if train else _bn_eval
    spec = (("conv0", 1, 1), ("conv1", 1, 1), ("conv2", 2, 2), ("conv3", 1, 1), ("conv4", 1, 1),
            ("conv5", 2, 2), ("conv6", 1, 1))
    x = img
    for name, stride, pad in spec:
        x = F.conv2d(x, sd[f"{prefix}{name}.conv.weight"], None, stride, pad)
        x = F.relu(bn(x, sd, f"{prefix}{name}.bn"))
    return F.conv2d(x, sd[prefix + "feature.weight"], sd[prefix + "feature.bias"], 1, 1)


def sweep_grid(src_proj, ref_proj, depth, H, W):
    """Normalised sampling grid [B, D, H*W, 2] (module.py:62-81).
    depth: [B,D] or [B,D,H,W]."""
    B, D = depth.shape[0], depth.shape[1]
    dev = depth.device
    M = src_proj @ torch.inverse(ref_proj)
    R, t = M[:, :3, :3], M[:, :3, 3:4]
    # the reference hard-codes float32 here (module.py:67-68); the working dtype is taken from `depth` so that the
    # same composition can be evaluated in float64 as the "true" answer of the error-budget checks
    ys, xs = torch.meshgrid(torch.arange(H, dtype=depth.dtype, device=dev),
                            torch.arange(W, dtype=depth.dtype, device=dev), indexing="ij")
    pix = torch.stack((xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, dtype=depth.dtype, device=dev)))
    ray = R @ pix.unsqueeze(0).expand(B, 3, H * W)
    pts = ray.unsqueeze(2) * depth.reshape(B, 1, D, -1) + t.reshape(B, 3, 1, 1)
    uv = pts[:, :2] / pts[:, 2:3]
    gx = uv[:, 0] / ((W - 1) / 2) - 1
    gy = uv[:, 1] / ((H - 1) / 2) - 1
    return torch.stack((gx, gy), dim=3)


def warp(src_fea, src_proj, ref_proj, depth, align_corners=False):
    """[B,C,H,W] -> [B,C,D,H,W]  (module.py:46-87)"""
    B, C, H, W = src_fea.shape
    D = depth.shape[1]
    with torch.no_grad():
        grid = sweep_grid(src_proj, ref_proj, depth, H, W)
    out = F.grid_sample(src_fea, grid.reshape(B, D * H, W, 2), mode="bilinear",
                        padding_mode="zeros", align_corners=align_corners)
    return out.reshape(B, C, D, H, W)


def variance_volume(features, projs, depth, align_corners=False, alias_quirk=False):
    """features: list of V [B,C,H,W]; projs: list of V [B,4,4] -> [B,C,D,H,W]
    (mvsnet.py:152-170; alias_quirk: CVP modules.py:228-229)."""
    V = len(features)
    D = depth.shape[1]
    ref = features[0].unsqueeze(2).expand(-1, -1, D, -1, -1)
    q = ref * ref
    s = q.clone() if alias_quirk else ref.clone()
    inplace = not torch.is_grad_enabled()     # the reference's eval branch (mvsnet.py:163-165, 170): same values
    for fea, proj in zip(features[1:], projs[1:]):
        w = warp(fea, proj, projs[0], depth, align_corners)
        if inplace:
            s += w
            q += w.pow_(2)
        else:
            s = s + w
            q = q + w * w
        del w
    if inplace:
        return q.div_(V).sub_(s.div_(V).pow_(2))
    return q / V - (s / V) ** 2


_COLUMN_BYTES = 3 << 30   # float64 only: bound on ATen's vol2col buffer per call


def conv3d_k3p1(x, w, bias, stride):
    """F.conv3d(x, w, bias, stride, padding=1) for 3x3x3 kernels.  float32 goes straight to ATen (the
    reference's call, module.py:29 / mvsnet.py:81).  float64 -- the "true answer" of the error-budget checks --
    has no oneDNN path: ATen's slow_conv3d materialises a [Cin*27, D*H*W] column buffer (157 GB for conv0 at
    1600x1184, D=192), so the output depth range is walked in slabs (same sums, plane by plane)."""
    B, Cin, D, H, W = x.shape
    if x.dtype != torch.float64 or Cin * 27 * D * H * W * 8 // (stride ** 3) <= _COLUMN_BYTES:
        return F.conv3d(x, w, bias, stride, 1)
    Do = (D - 1) // stride + 1
    per_plane = Cin * 27 * ((H - 1) // stride + 1) * ((W - 1) // stride + 1) * 8
    ch = max(1, _COLUMN_BYTES // per_plane)
    xp = F.pad(x, (0, 0, 0, 0, 1, 1))
    return torch.cat([F.conv3d(xp[:, :, z * stride:(min(Do, z + ch) - 1) * stride + 3], w, bias, stride, (0, 1, 1))
                      for z in range(0, Do, ch)], 2)


def deconv3d_k3s2(x, w):
    """F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1); float64 in slabs of input
    planes with overlap-add along depth (see conv3d_k3p1)."""
    B, Cin, D, H, W = x.shape
    Cout = w.shape[1]
    if x.dtype != torch.float64 or Cout * 27 * D * H * W * 8 <= _COLUMN_BYTES:
        return F.conv_transpose3d(x, w, None, 2, 1, 1)
    ch = max(1, _COLUMN_BYTES // (Cout * 27 * H * W * 8))
    full = x.new_zeros(B, Cout, 2 * D + 1, 2 * H, 2 * W)       # un-cropped along depth: p = 2 i + k
    for a in range(0, D, ch):
        b = min(D, a + ch)
        full[:, :, 2 * a:2 * b + 1] += F.conv_transpose3d(x[:, :, a:b], w, None, 2, (0, 1, 1), (0, 1, 1))
    return full[:, :, 1:2 * D + 1]


def cost_reg_net(x, sd, prefix="cost_regularization.", train=False, capture=None):
    """[B,32,D,H,W] -> [B,1,D,H,W]  (mvsnet.py:48-93).  capture: dict that receives the conv0 activation (error-budget
    diagnostics)."""
    bn = _bn_train if train else _bn_eval

    def conv(name, t, stride):
        t = conv3d_k3p1(t, sd[f"{prefix}{name}.conv.weight"], None, stride)
        return F.relu(bn(t, sd, f"{prefix}{name}.bn"))

    def up(name, t):
        t = deconv3d_k3s2(t, sd[f"{prefix}{name}.0.weight"])
        return F.relu(bn(t, sd, f"{prefix}{name}.1"))

    c0 = conv("conv0", x, 1)
    if capture is not None:
        capture["conv0"] = c0
    c2 = conv("conv2", conv("conv1", c0, 2), 1)
    c4 = conv("conv4", conv("conv3", c2, 2), 1)
    t = conv("conv6", conv("conv5", c4, 2), 1)
    t = c4 + up("conv7", t)
    t = c2 + up("conv9", t)
    t = c0 + up("conv11", t)
    return conv3d_k3p1(t, sd[prefix + "prob.weight"], sd[prefix + "prob.bias"], 1)


def regress(cost, depth, clamp_idx=False):
    """cost [B,D,H,W] -> (depth [B,H,W], confidence [B,H,W], prob [B,D,H,W])
    (mvsnet.py:183-191)."""
    B, D = cost.shape[:2]
    prob = F.softmax(cost, dim=1)
    dv = depth.reshape(B, D, 1, 1) if depth.dim() == 2 else depth
    est = (prob * dv).sum(1)
    with torch.no_grad():
        padded = F.pad(prob.unsqueeze(1), (0, 0, 0, 0, 1, 2))
        s4 = 4 * F.avg_pool3d(padded, (4, 1, 1), stride=1, padding=0).squeeze(1)
        ramp = torch.arange(D, dtype=cost.dtype, device=cost.device).reshape(1, D, 1, 1)
        idx = (prob * ramp).sum(1).long()
        if clamp_idx:
            idx = idx.clamp(0, D - 1)
        conf = torch.gather(s4, 1, idx.unsqueeze(1)).squeeze(1)
    return est, conf, prob


def mvsnet_forward(imgs, projs, depth, sd, train=False, stages=None):
    """imgs [B,V,3,H,W], projs [B,V,4,4], depth [B,D] -> dict like
    MVSNet.forward(refine=False) (mvsnet.py:136-195).  `stages`, if a dict,
    receives per-stage wall-clock seconds."""
    import time
    t0 = time.perf_counter()
    V = imgs.shape[1]
    feats = [feature_net(imgs[:, v], sd, train=train) for v in range(V)]
    t1 = time.perf_counter()
    var = variance_volume(feats, [projs[:, v] for v in range(V)], depth)
    t2 = time.perf_counter()
    cost = cost_reg_net(var, sd, train=train).squeeze(1)
    t3 = time.perf_counter()
    est, conf, _ = regress(cost, depth)
    t4 = time.perf_counter()
    if stages is not None:
        stages.update(feature=t1 - t0, costvol=t2 - t1, costreg=t3 - t2, regress=t4 - t3)
    return {"depth": est, "photometric_confidence": conf}


# ---- CasMVSNet (BASELINE config 3) -----------------------------------------------------

def cas_feature_net(img, sd, prefix="feature."):
    """FPN pyramid {"stage1": [B,32,H/4,W/4], "stage2": [B,16,H/2,W/2], "stage3": [B,8,H,W]}
    (CasMVSNet/models/module.py:304-405, arch "fpn")."""
    def cbr(name, t, stride, pad):
        t = F.conv2d(t, sd[f"{prefix}{name}.conv.weight"], None, stride, pad)
        return F.relu(_bn_eval(t, sd, f"{prefix}{name}.bn"))

    c0 = cbr("conv0.1", cbr("conv0.0", img, 1, 1), 1, 1)
    c1 = cbr("conv1.2", cbr("conv1.1", cbr("conv1.0", c0, 2, 2), 1, 1), 1, 1)
    top = cbr("conv2.2", cbr("conv2.1", cbr("conv2.0", c1, 2, 2), 1, 1), 1, 1)
    out = {"stage1": F.conv2d(top, sd[prefix + "out1.weight"])}
    top = F.interpolate(top, scale_factor=2, mode="nearest") + \
        F.conv2d(c1, sd[prefix + "inner1.weight"], sd[prefix + "inner1.bias"])
    out["stage2"] = F.conv2d(top, sd[prefix + "out2.weight"], None, 1, 1)
    top = F.interpolate(top, scale_factor=2, mode="nearest") + \
        F.conv2d(c0, sd[prefix + "inner2.weight"], sd[prefix + "inner2.bias"])
    out["stage3"] = F.conv2d(top, sd[prefix + "out3.weight"], None, 1, 1)
    return out


def cas_cost_reg_net(x, sd, prefix):
    """[B,Cin,D,H,W] -> [B,1,D,H,W]  (CasMVSNet/models/module.py:407-438; prob has no bias)."""
    def layer(name, t, stride, up=False):
        w = sd[f"{prefix}{name}.conv.weight"]
        t = deconv3d_k3s2(t, w) if up else conv3d_k3p1(t, w, None, stride)
        return F.relu(_bn_eval(t, sd, f"{prefix}{name}.bn"))

    c0 = layer("conv0", x, 1)
    c2 = layer("conv2", layer("conv1", c0, 2), 1)
    c4 = layer("conv4", layer("conv3", c2, 2), 1)
    t = layer("conv6", layer("conv5", c4, 2), 1)
    t = c4 + layer("conv7", t, 2, True)
    t = c2 + layer("conv9", t, 2, True)
    t = c0 + layer("conv11", t, 2, True)
    return conv3d_k3p1(t, sd[prefix + "prob.weight"], None, 1)


def cas_hypotheses(cur, ndepth, interval, H, W):
    """module.py:485-524: [B,D0] sweep end points or [B,H,W] previous depth -> [B,ndepth,H,W]."""
    ramp = torch.arange(ndepth, dtype=cur.dtype, device=cur.device)
    if cur.dim() == 2:
        lo, hi = cur[:, 0], cur[:, -1]
        step = (hi - lo) / (ndepth - 1)
        d = lo[:, None] + ramp[None] * step[:, None]
        return d[:, :, None, None].repeat(1, 1, H, W)
    lo, hi = cur - ndepth / 2 * interval, cur + ndepth / 2 * interval
    step = (hi - lo) / (ndepth - 1)
    return lo[:, None] + ramp.reshape(1, -1, 1, 1) * step[:, None]


def cascade_forward(imgs, projs, depth_values, sd, ndepths=(48, 32, 8), ratios=(4, 2, 1), stages=None):
    """imgs [B,V,3,H,W]; projs {"stageK": [B,V,2,4,4]}; depth_values [B,D] -> dict like
    CascadeMVSNet.forward (cas_mvsnet.py:108-164)."""
    import time
    B, V, _, H, W = imgs.shape
    interval = (float(depth_values[0, -1]) - float(depth_values[0, 0])) / depth_values.size(1)
    t0 = time.perf_counter()
    feats = [cas_feature_net(imgs[:, v], sd) for v in range(V)]
    if stages is not None:
        stages["feature"] = time.perf_counter() - t0
    out, depth = {}, None
    for s, scale in enumerate((4, 2, 1)):
        t0 = time.perf_counter()
        key = f"stage{s + 1}"
        cur = depth_values if depth is None else F.interpolate(
            depth.unsqueeze(1), [H, W], mode="bilinear", align_corners=False).squeeze(1)
        hyp = cas_hypotheses(cur, ndepths[s], ratios[s] * interval, H, W)
        hyp = F.interpolate(hyp.unsqueeze(1), [ndepths[s], H // scale, W // scale], mode="trilinear",
                            align_corners=False).squeeze(1)
        P = projs[key]
        full = P[:, :, 0].clone()
        full[:, :, :3, :4] = torch.matmul(P[:, :, 1, :3, :3], P[:, :, 0, :3, :4])
        var = variance_volume([f[key] for f in feats], [full[:, v] for v in range(V)], hyp)
        cost = cas_cost_reg_net(var, sd, f"cost_regularization.{s}.").squeeze(1)
        depth, conf, _ = regress(cost, hyp, clamp_idx=True)
        out[key] = {"depth": depth, "photometric_confidence": conf}
        out.update(out[key])
        if stages is not None:
            stages[key] = time.perf_counter() - t0
    return out


# ---- CVP-MVSNet (BASELINE configs[3]) --------------------------------------------------

def cvp_feature_pyramid(img, sd, nscale, prefix="featurePyramid."):
    """[B,3,H,W] -> nscale maps [B,16,H/2^l,W/2^l] (net.py:22-51: 9 x (3x3 conv + LeakyReLU 0.1)
    on a bilinear x0.5 image pyramid)."""
    names = ("conv0aa", "conv0ba", "conv0bb", "conv0bc", "conv0bd", "conv0be", "conv0bf", "conv0bg", "conv0bh")

    def cnn(x):
        for n in names:
            x = F.leaky_relu(F.conv2d(x, sd[f"{prefix}{n}.0.weight"], sd[f"{prefix}{n}.0.bias"], 1, 1), 0.1)
        return x

    out = [cnn(img)]
    for _ in range(nscale - 1):
        img = F.interpolate(img, scale_factor=0.5, mode="bilinear", align_corners=None)
        out.append(cnn(img))
    return out


def cvp_cost_reg_net(x, sd, prefix="cost_reg_refine."):
    """[B,16,D,H,W] -> [B,D,H,W]  (net.py:53-97)."""
    def cbr(name, t, stride=1):
        t = conv3d_k3p1(t, sd[f"{prefix}{name}.conv.weight"], None, stride)
        return F.relu(_bn_eval(t, sd, f"{prefix}{name}.bn"))

    def up(name, t, stride, outpad):
        w = sd[f"{prefix}{name}.0.weight"]
        if t.dtype == torch.float64 and stride == 1:      # = a convolution with the flipped, transposed kernel (slab-wise)
            t = conv3d_k3p1(t, w.flip(2, 3, 4).transpose(0, 1).contiguous(), None, 1)
        elif t.dtype == torch.float64:
            t = deconv3d_k3s2(t, w)
        else:
            t = F.conv_transpose3d(t, w, None, stride, 1, outpad)
        return F.relu(_bn_eval(t, sd, f"{prefix}{name}.1"))

    c0 = cbr("conv0a", cbr("conv0", x))
    c2 = cbr("conv2a", cbr("conv2", cbr("conv1", c0, 2)))
    c4 = cbr("conv4a", cbr("conv4", cbr("conv3", c2)))
    c5 = c2 + up("conv5", c4, 1, 0)
    c6 = c0 + up("conv6", c5, 2, 1)
    return conv3d_k3p1(c6, sd[prefix + "prob0.weight"], sd[prefix + "prob0.bias"], 1).squeeze(1)


def _cvp_proj(K, E):
    P = torch.zeros_like(E)
    P[..., :3, :] = K @ E[..., :3, :]
    P[..., 3, 3] = 1.0
    return P


def cvp_refine_hypotheses(depth_up, Kr, Ks, Er, Es, d=4):
    """modules.py:147-219 (test branch): depth_up [B,H,W] -> [B,2d,H,W], float64 geometry."""
    B, H, W = depth_up.shape
    out = depth_up.unsqueeze(1).repeat(1, 2 * d, 1, 1)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    pix = torch.stack((xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, dtype=torch.float64)))
    for b in range(B):
        kr, ks, er, es = Kr[b].double(), Ks[b].double(), Er[b].double(), Es[b].double()
        d1 = depth_up[b].reshape(-1).double()

        def src_pixel(depth):
            cam = torch.inverse(kr) @ (pix * depth)
            world = torch.inverse(er) @ torch.cat((cam, torch.ones_like(cam[:1])), 0)
            p = ks @ (es @ world)[:3]
            return p / p[2], p[2].clone()

        x1, z1 = src_pixel(d1)
        x2, _ = src_pixel(d1 + 1)
        th = torch.atan((x2[1] - x1[1]) / (x2[0] - x1[0]))
        x3 = x1 + torch.stack((torch.cos(th), torch.sin(th), torch.zeros_like(th)))
        A = (kr @ er[:3, :3]) @ torch.inverse(ks @ es[:3, :3])
        t1, t2 = z1 * (A @ x1), A @ x3
        M = torch.stack((pix.t()[:, 1:], t2.t()[:, 1:]), 2)
        step = (torch.inverse(M) @ t1.t()[:, 1:].unsqueeze(2))[:, 0, 0].abs().mean().to(depth_up.dtype)
        for k in range(-d, d):
            out[b, k + d] += k * step
    return out.to(depth_up.dtype)


def cvp_forward(ref_img, src_imgs, ref_in, src_in, ref_ex, src_ex, depth_min, depth_max, sd, nscale):
    """CVP-MVSNet `network.forward` in test mode (net.py:106-207) -> {"depth_est_list" (finest
    first), "prob_confidence"}."""
    import warnings
    nsrc = src_imgs.shape[1]
    pyr = [cvp_feature_pyramid(ref_img, sd, nscale)] + [cvp_feature_pyramid(src_imgs[:, i], sd, nscale) for i in range(nsrc)]
    H0 = ref_img.shape[2]

    def K_at(K, level):
        K = K.clone()
        K[:, :2, :] = K[:, :2, :] / (H0 / pyr[0][level].shape[2])
        return K

    def level_cost(level, hyp):
        feats = [p[level] for p in pyr]
        projs = [_cvp_proj(K_at(ref_in, level), ref_ex)] + \
                [_cvp_proj(K_at(src_in[:, i], level), src_ex[:, i]) for i in range(nsrc)]
        return cvp_cost_reg_net(variance_volume(feats, projs, hyp, alias_quirk=True), sd)

    step = (depth_max[0] - depth_min[0]) / 47
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hyp = torch.range(float(depth_min[0]), float(depth_max[0]), float(step),
                          dtype=ref_img.dtype).unsqueeze(0).repeat(ref_img.shape[0], 1)
    depth, conf, _ = regress(level_cost(nscale - 1, hyp), hyp)
    depths = [depth]
    for level in range(nscale - 2, -1, -1):
        up = F.interpolate(depth[None], scale_factor=2, mode="bicubic", align_corners=None).squeeze(0)
        hyp = cvp_refine_hypotheses(up, K_at(ref_in, level), K_at(src_in[:, 0], level), ref_ex, src_ex[:, 0])
        depth, conf, _ = regress(level_cost(level, hyp), hyp)
        depths.append(depth)
    depths.reverse()
    return {"depth_est_list": depths, "prob_confidence": conf}


def masked_smooth_l1(est, gt, mask):
    """mvsnet.py:201-203"""
    m = mask > 0.5
    return F.smooth_l1_loss(est[m], gt[m], reduction="mean")


from mvs_amd.synth import random_state_dict  # noqa: E402,F401  (seeded synthetic weights)
