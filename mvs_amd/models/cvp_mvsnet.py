"""CVP-MVSNet (BASELINE configs[3]) with its cost-volume path on the HIP kernels.

Mirrors the reference's surface (CVP-MVSNet/models/net.py:99-207):

    net = network(args)          # args.nscale, args.nsrc, args.mode ("test" here)
    out = net(ref_img [B,3,H,W], src_imgs [B,nsrc,3,H,W], ref_in [B,3,3], src_in [B,nsrc,3,3],
              ref_ex [B,4,4], src_ex [B,nsrc,4,4], depth_min [B], depth_max [B])
    out["depth_est_list"]   # finest first
    out["prob_confidence"]

Module and parameter names equal the reference's.  What runs where: the 9-layer LeakyReLU
feature pyramid (net.py:22-51) and the glue between levels -- intrinsics per level
(modules.py:29-50), the 48 sweep planes (:57-78), bicubic x2 upsampling (net.py:170), the
fp64 "one source pixel along the epipolar line" hypothesis interval (modules.py:122-219) -- are
PyTorch-ROCm ops; per level the variance volume over all views (net.py:130-149,
modules.py:221-275: S0 = Q0 = ref^2, the in-place pow_ aliasing, kept), the 3D U-Net
(net.py:53-97) and the softmax regression run on the kernels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _conv_lrelu(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=True), nn.LeakyReLU(0.1))


class FeaturePyramid(nn.Module):
    """Shared 9-layer CNN applied to an image pyramid (bilinear x0.5 per level):
    -> [[B,16,H,W], [B,16,H/2,W/2], ...]  (net.py:22-51)."""
    _ORDER = ("conv0aa", "conv0ba", "conv0bb", "conv0bc", "conv0bd", "conv0be", "conv0bf", "conv0bg", "conv0bh")
    _CH = (3, 64, 64, 64, 32, 32, 32, 16, 16, 16)

    def __init__(self):
        super().__init__()
        for name, cin, cout in zip(self._ORDER, self._CH[:-1], self._CH[1:]):
            setattr(self, name, _conv_lrelu(cin, cout))

    def _cnn(self, x):
        for name in self._ORDER:
            x = getattr(self, name)(x)
        return x

    def forward(self, img, scales=5):
        out = [self._cnn(img)]
        for _ in range(scales - 1):
            img = F.interpolate(img, scale_factor=0.5, mode="bilinear", align_corners=None).detach()
            out.append(self._cnn(img))
        return out

    # -- inference path on the HIP 2D kernels: conv bias in the epilogue, LeakyReLU(0.1)
    def _hip_params(self):
        key = tuple((p._version, p.data_ptr()) for p in self.parameters())
        cache = getattr(self, "_hip_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1]
        P = []
        with torch.no_grad():
            for name in self._ORDER:
                conv = getattr(self, name)[0]
                w = conv.weight.detach().float().contiguous()
                P.append(dict(cin=w.shape[1], cout=w.shape[0], packed=ops.pack_conv2d_weight(w, 1, split=True),
                              shift=conv.bias.detach().float().contiguous()))
        self._hip_cache = (key, P)
        return P

    def hip_supported(self):
        return all(p["packed"] is not None for p in self._hip_params())

    def forward_hip(self, img, scales=5):
        """[N,3,H,W] -> `scales` channels-last maps [N,H/2^l,W/2^l,16]; the image pyramid's
        bilinear x0.5 is mvs_downsample_bilinear_half_f32 (bit-identical to ATen's CPU kernel)."""
        P = self._hip_params()

        def cnn(x):
            # a layer on a two-piece fp16 kernel scales its input by the absmax block the layer in front of it left (ops.conv2d)
            blocks = ops.absmax_block(x.device, zero=True, n=len(P)) if ops.split_f16_enabled() else None
            for i, p in enumerate(P):
                x = ops.conv2d(x, p["packed"], p["cin"], p["cout"], 3, 1, None, p["shift"], 2, planar=(i == 0),
                               x_absmax=blocks[i - 1] if (blocks is not None and i > 0) else None,
                               out_absmax=blocks[i] if (blocks is not None and i + 1 < len(P)) else None)
            return x

        out = [cnn(img)]
        for _ in range(scales - 1):
            img = ops.downsample_bilinear_half(img)
            out.append(cnn(img))
        return out


class _CBR3d(nn.Module):
    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn = nn.BatchNorm3d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class CostRegNet(nn.Module):
    """net.py:53-97: one stride-2 level; conv5 is a STRIDE-1 transposed convolution."""

    def __init__(self):
        super().__init__()
        self.conv0, self.conv0a = _CBR3d(16, 16), _CBR3d(16, 16)
        self.conv1 = _CBR3d(16, 32, 2)
        self.conv2, self.conv2a = _CBR3d(32, 32), _CBR3d(32, 32)
        self.conv3 = _CBR3d(32, 64)
        self.conv4, self.conv4a = _CBR3d(64, 64), _CBR3d(64, 64)
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(64, 32, 3, padding=1, output_padding=0, stride=1, bias=False),
                                   nn.BatchNorm3d(32), nn.ReLU(inplace=True))
        self.conv6 = nn.Sequential(nn.ConvTranspose3d(32, 16, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(16), nn.ReLU(inplace=True))
        self.prob0 = nn.Conv3d(16, 1, 3, stride=1, padding=1)

    def forward(self, x):   # planar torch form
        c0 = self.conv0a(self.conv0(x))
        c2 = self.conv2a(self.conv2(self.conv1(c0)))
        c4 = self.conv4a(self.conv4(self.conv3(c2)))
        c5 = c2 + self.conv5(c4)
        c6 = c0 + self.conv6(c5)
        return self.prob0(c6).squeeze(1)

    def _hip_params(self):
        key = tuple((p._version, p.data_ptr()) for p in list(self.parameters()) + list(self.buffers()))
        cache = getattr(self, "_hip_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1]

        def fold(bn):
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
            return scale, (bn.bias - bn.running_mean * scale).float().contiguous()

        def layer(w, bn, stride, transposed):
            w = w.detach().float().contiguous()
            scale, shift = fold(bn)
            return dict(weight=w, scale=scale, shift=shift, stride=stride, transposed=transposed,
                        packed=ops.pack_conv3d_weight(w, transposed, stride, split=True))

        P = {}
        with torch.no_grad():
            for name, stride in (("conv0", 1), ("conv0a", 1), ("conv1", 2), ("conv2", 1), ("conv2a", 1),
                                 ("conv3", 1), ("conv4", 1), ("conv4a", 1)):
                m = getattr(self, name)
                P[name] = layer(m.conv.weight, m.bn, stride, False)
            # stride-1 transposed conv == convolution with the kernel flipped and its channel
            # axes swapped: w'[co][ci][k] = w[ci][co][2-k]
            w5 = self.conv5[0].weight.detach().flip(2, 3, 4).transpose(0, 1)
            P["conv5"] = layer(w5, self.conv5[1], 1, False)
            P["conv6"] = layer(self.conv6[0].weight, self.conv6[1], 2, True)
            w = self.prob0.weight.detach().float().contiguous()
            P["prob0"] = dict(weight=w, scale=None, shift=self.prob0.bias.detach().float().contiguous(), stride=1,
                              transposed=False, packed=ops.pack_conv3d_weight(w, False, 1))
        self._hip_cache = (key, P)
        return P

    def forward_hip(self, x_cl, x_absmax=None):
        """x_cl [B,D,H,W,16] channels-last -> cost [B,D,H,W].  x_absmax: the absmax block of x_cl from the variance op -- the
        layers then run on the two-piece fp16 kernels, each scaling its input by the block the layer in front of it left."""
        P = self._hip_params()
        f16 = x_absmax is not None and ops.split_f16_enabled()
        blocks = ops.absmax_block(x_cl.device, zero=True, n=9) if f16 else None
        blk = (lambda i: blocks[i]) if f16 else (lambda i: None)

        def run(name, t, skip=None, relu=True, xa=None, oa=None):
            p = P[name]
            return ops.conv3d(t, p["weight"], p["scale"], p["shift"], skip, relu, p["transposed"], p["stride"],
                              channels_last=True, packed=p["packed"], x_absmax=xa, out_absmax=oa)

        c0 = run("conv0a", run("conv0", x_cl, xa=x_absmax if f16 else None, oa=blk(0)), xa=blk(0), oa=blk(1))
        c2 = run("conv2a", run("conv2", run("conv1", c0, xa=blk(1), oa=blk(2)), xa=blk(2), oa=blk(3)), xa=blk(3), oa=blk(4))
        c4 = run("conv4a", run("conv4", run("conv3", c2, xa=blk(4), oa=blk(5)), xa=blk(5), oa=blk(6)), xa=blk(6), oa=blk(7))
        c5 = run("conv5", c4, c2, xa=blk(7), oa=blk(8))
        c6 = run("conv6", c5, c0, xa=blk(8))
        return run("prob0", c6, None, relu=False).squeeze(-1)


def condition_intrinsics(K, img_shape, fp_shapes):
    """[B,3,3] -> [B,nscale,3,3]: rows 0-1 divided by the level's downsampling ratio."""
    out = []
    for s in fp_shapes:
        Kl = K.clone()
        Kl[:, :2, :] = Kl[:, :2, :] / (img_shape[2] / s[2])
        out.append(Kl)
    return torch.stack(out, 1)


def sweep_planes(depth_min, depth_max, n=48):
    """[B,n] uniform planes from depth_min[0] to depth_max[0] (modules.py:57-78: every batch
    item gets item 0's range, as the reference's loop does)."""
    import warnings
    step = (depth_max[0] - depth_min[0]) / (n - 1)
    with warnings.catch_warnings():   # torch.range (inclusive end) is what the reference calls
        warnings.simplefilter("ignore")
        row = torch.range(float(depth_min[0]), float(depth_max[0]), float(step), device=depth_min.device)
    return row.unsqueeze(0).repeat(depth_min.shape[0], 1)


def _full_proj(K, E):
    """[...,3,3], [...,4,4] -> [...,4,4] = [K @ E[:3]; 0 0 0 1]."""
    P = torch.zeros_like(E)
    P[..., :3, :] = torch.matmul(K, E[..., :3, :])
    P[..., 3, 3] = 1.0
    return P


def refine_hypotheses_hip(depth_up, K_ref, K_src0, E_ref, E_src0, d=4, pixel_interval=1.0):
    """refine_hypotheses with the per-pixel fp64 arithmetic and the mean in one kernel
    (mvs_cvp_interval_sum_f64); the 3x3 / 4x4 matrix algebra stays fp64 torch ops on the device."""
    import ctypes
    from .._lib import check, load, stream
    B, H, W = depth_up.shape
    dev = depth_up.device
    depth_up = depth_up.float().contiguous()
    ks = torch.arange(-d, d, device=dev, dtype=torch.float32).view(1, -1, 1, 1)
    intervals = []
    for b in range(B):
        Kr, Ks, Er, Es = (t[b].double() for t in (K_ref, K_src0, E_ref, E_src0))
        A = torch.matmul(torch.matmul(Kr, Er[:3, :3]), torch.inverse(torch.matmul(Ks, Es[:3, :3])))
        mats = torch.cat((torch.inverse(Kr).reshape(-1), torch.inverse(Er).reshape(-1), Ks.reshape(-1),
                          Es.reshape(-1), A.reshape(-1))).contiguous()
        total = torch.empty((1,), device=dev, dtype=torch.float64)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        check(load().mvs_cvp_interval_sum_f64(vp(depth_up[b]), vp(mats), H, W, float(pixel_interval), vp(total),
                                              stream()), "mvs_cvp_interval_sum_f64")
        intervals.append((total / (H * W)).float())
    interval = torch.stack(intervals).view(B, 1, 1, 1)
    return depth_up.unsqueeze(1) + ks * interval


def refine_hypotheses(depth_up, K_ref, K_src0, E_ref, E_src0, d=4, pixel_interval=1.0):
    """[B,H,W] upsampled depth -> [B,2d,H,W] hypotheses depth_up + k * interval, k = -d..d-1,
    where `interval` is the mean over the pixels of the depth step that moves the projection
    into the FIRST source view by one pixel along the epipolar line (modules.py:147-219), in
    float64 like the reference."""
    B, H, W = depth_up.shape
    dev = depth_up.device
    out = depth_up.unsqueeze(1).repeat(1, 2 * d, 1, 1)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float64),
                            torch.arange(W, device=dev, dtype=torch.float64), indexing="ij")
    pix = torch.stack((xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, device=dev, dtype=torch.float64)))  # [3,N]
    for b in range(B):
        Kr, Ks, Er, Es = (t[b].double() for t in (K_ref, K_src0, E_ref, E_src0))
        d1 = depth_up[b].reshape(-1).double()

        def to_src(depth):   # reference pixel at `depth` -> homogeneous source pixel, its z
            cam = torch.matmul(torch.inverse(Kr), pix * depth)
            world = torch.matmul(torch.inverse(Er), torch.cat((cam, torch.ones_like(cam[:1])), 0))
            p = torch.matmul(Ks, torch.matmul(Es, world)[:3])
            z = p[2].clone()
            return p / z, z

        x1, z1 = to_src(d1)
        x2, _ = to_src(d1 + 1)
        theta = torch.atan((x2[1] - x1[1]) / (x2[0] - x1[0]))
        x3 = x1 + torch.stack((torch.cos(theta) * pixel_interval, torch.sin(theta) * pixel_interval,
                               torch.zeros_like(theta)))
        A = torch.matmul(torch.matmul(Kr, Er[:3, :3]), torch.inverse(torch.matmul(Ks, Es[:3, :3])))
        t1 = z1 * torch.matmul(A, x1)
        t2 = torch.matmul(A, x3)
        # first unknown of the 2x2 system  [y t2_y; 1 t2_z] [a; b] = [t1_y; t1_z]  per pixel
        # (Cramer's rule; the reference inverts the N 2x2 matrices with a batched LU)
        det = pix[1] * t2[2] - t2[1] * pix[2]
        ans0 = (t1[1] * t2[2] - t2[1] * t1[2]) / det
        interval = ans0.abs().mean().float()
        for k in range(-d, d):
            out[b, k + d] += k * interval
    return out.float()


class CVPMVSNet(nn.Module):
    def __init__(self, args=None, nscale=2, nsrc=2, proj_where="host", feature_impl="hip"):
        super().__init__()
        self.feature_impl = feature_impl   # "hip": 2D MFMA kernels; "torch": PyTorch-ROCm / MIOpen
        self.nscale = args.nscale if args is not None else nscale
        self.nsrc = args.nsrc if args is not None else nsrc
        self.args = args
        self.proj_where = proj_where
        self.featurePyramid = FeaturePyramid()
        self.cost_reg_refine = CostRegNet()

    def _level(self, feats_cl, K_ref, K_src, E_ref, E_src, hypos):
        """feats_cl: V tensors [B,H,W,16]; hypos [B,D] or [B,D,H,W] -> (cost [B,D,H,W])."""
        B = K_ref.shape[0]
        dev = K_ref.device
        where = self.proj_where
        Ks = torch.cat((K_ref.unsqueeze(1), K_src), 1)
        Es = torch.cat((E_ref.unsqueeze(1), E_src), 1)
        proj = _full_proj(Ks.cpu() if where == "host" else Ks, Es.cpu() if where == "host" else Es)
        rts = ops.rot_trans_all(proj, where, device=dev)
        f = torch.stack(feats_cl)                                   # [V,B,H,W,16] = [V,B,1,H,W,16] blocked
        f16 = f.reshape(f.shape[0], B, 1, f.shape[2], f.shape[3], 16).contiguous()
        amax = ops.absmax_block(dev) if ops.split_f16_enabled() else None                # the regulariser's first operand scale
        var = ops.costvol_variance_c16(f16[0], f16[1:], rts, hypos, alias_quirk=True, absmax_out=amax)   # [B,D,H,W,16]
        return self.cost_reg_refine.forward_hip(var, amax)

    def forward(self, ref_img, src_imgs, ref_in, src_in, ref_ex, src_ex, depth_min, depth_max):
        if self.training:
            raise NotImplementedError("CVPMVSNet here is the inference path (BASELINE configs[3])")
        nsrc, nscale = self.nsrc, self.nscale
        use_hip = self.feature_impl == "hip" and self.featurePyramid.hip_supported()
        if use_hip:   # all views as one batch, channels-last maps
            B = ref_img.shape[0]
            allv = torch.cat((ref_img.unsqueeze(1), src_imgs), 1).reshape(B * (nsrc + 1), *ref_img.shape[1:])
            maps = self.featurePyramid.forward_hip(allv, nscale)
            pyr = [[m.reshape(B, nsrc + 1, *m.shape[1:])[:, v] for m in maps] for v in range(nsrc + 1)]
            shapes = [(B, 16, m.shape[1], m.shape[2]) for m in maps]
        else:
            pyr = [self.featurePyramid(ref_img, nscale)] + \
                  [self.featurePyramid(src_imgs[:, i], nscale) for i in range(nsrc)]
            shapes = [f.shape for f in pyr[0]]
        K_ref = condition_intrinsics(ref_in, ref_img.shape, shapes)                             # [B,nscale,3,3]
        K_src = torch.stack([condition_intrinsics(src_in[:, i], ref_img.shape, shapes)
                             for i in range(nsrc)], 1)                                         # [B,nsrc,nscale,3,3]

        def level_feats(level):
            if use_hip:
                return [p[level].contiguous() for p in pyr]
            return [ops.nchw_to_nhwc(p[level]) for p in pyr]

        depths = []
        hypos = sweep_planes(depth_min, depth_max).to(ref_img.device)
        cost = self._level(level_feats(nscale - 1), K_ref[:, -1], K_src[:, :, -1], ref_ex, src_ex, hypos)
        depth, conf, _ = ops.softmax_regress_conf(cost, hypos)
        depths.append(depth)
        for level in range(nscale - 2, -1, -1):
            if use_hip:   # bicubic x2, fp64 camera algebra, epipolar step, hypotheses: all HIP kernels
                up = ops.upsample_bicubic2x(depth)
                hypos = ops.cvp_refine_hypotheses(up, K_ref[:, level], K_src[:, 0, level], ref_ex, src_ex[:, 0])
            else:
                up = F.interpolate(depth[None], scale_factor=2, mode="bicubic", align_corners=None).squeeze(0)
                hypos = refine_hypotheses(up, K_ref[:, level], K_src[:, 0, level], ref_ex, src_ex[:, 0]).contiguous()
            cost = self._level(level_feats(level), K_ref[:, level], K_src[:, :, level], ref_ex, src_ex, hypos)
            depth, conf, _ = ops.softmax_regress_conf(cost, hypos)
            depths.append(depth)
        depths.reverse()
        return {"depth_est_list": depths, "prob_confidence": conf}


network = CVPMVSNet   # the reference's class name (net.py:99)
