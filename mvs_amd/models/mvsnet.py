"""MVSNet with the reference's Python surface (MVSNet/models/mvsnet.py) and the
cost-volume path on MI355X HIP kernels.

    model = MVSNet(refine=False)
    out = model(imgs[B,V,3,H,W], proj_matrices[B,V,4,4], depth_values[B,D])
    out["depth"], out["photometric_confidence"]            # [B,H/4,W/4]

Module / parameter names equal the reference's, so `load_state_dict` accepts
its checkpoints (including `module.`-prefixed DataParallel ones via
`load_reference_checkpoint`).

eval():  FeatureNet on the HIP 2D kernels (fused head + split-operand MFMA layers, BatchNorm folded)
         -> fused warp+variance (persistent LDS-DMA sweep kernel, per-view volumes never
         materialised) -> CostRegNet as split-operand MFMA 3D convolutions with folded BatchNorm
         (one C call, mvs_costreg_fwd2_f32) -> fused softmax / expectation / confidence.
train(): every layer of FeatureNet and CostRegNet runs on the HIP kernels in both directions under
         autograd (mvs_amd/train_ops.py: forward, input gradient, weight gradient; batch-statistics
         BatchNorm + ReLU as one fused HIP op), warp+variance and softmax-regression with their
         hand-written backward kernels; no MIOpen kernel is in a step.  PyTorch-ROCm layers remain as
         A/B switches (feature_impl / train_feature_impl / train_impl = "torch").
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .module import ConvBnReLU, ConvBnReLU3D


class FeatureNet(nn.Module):
    """[B,3,H,W] -> [B,32,H/4,W/4]  (mvsnet.py:8-45)"""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        self.conv0 = ConvBnReLU(3, 8, 3, 1, 1)
        self.conv1 = ConvBnReLU(8, 8, 3, 1, 1)
        self.conv2 = ConvBnReLU(8, 16, 5, 2, 2)
        self.conv3 = ConvBnReLU(16, 16, 3, 1, 1)
        self.conv4 = ConvBnReLU(16, 16, 3, 1, 1)
        self.conv5 = ConvBnReLU(16, 32, 5, 2, 2)
        self.conv6 = ConvBnReLU(32, 32, 3, 1, 1)
        self.feature = nn.Conv2d(32, 32, 3, 1, 1)

    def forward(self, x):
        for blk in (self.conv0, self.conv1, self.conv2, self.conv3, self.conv4, self.conv5,
                    self.conv6):
            x = blk(x)
        return self.feature(x)

    # -- inference path on the HIP 2D MFMA kernels (BN folded), channels-last output
    _PLAN = (("conv0", 1), ("conv1", 1), ("conv2", 2), ("conv3", 1), ("conv4", 1), ("conv5", 2),
             ("conv6", 1))

    def _hip_params(self):
        mods = [(n, getattr(self, n), s) for n, s in self._PLAN]
        key = tuple((p._version, p.data_ptr()) for _, m, _ in mods
                    for p in (m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean,
                              m.bn.running_var))
        key += ((self.feature.weight._version, self.feature.weight.data_ptr()),
                (self.feature.bias._version, self.feature.bias.data_ptr()))
        cache = getattr(self, "_hip_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1]
        P = []
        with torch.no_grad():
            for name, m, stride in mods:
                w = m.conv.weight.detach().float().contiguous()
                scale = (m.bn.weight / torch.sqrt(m.bn.running_var + m.bn.eps)).float().contiguous()
                shift = (m.bn.bias - m.bn.running_mean * scale).float().contiguous()
                P.append(dict(name=name, cin=w.shape[1], cout=w.shape[0], k=w.shape[2],
                              stride=stride, packed=ops.pack_conv2d_weight(w, stride, split=True), scale=scale,
                              shift=shift, relu=True, weight=w,
                              head=ops.pack_feature_head_weight(w) if name == "conv1" else None))
            w = self.feature.weight.detach().float().contiguous()
            P.append(dict(name="feature", cin=w.shape[1], cout=w.shape[0], k=w.shape[2], stride=1,
                          packed=ops.pack_conv2d_weight(w, 1, split=True), scale=None,
                          shift=self.feature.bias.detach().float().contiguous(), relu=False))
        self._hip_cache = (key, P)
        return P

    def hip_supported(self):
        return all(p["packed"] is not None for p in self._hip_params())

    def forward_train_hip(self, img_nchw, groups=1):
        """Autograd path on the HIP 2D kernels, forward and both gradients (mvs_amd/train_ops.py: no MIOpen):
        [N,3,H,W] -> [N,H/4,W/4,32] channels-last, batch-statistics BatchNorm (fused HIP op) as in forward().
        groups: the N images are that many consecutive blocks (the V views of a sample batch, view-major), each normalised
        with its own batch statistics and counted as one update of the running statistics -- the reference's V calls
        `self.feature(imgs[:, v])` (mvsnet.py:146) as ONE launch per layer (round 4: a third of the launches of a step)."""
        from ..train_ops import conv2d_bn_relu_cl, conv2d_cl
        x = conv2d_bn_relu_cl(img_nchw, self.conv0.conv, self.conv0.bn, 1, planar=True, groups=groups)
        for name, stride in self._PLAN[1:]:
            m = getattr(self, name)
            x = conv2d_bn_relu_cl(x, m.conv, m.bn, stride, groups=groups)
        return conv2d_cl(x, self.feature.weight, 1) + self.feature.bias      # [N,H/4,W/4,32] channels-last

    def forward_train_cl(self, img_nchw):
        """Autograd path in channels-last: torch's conv2d (MIOpen NHWC kernels, forward and both
        gradients) + the fused HIP BatchNorm/ReLU op (batch statistics, as in forward()).
        [N,3,H,W] -> [N,H/4,W/4,32] channels-last."""
        x = ops.nchw_to_nhwc(img_nchw).permute(0, 3, 1, 2)        # NCHW view of NHWC storage
        for name, stride in self._PLAN:
            m = getattr(self, name)
            y = F.conv2d(x, m.conv.weight, None, stride, m.conv.padding)
            y = ops.bn_relu_cl(y.permute(0, 2, 3, 1).contiguous(), m.bn)   # no copy when y is NHWC
            x = y.permute(0, 3, 1, 2)
        y = F.conv2d(x, self.feature.weight, self.feature.bias, 1, 1)
        return y.permute(0, 2, 3, 1).contiguous()

    def forward_hip(self, imgs_nchw, out_c4=False, collect_absmax=False):
        """[N,3,H,W] image batch (the reference's layout) -> [N,H/4,W/4,32] channels-last, or with
        out_c4 the last layer's epilogue writes 4-channel blocks [N,8,H/4,W/4,4] (MVS_LAYOUT_C4).  collect_absmax: returns
        (maps, block) -- the last layer's epilogue collects the maps' largest magnitude in an absmax block (the bound a hand-over
        sweep scales its pieces by; a row of the layers' own blocks: one fill zeroes them all)."""
        x = imgs_nchw
        P = self._hip_params()
        first = 0
        # a layer on a two-piece fp16 kernel scales its input by the absmax block the layer in front of it collected in its
        # epilogue (ops.conv2d: x_absmax / out_absmax); the chain starts at the fused head
        head = ops.feature_head_enabled() and ops.feature_head_supported(x.shape[2], x.shape[3])
        # (+ two rows: the flag words and the spare block of the fused conv3 + conv4 kernel -- zeroed by the same fill)
        blocks = ops.absmax_block(x.device, zero=True, n=len(P) + 3) if (head and ops.split_f16_enabled()) else None
        out_absmax = None
        if collect_absmax:      # (row len(P) + 2; without the layers' blocks: one of its own)
            out_absmax = blocks[len(P) + 2] if blocks is not None else ops.absmax_block(x.device, zero=True)
        if head:
            with ops.stage("feature.head"):   # conv0 + conv1 in one kernel
                x = ops.feature_head(x, P[0]["weight"], P[0]["scale"], P[0]["shift"], P[1]["head"], P[1]["scale"],
                                     P[1]["shift"], out_absmax=blocks[1] if blocks is not None else None)
            first = 2
        skip = -1
        for i, p in enumerate(P):
            if i < first or i == skip:
                continue
            last = i == len(P) - 1
            # two consecutive 3x3 stride-1 16 -> 16 layers (conv3 + conv4): one kernel, the map between them stays in LDS
            nx = P[i + 1] if i + 1 < len(P) else None
            if (blocks is not None and nx is not None and ops.conv2d_pair_enabled() and p["k"] == 3 and nx["k"] == 3 and
                    p["stride"] == 1 and nx["stride"] == 1 and p["cin"] == p["cout"] == nx["cin"] == nx["cout"] == 16 and
                    p["relu"] is True and ops.f16_companion(p["packed"]) is not None and ops.f16_companion(nx["packed"]) is not None):
                if p.get("pair") is None:
                    p["pair"] = ops.pack_conv2d_pair(p["weight"], nx["weight"])
                if p["pair"] is not None:
                    nlast = i + 1 == len(P) - 1
                    with ops.stage("feature." + p["name"] + "+" + nx["name"]):
                        x = ops.conv2d_pair(x, blocks[i - 1], p["pair"], p, nx, out_c4=(out_c4 and nlast),
                                            out_absmax=blocks[i + 1] if not nlast else out_absmax, flag=blocks[len(P):len(P) + 2].reshape(-1))
                    skip = i + 1
                    continue
            with ops.stage("feature." + p["name"]):
                x = ops.conv2d(x, p["packed"], p["cin"], p["cout"], p["k"], p["stride"], p["scale"],
                               p["shift"], p["relu"], planar=(i == 0), out_c4=(out_c4 and last),
                               x_absmax=blocks[i - 1] if blocks is not None else None,
                               out_absmax=out_absmax if last else (blocks[i] if blocks is not None else None))
        self._last_blocks = blocks        # (diagnostics, as CostRegNet's)
        return (x, out_absmax) if collect_absmax else x


def _deconv_block(cin, cout):
    return nn.Sequential(
        nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=1, stride=2,
                           bias=False),
        nn.BatchNorm3d(cout),
        nn.ReLU(inplace=True))


class CostRegNet(nn.Module):
    """3D U-Net regulariser [B,32,D,H,W] -> [B,1,D,H,W]  (mvsnet.py:48-93)"""

    # (name, kind, stride, skip-from)
    _PLAN = (("conv0", "conv", 1), ("conv1", "conv", 2), ("conv2", "conv", 1),
             ("conv3", "conv", 2), ("conv4", "conv", 1), ("conv5", "conv", 2),
             ("conv6", "conv", 1), ("conv7", "deconv", 2), ("conv9", "deconv", 2),
             ("conv11", "deconv", 2))

    def __init__(self):
        super().__init__()
        self.conv0 = ConvBnReLU3D(32, 8)
        self.conv1 = ConvBnReLU3D(8, 16, stride=2)
        self.conv2 = ConvBnReLU3D(16, 16)
        self.conv3 = ConvBnReLU3D(16, 32, stride=2)
        self.conv4 = ConvBnReLU3D(32, 32)
        self.conv5 = ConvBnReLU3D(32, 64, stride=2)
        self.conv6 = ConvBnReLU3D(64, 64)
        self.conv7 = _deconv_block(64, 32)
        self.conv9 = _deconv_block(32, 16)
        self.conv11 = _deconv_block(16, 8)
        self.prob = nn.Conv3d(8, 1, 3, stride=1, padding=1)
        self._hip_cache = {}
        self.conv_impl = ops.IMPL_AUTO

    # -- autograd path (training): PyTorch-ROCm ops, batch-statistics BN
    def forward(self, x):
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        c4 = self.conv4(self.conv3(c2))
        x = self.conv6(self.conv5(c4))
        x = c4 + self.conv7(x)
        x = c2 + self.conv9(x)
        x = c0 + self.conv11(x)
        return self.prob(x)

    # -- autograd path on the HIP conv kernels (forward + input gradients), channels-last
    def forward_train_hip(self, x_cl, conv0_raw=None):
        """x_cl [B,D,H,W,32] -> cost [B,D,H,W]; same graph as forward() (mvsnet.py:83-93).  conv0_raw: conv0's output
        before BatchNorm from ops.variance_conv0_autograd (then x_cl is not used)."""
        from ..train_ops import conv3d_cl, conv_bn_relu_cl

        def blk(name, t, stride=1):
            m = getattr(self, name)
            return conv_bn_relu_cl(t, m.conv, m.bn, False, stride, tag=name)

        def up(name, t, skip):
            m = getattr(self, name)
            return conv_bn_relu_cl(t, m[0], m[1], True, 2, skip=skip, tag=name)

        if conv0_raw is not None and self.conv0.bn.training and self.conv0.bn.momentum is not None:
            c0 = ops.bn_relu_cl(conv0_raw, self.conv0.bn)
        else:
            c0 = blk("conv0", x_cl)
        c2 = blk("conv2", blk("conv1", c0, 2))
        c4 = blk("conv4", blk("conv3", c2, 2))
        t = blk("conv6", blk("conv5", c4, 2))
        t = up("conv7", t, c4)     # c4 + relu(bn(deconv))
        t = up("conv9", t, c2)
        t = up("conv11", t, c0)
        return (conv3d_cl(t, self.prob.weight, tag="prob") + self.prob.bias).squeeze(-1)

    # -- inference path: HIP kernels, BN folded to a per-channel affine
    def _layers(self):
        out = []
        for name, kind, stride in self._PLAN:
            m = getattr(self, name)
            conv, bn = (m.conv, m.bn) if kind == "conv" else (m[0], m[1])
            out.append((name, kind, stride, conv, bn))
        return out

    def _hip_params(self):
        layers = self._layers()
        key = tuple((p._version, p.data_ptr()) for _, _, _, conv, bn in layers
                    for p in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var))
        key += ((self.prob.weight._version, self.prob.weight.data_ptr()),
                (self.prob.bias._version, self.prob.bias.data_ptr()))
        dev = self.prob.weight.device
        cache = self._hip_cache   # {device: (key, params)}: shared by the replicas of nn.DataParallel
        hit = cache.get(dev)
        if hit is not None and hit[0] == key:
            return hit[1]
        params = {}
        with torch.no_grad():
            for name, kind, stride, conv, bn in layers:
                scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
                shift = (bn.bias - bn.running_mean * scale).float().contiguous()
                w = conv.weight.detach().float().contiguous()
                params[name] = dict(weight=w, scale=scale, shift=shift, stride=stride,
                                    transposed=(kind == "deconv"),
                                    packed=ops.pack_conv3d_weight(w, kind == "deconv", stride, split=True))
            if ops.conv_split_enabled():
                params["conv0"]["packed_split"] = ops.pack_conv3d_weight_split(params["conv0"]["weight"])
            if ops.conv0_f16_enabled():     # default: conv0 on the two-piece fp16 kernel (MVS_CONV0_F16=0: the bf16 one)
                params["conv0"]["packed_f16x3"] = ops.pack_conv3d_weight_f16x3(params["conv0"]["weight"])
            w = self.prob.weight.detach().float().contiguous()
            params["prob"] = dict(weight=w, scale=None,
                                  shift=self.prob.bias.detach().float().contiguous(), stride=1,
                                  transposed=False, packed=ops.pack_conv3d_weight(w, False, 1))
        cache[dev] = (key, params)
        return params

    def wants_c8_input(self):
        """True when conv0 runs on the MFMA kernel, which streams the variance
        volume as [B,D,H,C/8,W,8] (see include/mvs_hip.h, MVS_LAYOUT_C8)."""
        return self.conv_impl != ops.IMPL_DIRECT and self._hip_params()["conv0"]["packed"] is not None

    def forward_hip(self, x_cl, in_c8=False, x_absmax=None):
        """x_cl: variance volume, channels-last [B,D,H,W,32] (or 8-channel blocked
        [B,D,H,4,W,8] with in_c8) -> cost [B,D,H,W].  x_absmax: the word the variance op filled with the volume's
        largest magnitude (conv0's operand scale on the fp16 kernel; None: collected by one more pass)."""
        P = self._hip_params()
        handed = isinstance(x_cl, ops.HandedVolume)       # the sweep's hand-over: pieces or fp32, decided on the device
        D, H, W = x_cl.shape[2:] if handed else (x_cl.shape[1], x_cl.shape[2], x_cl.shape[4 if in_c8 else 3])
        if not ops.timing_enabled() and D % 8 == 0 and H % 8 == 0 and W % 8 == 0:
            # one C call for the eleven layers (mvs_costreg_fwd_f32); the per-layer calls below
            # remain for stage timing and for sizes the whole-net entry does not take
            return ops.costreg_forward(x_cl, P, in_c8=in_c8, impl=self.conv_impl, x_absmax=x_absmax)

        # the per-layer chain of mvs_costreg_fwd2_f32: a layer on a two-piece fp16 kernel scales its input by the absmax block the
        # layer in front of it collected (blocks of activations nobody reads that way stay None)
        f16 = ops.split_f16_enabled() and self.conv_impl == ops.IMPL_AUTO
        if handed and not f16:
            raise ops.MvsHipError("CostRegNet.forward_hip: a handed-over volume needs the two-piece layers (MVS_SPLIT_F16, conv_impl auto)")
        blocks = ops.absmax_block(x_cl.device, zero=True, n=10) if f16 else None      # (row 9: the fused tail's flag word)
        blk = (lambda i: blocks[i]) if f16 else (lambda i: None)
        self._last_blocks = blocks        # (diagnostics: scripts/trained_guard_stats.py reads what the range guard judged)

        def run(name, t, skip=None, relu=True, x_abs=None, out_abs=None):
            p = P[name]
            with ops.stage("costreg." + name):
                return ops.conv3d(t, p["weight"], p["scale"], p["shift"], skip, relu, p["transposed"],
                                  p["stride"], channels_last=True, packed=p["packed"],
                                  impl=self.conv_impl, x_absmax=x_abs, out_absmax=out_abs)

        p0 = P["conv0"]
        with ops.stage("costreg.conv0"):
            if handed:
                c0 = ops.conv3d_c8_handed(x_cl, p0["packed_f16x3"], p0["scale"], p0["shift"], True, out_absmax=blk(0))
            elif in_c8 and p0.get("packed_f16x3") is not None and self.conv_impl != ops.IMPL_DIRECT:
                c0 = ops.conv3d_c8_f16x3(x_cl, p0["packed_f16x3"], x_absmax, p0["scale"], p0["shift"], None, True, out_absmax=blk(0))
            elif in_c8 and p0.get("packed_split") is not None and self.conv_impl != ops.IMPL_DIRECT:
                c0 = ops._with_absmax(ops.conv3d_c8_split(x_cl, p0["packed_split"], p0["scale"], p0["shift"], None, True), blk(0))
            else:
                c0 = ops.conv3d(x_cl, p0["weight"], p0["scale"], p0["shift"], None, True, False, 1, channels_last=True,
                                packed=p0["packed"], impl=self.conv_impl, in_c8=in_c8, out_absmax=blk(0))
        t1 = run("conv1", c0, x_abs=blk(0), out_abs=blk(1))
        c2 = run("conv2", t1, x_abs=blk(1), out_abs=blk(7))
        t3 = run("conv3", c2, x_abs=blk(7), out_abs=blk(2))
        c4 = run("conv4", t3, x_abs=blk(2), out_abs=blk(8))
        t5 = run("conv5", c4, x_abs=blk(8), out_abs=blk(3))
        t = run("conv6", t5, x_abs=blk(3), out_abs=blk(4))
        t = run("conv7", t, c4, x_abs=blk(4), out_abs=blk(5))
        t = run("conv9", t, c2, x_abs=blk(5), out_abs=blk(6))
        # conv11 + prob: one kernel, as inside mvs_costreg_fwd3_f32 (the two layers enqueued behind it run only if its range guard
        # declined: no host synchronisation)
        p11 = P["conv11"]
        if f16 and ops.tail_fused_enabled() and ops.f16_companion(p11["packed"]) is not None and tuple(p11["weight"].shape[:2]) == (16, 8):
            if p11.get("packed_tail") is None:
                p11["packed_tail"] = ops.pack_costreg_tail(p11["weight"])
            with ops.stage("costreg.tail"):
                return ops.costreg_tail_guarded(t, blk(6), c0, blk(0), p11, P["prob"], flag=blocks[9])
        t = run("conv11", t, c0, x_abs=blk(6))
        cost = run("prob", t, None, relu=False)     # [B,D,H,W,1]
        return cost.squeeze(-1)


class RefineNet(nn.Module):
    """mvsnet.py:96-114.  Never enabled by the reference drivers
    (train.py:93, eval.py:103 pass refine=False); kept for state_dict parity."""

    def __init__(self):
        super().__init__()
        self.conv1 = ConvBnReLU(4, 32)
        self.conv2 = ConvBnReLU(32, 32)
        self.conv3 = ConvBnReLU(32, 32)
        self.res = ConvBnReLU(32, 1)

    def forward(self, img, depth_init):
        x = torch.cat((F.interpolate(img, size=[128, 160]), depth_init.unsqueeze(1)), 1)
        return depth_init + self.res(self.conv3(self.conv2(self.conv1(x))))


class MVSNet(nn.Module):
    def __init__(self, refine=True, align_corners=False, proj_where="host"):
        super().__init__()
        self.refine = refine
        self.align_corners = align_corners
        self.proj_where = proj_where
        self.train_proj_where = "device"
        self.train_conv0_fused = True   # training: variance -> conv0 as one autograd node on the bf16 split-operand kernel
        self.variance_impl = "lds"      # "lds" (LDS-staged source tiles) | "gather"
        # True (eval default): MVS_SWEEP_FAST -- sampling positions within ~1e-4 texel of the
        # reference's, 0.25 ms less per view; the depth map stays as close to the reference's CPU
        # forward as with False (7.3e-4 mm at configs[1] either way, tests/test_gpu_fullsize_reference.py).
        # False: the reference's coordinate arithmetic op for op, variance volume bit-identical.
        self.variance_fast = True
        self.feature_impl = "hip"       # "hip" (2D MFMA kernels) | "torch" (PyTorch-ROCm / MIOpen)
        self.train_impl = "hip"         # CostRegNet autograd convs: "hip" (MFMA fwd+dgrad) | "torch"
        # FeatureNet autograd: "hip" (default since round 3) = the HIP 2D kernels forward, input gradient and weight
        # gradient (mvs_amd.train_ops.conv2d_cl) + the fused HIP BatchNorm/ReLU; "torch_cl" = MIOpen's NHWC 2D
        # kernels + the fused HIP BatchNorm/ReLU (A/B); "torch" = plain nn modules
        self.train_feature_impl = "hip"
        # True: the V per-view FeatureNet calls of a training step run as ONE batch with per-view BatchNorm statistics
        # (FeatureNet.forward_train_hip(groups=V)); False: V calls, as the reference's loop (A/B)
        self.train_feature_batched = True
        self._feature_cl = False
        self.feature = FeatureNet()
        self.cost_regularization = CostRegNet()
        if self.refine:
            self.refine_network = RefineNet()

    def extract_features(self, imgs_flat, chunk=7):
        """FeatureNet of a batch of images [N,3,H,W] on the inference kernels -> 4-channel blocked maps [N,8,H/4,W/4,4]
        (the sweep kernel's input layout), or None when this model / shape does not run that path.  An eval driver
        that keeps a scan's images on the device calls this once per image and hands forward() the maps of a sample's
        views (`features=`): the reference runs FeatureNet on all V views of every sample (mvsnet.py:146), i.e. ~V
        times per image of a scan; eval-mode FeatureNet is per image, so the maps are the same bits either way."""
        if self.training or not (self.feature_impl == "hip" and self.feature.hip_supported()
                                 and self.variance_impl == "lds" and ops.conv2d_persistent_enabled()):
            return None
        with torch.no_grad():
            return torch.cat([self.feature.forward_hip(imgs_flat[i:i + chunk], out_c4=True)
                              for i in range(0, imgs_flat.shape[0], chunk)])

    def forward(self, imgs, proj_matrices, depth_values, features=None):
        """features (inference only): [B,V,8,h,w,4] from extract_features() -- FeatureNet is then skipped."""
        if imgs.shape[1] != proj_matrices.shape[1]:
            raise AssertionError("Different number of images and projection matrices")
        V = imgs.shape[1]
        autograd_path = self.training or (torch.is_grad_enabled() and
                                          any(p.requires_grad for p in self.parameters()))
        if autograd_path and features is not None:
            raise ops.MvsHipError("forward: precomputed features are an inference-path input (model.eval(), torch.no_grad())")
        if autograd_path:
            # training: rot_trans on the device (mvs_rot_trans_f32, float64 internally; no host hop -- the hop's D2H /
            # host function / H2D cost the launching thread ~3 ms of a 13 ms step once FeatureNet ran on the HIP
            # kernels).  train_proj_where = "host" restores the reference's float32 LAPACK inverse (stream-ordered hop).
            where = self.train_proj_where if proj_matrices.is_cuda else self.proj_where
            rt_job = ops.HostRotTrans(proj_matrices) if where == "host" and proj_matrices.is_cuda else None
            with ops.stage("feature"):
                # per-view calls: BatchNorm batch statistics are per call in the
                # reference (mvsnet.py:146)
                feats_cl = feats_stacked = None
                if self.train_feature_impl == "hip" and self.feature.hip_supported():
                    if self.train_impl == "hip":
                        from .. import train_ops
                        train_ops.prepare_step(self.feature, self.cost_regularization)   # the step's packs up front, the bf16 ones as one launch
                    if self.train_feature_batched:
                        # the V per-view calls as one batch of V groups (view-major): per-view BatchNorm statistics, one launch per layer
                        Bn = imgs.shape[0]
                        fall = self.feature.forward_train_hip(imgs.transpose(0, 1).reshape(V * Bn, *imgs.shape[2:]), groups=V)
                        feats_stacked = fall.reshape(V, Bn, *fall.shape[1:])                           # [V,B,h,w,C]
                        feats_cl = list(feats_stacked.unbind(0))
                    else:
                        feats_cl = [self.feature.forward_train_hip(imgs[:, v]) for v in range(V)]   # [B,h,w,C]
                    feats = [f.permute(0, 3, 1, 2) for f in feats_cl]
                elif self.train_feature_impl == "torch_cl" and self.feature.training:
                    if not self._feature_cl:   # weights in NHWC once, not re-laid-out by every conv2d call
                        self.feature.to(memory_format=torch.channels_last)
                        self._feature_cl = True
                    feats_cl = [self.feature.forward_train_cl(imgs[:, v]) for v in range(V)]   # [B,h,w,C]
                    feats = [f.permute(0, 3, 1, 2) for f in feats_cl]
                else:
                    feats = [self.feature(imgs[:, v]) for v in range(V)]
            with ops.stage("rot_trans"):
                rts = rt_job.result() if rt_job is not None else \
                    ops.rot_trans_all(proj_matrices, where)   # [V-1,B,12]
            C = feats[0].shape[1]
            if self.train_impl == "hip" and C % 16 == 0 and depth_values.dim() == 2:
                # channels-last all the way: 16-channel-blocked maps (torch layout ops, in the
                # autograd graph) -> DMA sweep kernel -> [B,D,h,w,C] for the conv kernels;
                # backward on the LDS-accumulating kernel
                if feats_cl is not None:
                    f16 = feats_stacked if feats_stacked is not None else torch.stack(feats_cl)   # [V,B,h,w,C]
                    f16 = f16.reshape(*f16.shape[:4], C // 16, 16).permute(0, 1, 4, 2, 3, 5).contiguous()
                else:
                    f16 = torch.stack(feats)                            # [V,B,C,h,w]
                    f16 = f16.reshape(V, f16.shape[1], C // 16, 16, *f16.shape[3:]).permute(0, 1, 2, 4, 5, 3).contiguous()
                bn0 = self.cost_regularization.conv0.bn
                # (the fused node hands over conv0's RAW output, which only the batch-statistics BatchNorm op consumes: a frozen
                # conv0.bn -- model.train() followed by bn.eval(), or momentum=None -- takes the unfused layers)
                if C == 32 and self.training and self.train_conv0_fused and ops.conv_split_enabled() and bn0.training and bn0.momentum is not None:
                    # warp + variance -> conv0 as one autograd node: the volume stays 8-channel blocked for the bf16 kernel
                    c0 = ops.variance_conv0_autograd(f16[0], f16[1:], rts, depth_values,
                                                     self.cost_regularization.conv0.conv.weight, self.align_corners)
                    cost = self.cost_regularization.forward_train_hip(None, conv0_raw=c0)
                else:
                    var = ops.costvol_variance_c16_autograd(f16[0], f16[1:], rts, depth_values, self.align_corners)
                    cost = self.cost_regularization.forward_train_hip(var)
            else:
                var = ops.costvol_variance(feats[0], torch.stack(feats[1:]), rts, depth_values,
                                           self.align_corners)          # [B,32,D,h,w]
                if self.train_impl == "hip":
                    cost = self.cost_regularization.forward_train_hip(var.permute(0, 2, 3, 4, 1))
                else:
                    cost = self.cost_regularization(var).squeeze(1)
        else:
            B = imgs.shape[0]
            # the host hop of rot_trans runs while FeatureNet occupies the GPU
            rt_job = ops.HostRotTrans(proj_matrices) if self.proj_where == "host" else None
            flat = imgs.reshape(B * V, *imgs.shape[2:])
            # eval: running-stat BN is per-sample, so all B*V views go through FeatureNet
            # as one batch (same values as the reference's per-view loop, mvsnet.py:146)
            half = lambda n: (n - 1) // 2 + 1            # a 5x5 stride-2 layer with padding 2
            h, w, C = half(half(imgs.shape[3])), half(half(imgs.shape[4])), 32
            use_lds = self.variance_impl == "lds"
            # shared depth planes: the persistent sweep kernel copies 4-channel blocked maps fastest, and
            # FeatureNet's last layer writes them directly (a lane of its MFMA epilogue holds 4 channels)
            c4 = use_lds and ops.variance_persistent_supported(depth_values, B, V, C, h, w)
            f4 = None
            # the variance volume leaves the sweep as two fp16 pieces per value where conv0 reads them
            p0 = self.cost_regularization._hip_params()["conv0"] if c4 else None
            hand_over = (c4 and ops.handover_enabled() and self.cost_regularization.wants_c8_input() and
                         self.cost_regularization.conv_impl != ops.IMPL_DIRECT and p0.get("packed_f16x3") is not None)
            fea_amax = None
            if features is not None:
                if tuple(features.shape) != (B, V, C // 4, h, w, 4):
                    raise ops.MvsHipError(f"forward: features {tuple(features.shape)} do not fit this sample "
                                          f"({(B, V, C // 4, h, w, 4)})")
                if c4:
                    f4 = features.reshape(B * V, C // 4, h, w, 4)
                else:      # a sweep kernel that takes other blockings (few depth planes, per-pixel hypotheses)
                    f = features.permute(0, 1, 3, 4, 2, 5).reshape(B * V, h, w, C)
            elif self.feature_impl == "hip" and self.feature.hip_supported():
                if c4 and ops.conv2d_persistent_enabled():
                    # (the last layer collects the maps' largest magnitude: the bound a hand-over sweep scales its pieces by)
                    if hand_over:
                        f4, fea_amax = self.feature.forward_hip(flat, out_c4=True, collect_absmax=True)     # [B*V,8,h,w,4]
                    else:
                        f4 = self.feature.forward_hip(flat, out_c4=True)
                else:
                    f = self.feature.forward_hip(flat)                   # [B*V,h,w,32]: HIP 2D MFMA kernels
            else:
                with ops.stage("feature"):
                    # PyTorch-ROCm in channels_last: MIOpen's NHWC kernels are the faster
                    # ones here and hand over the layout the sweep kernel wants.  The
                    # NCHW->NHWC image conversion goes through the HIP transpose (torch's
                    # strided copy of a 3-channel image costs ~2 ms).
                    if not self._feature_cl:
                        self.feature.to(memory_format=torch.channels_last)
                        self._feature_cl = True
                    f = self.feature(ops.nchw_to_nhwc(flat).permute(0, 3, 1, 2))
                    f = f.permute(0, 2, 3, 1)                # [B*V,h,w,32] view of NHWC storage
            with ops.stage("rot_trans"):
                rts = rt_job.result() if rt_job is not None else \
                    ops.rot_trans_all(proj_matrices, self.proj_where)   # [V-1,B,12]
            c8 = self.cost_regularization.wants_c8_input()
            with ops.stage("to_channels_last"):
                if f4 is not None:     # [B*V,8,h,w,4] -> [V,B,8,h,w,4]: a view for one sample
                    f16 = f4.reshape(B, V, C // 4, h, w, 4).transpose(0, 1).contiguous()
                elif use_lds:
                    # [B*V,h,w,C] -> [V,B,C/blk,h,w,blk]: 4-channel blocks for the persistent sweep
                    # kernel (shared depth planes), 16-channel blocks for the per-tile kernels
                    blk = 4 if c4 else 16
                    f16 = f.reshape(B, V, h, w, C // blk, blk).permute(1, 0, 4, 2, 3, 5).contiguous()
                else:
                    fcl = f.reshape(B, V, h, w, C).transpose(0, 1).contiguous()   # [V,B,h,w,C]
            amax = None
            with ops.stage("costvol_variance"):
                var = None
                if use_lds and hand_over and c8:
                    if fea_amax is None:         # maps that came without their block (features=, stage timing): one pass over them
                        fea_amax = ops.absmax(f16)
                    var = ops.costvol_variance_handover(f16[0], f16[1:], rts, depth_values, fea_amax, self.align_corners,
                                                        fast=self.variance_fast, veto=ops.conv0_veto_word(p0["packed_f16x3"], C))
                if var is not None:
                    self._last_handover_words = (var.hand, var.absmax, var.redo)      # (diagnostics: three small device tensors)
                elif use_lds:
                    if c8 and ops.conv0_f16_enabled():   # the sweep kernels collect conv0's operand scale as they store
                        amax = ops.absmax_block(f16.device)
                    var = ops.costvol_variance_c16(f16[0], f16[1:], rts, depth_values,
                                                   self.align_corners, out_c8=c8, fast=self.variance_fast, absmax_out=amax)
                else:
                    var = ops.costvol_variance_cl(fcl[0], fcl[1:], rts, depth_values,
                                                  self.align_corners, out_c8=c8)
            cost = self.cost_regularization.forward_hip(var, in_c8=c8, x_absmax=amax)  # [B,D,h,w]
        with ops.stage("softmax_regress_conf"):
            depth, conf, _ = ops.softmax_regress_conf(cost, depth_values)
        out = {"depth": depth, "photometric_confidence": conf}
        if self.refine:
            out["refined_depth"] = self.refine_network(imgs[:, 0], depth)
        return out


def mvsnet_loss(depth_est, depth_gt, mask):
    """mvsnet.py:201-203: smooth-L1 (beta 1), mean over mask > 0.5.  On the GPU the mean is taken
    as sum / count over the full map: boolean indexing (the reference's `depth_est[mask]`) needs
    the element count on the host, i.e. a device sync in the middle of every training step."""
    m = mask > 0.5
    if not depth_est.is_cuda:
        return F.smooth_l1_loss(depth_est[m], depth_gt[m], reduction="mean")
    per_pixel = F.smooth_l1_loss(depth_est, depth_gt, reduction="none")
    return torch.where(m, per_pixel, torch.zeros_like(per_pixel)).sum() / m.sum()


def load_reference_checkpoint(model, ckpt):
    """Accepts the reference's checkpoint dicts (train.py:159-164): {'model':
    state_dict, ...} with or without the DataParallel `module.` prefix."""
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=True)
