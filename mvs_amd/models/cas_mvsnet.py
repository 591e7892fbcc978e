"""CascadeMVSNet (BASELINE config 3) with every stage's cost-volume path on the HIP kernels.

Mirrors the reference's Python surface (CasMVSNet/models/cas_mvsnet.py:69-164):

    model = CascadeMVSNet(ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1])
    out = model(imgs[B,V,3,H,W], {"stage1": P1, "stage2": P2, "stage3": P3}, depth_values[B,D])
    out["stage1"]["depth"], ..., out["depth"], out["photometric_confidence"]

with P_s [B,V,2,4,4] = (extrinsic, stage intrinsic) pairs (general_eval.py:158-180).  Module
and parameter names equal the reference's, so its checkpoints load unchanged.

What runs where (eval): the FPN FeatureNet (module.py:304-405) and the glue between stages
-- hypothesis ranges (module.py:485-524), the bilinear / trilinear resizes
(cas_mvsnet.py:134-151) -- are PyTorch-ROCm ops ("next" rows); each stage's
DepthNet (cas_mvsnet.py:12-66: warp + variance with per-pixel hypotheses, CostRegNet,
softmax regression, clamped confidence) runs on the kernels via models/cascade.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cascade
from .. import ops


class _CBR2d(nn.Module):
    """conv + BatchNorm + ReLU with the reference's child names `conv` / `bn`."""

    def __init__(self, cin, cout, k, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class _CBR3d(nn.Module):
    def __init__(self, cin, cout, stride=1, transposed=False):
        super().__init__()
        if transposed:
            self.conv = nn.ConvTranspose3d(cin, cout, 3, stride=stride, padding=1, output_padding=1,
                                           bias=False)
        else:
            self.conv = nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn = nn.BatchNorm3d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class FeatureNet(nn.Module):
    """FPN feature pyramid: stage1 [B,32,H/4,W/4], stage2 [B,16,H/2,W/2], stage3 [B,8,H,W]
    (module.py:304-405, arch_mode "fpn", 3 stages)."""

    def __init__(self, base_channels=8):
        super().__init__()
        b = base_channels
        self.conv0 = nn.Sequential(_CBR2d(3, b, 3), _CBR2d(b, b, 3))
        self.conv1 = nn.Sequential(_CBR2d(b, 2 * b, 5, 2, 2), _CBR2d(2 * b, 2 * b, 3), _CBR2d(2 * b, 2 * b, 3))
        self.conv2 = nn.Sequential(_CBR2d(2 * b, 4 * b, 5, 2, 2), _CBR2d(4 * b, 4 * b, 3), _CBR2d(4 * b, 4 * b, 3))
        self.out1 = nn.Conv2d(4 * b, 4 * b, 1, bias=False)
        self.inner1 = nn.Conv2d(2 * b, 4 * b, 1, bias=True)
        self.inner2 = nn.Conv2d(b, 4 * b, 1, bias=True)
        self.out2 = nn.Conv2d(4 * b, 2 * b, 3, padding=1, bias=False)
        self.out3 = nn.Conv2d(4 * b, b, 3, padding=1, bias=False)
        self.out_channels = [4 * b, 2 * b, b]

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        top = self.conv2(c1)
        out = {"stage1": self.out1(top)}
        top = F.interpolate(top, scale_factor=2, mode="nearest") + self.inner1(c1)
        out["stage2"] = self.out2(top)
        top = F.interpolate(top, scale_factor=2, mode="nearest") + self.inner2(c0)
        out["stage3"] = self.out3(top)
        return out

    # -- inference path on the HIP 2D kernels (BatchNorm folded), channels-last features
    def _hip_params(self):
        params = list(self.parameters()) + list(self.buffers())
        key = tuple((p._version, p.data_ptr()) for p in params)
        cache = getattr(self, "_hip_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1]

        def cbr(m, stride):
            w = m.conv.weight.detach().float().contiguous()
            scale = (m.bn.weight / torch.sqrt(m.bn.running_var + m.bn.eps)).float().contiguous()
            shift = (m.bn.bias - m.bn.running_mean * scale).float().contiguous()
            return dict(cin=w.shape[1], cout=w.shape[0], k=w.shape[2], stride=stride, weight=w,
                        packed=ops.pack_conv2d_weight(w, stride, split=True), scale=scale, shift=shift, relu=True)

        def plain(m):
            w = m.weight.detach().float().contiguous()
            return dict(cin=w.shape[1], cout=w.shape[0], k=w.shape[2], stride=1, weight=w,
                        packed=ops.pack_conv2d_weight(w, 1, split=True), scale=None,
                        shift=None if m.bias is None else m.bias.detach().float().contiguous(), relu=False)

        with torch.no_grad():
            P = {"conv0": [cbr(self.conv0[0], 1), cbr(self.conv0[1], 1)],   # ("head": see forward_hip)
                 "conv1": [cbr(self.conv1[0], 2), cbr(self.conv1[1], 1), cbr(self.conv1[2], 1)],
                 "conv2": [cbr(self.conv2[0], 2), cbr(self.conv2[1], 1), cbr(self.conv2[2], 1)],
                 "out1": plain(self.out1), "inner1": plain(self.inner1), "inner2": plain(self.inner2),
                 "out2": plain(self.out2), "out3": plain(self.out3)}
        self._hip_cache = (key, P)
        return P

    def hip_supported(self):
        P = self._hip_params()
        flat = [q for v in P.values() for q in (v if isinstance(v, list) else [v])]
        return all(q["packed"] is not None for q in flat)

    def forward_hip(self, imgs_nchw):
        """[N,3,H,W] -> {"stage1": [N,H/4,W/4,32], "stage2": [N,H/2,W/2,16], "stage3": [N,H,W,8]}
        channels-last (module.py:343-405; the nearest x2 upsample + lateral add ride in the 1x1 kernels)."""
        P = self._hip_params()

        # the 3x3 / 5x5 layers between the head and the top of the pyramid chain their absmax blocks (ops.conv2d): a layer on a
        # two-piece fp16 kernel scales its input by what the layer in front of it left
        nchain = 1 + len(P["conv1"]) + len(P["conv2"])
        blocks = ops.absmax_block(imgs_nchw.device, zero=True, n=nchain) if ops.split_f16_enabled() else None

        def run(x, p, planar=False, coarse=None, xa=None, oa=None):
            return ops.conv2d(x, p["packed"], p["cin"], p["cout"], p["k"], p["stride"], p["scale"], p["shift"],
                              p["relu"], planar=planar, coarse=coarse, x_absmax=xa, out_absmax=oa)

        def lateral(top, x, p):   # F.interpolate(top, x2, nearest) + inner(x)  (module.py:392,396)
            if top.shape[1] * 2 == x.shape[1] and top.shape[2] * 2 == x.shape[2]:
                return run(x, p, coarse=top)            # fused into the 1x1 convolution's epilogue
            up = top.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)   # odd sizes: as the reference
            return up + run(x, p)

        h0, h1 = P["conv0"]
        if (ops.feature_head_enabled() and (h0["cin"], h0["cout"], h1["cout"]) == (3, 8, 8)
                and ops.feature_head_supported(imgs_nchw.shape[2], imgs_nchw.shape[3])):
            if "head" not in h1:
                h1["head"] = ops.pack_feature_head_weight(h1["weight"])
            c0 = ops.feature_head(imgs_nchw, h0["weight"], h0["scale"], h0["shift"], h1["head"], h1["scale"],
                                  h1["shift"], out_absmax=blocks[0] if blocks is not None else None)      # the two 3x3 layers of conv0 in one kernel
        else:
            c0 = run(run(imgs_nchw, h0, planar=True), h1, oa=blocks[0] if blocks is not None else None)
        c1, bi = c0, 0
        for p in P["conv1"]:
            c1 = run(c1, p, xa=blocks[bi] if blocks is not None else None, oa=blocks[bi + 1] if blocks is not None else None)
            bi += 1
        top = c1
        for p in P["conv2"]:
            top = run(top, p, xa=blocks[bi] if blocks is not None else None, oa=blocks[bi + 1] if blocks is not None else None)
            bi += 1
        out = {"stage1": run(top, P["out1"])}
        top = lateral(top, c1, P["inner1"])
        out["stage2"] = run(top, P["out2"])
        i2, o3 = P["inner2"], P["out3"]
        if ((i2["cin"], i2["cout"], o3["cout"]) == (8, 32, 8) and top.shape[1] * 2 == c0.shape[1]
                and top.shape[2] * 2 == c0.shape[2] and ops.fpn_tail_supported(c0.shape[1], c0.shape[2])):
            if "tail" not in o3:
                o3["tail"] = ops.pack_fpn_tail_weight(o3["weight"])
            # the lateral 1x1 layer, the top-down add and out3 in one kernel: the 32-channel full-resolution map stays in LDS
            out["stage3"] = ops.fpn_tail(c0, top, i2["weight"], i2["shift"], o3["tail"], o3["shift"])
        else:
            top = lateral(top, c0, i2)
            out["stage3"] = run(top, o3)
        return out


class CostRegNet(nn.Module):
    """module.py:407-438; holds the parameters (state_dict contract) -- the arithmetic runs in
    cascade.costreg_forward on the kernels."""

    def __init__(self, in_channels, base_channels):
        super().__init__()
        b = base_channels
        self.conv0 = _CBR3d(in_channels, b)
        self.conv1 = _CBR3d(b, 2 * b, 2)
        self.conv2 = _CBR3d(2 * b, 2 * b)
        self.conv3 = _CBR3d(2 * b, 4 * b, 2)
        self.conv4 = _CBR3d(4 * b, 4 * b)
        self.conv5 = _CBR3d(4 * b, 8 * b, 2)
        self.conv6 = _CBR3d(8 * b, 8 * b)
        self.conv7 = _CBR3d(8 * b, 4 * b, 2, transposed=True)
        self.conv9 = _CBR3d(4 * b, 2 * b, 2, transposed=True)
        self.conv11 = _CBR3d(2 * b, b, 2, transposed=True)
        self.prob = nn.Conv3d(b, 1, 3, stride=1, padding=1, bias=False)
        self._packed = None

    def forward(self, x):   # planar autograd form (training)
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        c4 = self.conv4(self.conv3(c2))
        t = self.conv6(self.conv5(c4))
        t = c4 + self.conv7(t)
        t = c2 + self.conv9(t)
        t = c0 + self.conv11(t)
        return self.prob(t)

    def hip_params(self):
        key = tuple((p._version, p.data_ptr()) for p in list(self.parameters()) + list(self.buffers()))
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, cascade.pack_costreg(self.state_dict()))
        return self._packed[1]


def depth_hypotheses(cur_depth, ndepth, interval, shape):
    """Per-pixel hypothesis volume [B,D,H,W] (module.py:485-524).  cur_depth [B,D0] (first
    stage: the full sweep's end points) or [B,H,W] (later stages: +-ndepth/2 intervals around
    the previous estimate)."""
    B, H, W = shape
    ramp = torch.arange(ndepth, device=cur_depth.device, dtype=cur_depth.dtype)
    if cur_depth.dim() == 2:
        lo, hi = cur_depth[:, 0], cur_depth[:, -1]
        step = (hi - lo) / (ndepth - 1)
        d = lo.unsqueeze(1) + ramp.reshape(1, -1) * step.unsqueeze(1)
        return d.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, H, W)
    lo = cur_depth - ndepth / 2 * interval
    hi = cur_depth + ndepth / 2 * interval
    step = (hi - lo) / (ndepth - 1)
    return lo.unsqueeze(1) + ramp.reshape(1, -1, 1, 1) * step.unsqueeze(1)


class CascadeMVSNet(nn.Module):
    def __init__(self, refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4, 2, 1),
                 cr_base_chs=(8, 8, 8), proj_where="host", feature_impl="hip"):
        super().__init__()
        if refine:
            raise NotImplementedError("the reference never enables refine (test.py:168)")
        assert len(ndepths) == len(depth_interals_ratio) == 3
        self.ndepths = list(ndepths)
        self.depth_interals_ratio = list(depth_interals_ratio)
        self.num_stage = 3
        self.proj_where = proj_where
        self.feature_impl = feature_impl   # "hip": 2D MFMA kernels; "torch": PyTorch-ROCm / MIOpen
        self.stage_scale = {"stage1": 4, "stage2": 2, "stage3": 1}
        self.feature = FeatureNet(base_channels=8)
        self.cost_regularization = nn.ModuleList(
            [CostRegNet(self.feature.out_channels[i], cr_base_chs[i]) for i in range(3)])

    def forward(self, imgs, proj_matrices, depth_values):
        B, V, _, H, W = imgs.shape
        depth_min, depth_max = float(depth_values[0, 0]), float(depth_values[0, -1])
        depth_interval = (depth_max - depth_min) / depth_values.size(1)
        # the three stages' host hops (K @ E, inverse, product on the CPU like the reference's CPU
        # forward) depend on the inputs only: start them now, under the feature pyramid
        jobs = {k: ops.HostRotTrans(proj_matrices[k], cas_pairs=True) for k in self.stage_scale} \
            if self.proj_where == "host" else {}
        use_hip = self.feature_impl == "hip" and not self.training and self.feature.hip_supported()
        with ops.stage("feature"):
            if use_hip:   # all views in one view-major batch: the pyramids come out as [V,B,h,w,C]
                pyr = self.feature.forward_hip(imgs.transpose(0, 1).reshape(V * B, 3, H, W))
                pyr = {k: t.reshape(V, B, *t.shape[1:]) for k, t in pyr.items()}
            else:
                feats = [self.feature(imgs[:, v]) for v in range(V)]
        outputs, depth = {}, None
        for s in range(3):
            key = f"stage{s + 1}"
            scale = self.stage_scale[key]
            with ops.stage(key + ".hypotheses"):
                interval = self.depth_interals_ratio[s] * depth_interval
                if depth is None:
                    # first stage: planes shared by all pixels.  The reference repeats them to [B,D,H,W]
                    # and resizes (cas_mvsnet.py:150); a resize of constant planes returns the planes, so
                    # the [B,D] values go to the kernels as shared depth planes
                    lo, hi = depth_values[:, 0], depth_values[:, -1]
                    ramp = torch.arange(self.ndepths[s], device=depth_values.device, dtype=depth_values.dtype)
                    hyp = lo.unsqueeze(1) + ramp.reshape(1, -1) * ((hi - lo) / (self.ndepths[s] - 1)).unsqueeze(1)
                elif use_hip:
                    hyp = ops.cas_depth_hypotheses(depth.detach(), self.ndepths[s], interval, (H, W),
                                                   (H // scale, W // scale))
                else:
                    cur = F.interpolate(depth.detach().unsqueeze(1), [H, W], mode="bilinear",
                                        align_corners=False).squeeze(1)
                    hyp = depth_hypotheses(cur, self.ndepths[s], interval, (B, H, W))
                    hyp = F.interpolate(hyp.unsqueeze(1), [self.ndepths[s], H // scale, W // scale],
                                        mode="trilinear", align_corners=False).squeeze(1).contiguous()
            stage_feats = pyr[key] if use_hip else [f[key] for f in feats]
            if self.training:
                raise NotImplementedError("CascadeMVSNet here is the inference path (config 3)")
            out = cascade.depthnet_forward(stage_feats, proj_matrices[key], hyp,
                                           self.cost_regularization[s].hip_params(),
                                           proj_where=self.proj_where, tag=key + ".",
                                           features_cl=use_hip, rts_job=jobs.get(key))
            depth = out["depth"]
            outputs[key] = out
            outputs.update(out)
        return outputs
