"""CasMVSNet stage (DepthNet) on the HIP kernels: SURVEY.md 8(a) row a9.

One cascade stage of CasMVSNet/models/cas_mvsnet.py:12-66 -- projection matrices
composed from (extrinsic, intrinsic) pairs (:30-33), per-pixel depth hypotheses
[B,D,H,W] (module.py:249,267), variance cost volume, the parametrised CostRegNet
(module.py:407-438: same topology as MVSNet's with `base_channels`, `prob` without
bias), softmax regression and the index-clamped confidence (:63).  The cascade glue
around it (hypothesis ranges, resizes, FPN features) is a "next" row.
"""
import torch

from .. import ops

_CONVS = (("conv0", 1), ("conv1", 2), ("conv2", 1), ("conv3", 2), ("conv4", 1), ("conv5", 2),
          ("conv6", 1))
_DECONVS = ("conv7", "conv9", "conv11")


def compose_cas_proj(cas_proj):
    """[B,V,2,4,4] (extrinsic, intrinsic) -> [B,V,4,4] with top 3x4 = K @ E[:3,:4]
    (cas_mvsnet.py:30-33), evaluated with torch on the tensor's device."""
    E, K = cas_proj[:, :, 0], cas_proj[:, :, 1]
    P = E.clone()
    P[:, :, :3, :4] = torch.matmul(K[:, :, :3, :3], E[:, :, :3, :4])
    return P


def pack_costreg(sd, prefix=""):
    """Fold BatchNorm(eval) and pack the weights of a Cas-style CostRegNet state_dict
    (keys `convN.conv.weight`, `convN.bn.*`, `prob.weight`)."""
    P = {}
    eps = 1e-5
    with torch.no_grad():
        for name, stride, tr in [(n, s, False) for n, s in _CONVS] + [(n, 2, True) for n in _DECONVS]:
            w = sd[f"{prefix}{name}.conv.weight"].float().contiguous()
            g, b = sd[f"{prefix}{name}.bn.weight"], sd[f"{prefix}{name}.bn.bias"]
            mu, var = sd[f"{prefix}{name}.bn.running_mean"], sd[f"{prefix}{name}.bn.running_var"]
            scale = (g / torch.sqrt(var + eps)).float().contiguous()
            P[name] = dict(weight=w, scale=scale, shift=(b - mu * scale).float().contiguous(),
                           stride=stride, transposed=tr, packed=ops.pack_conv3d_weight(w, tr, stride, split=True))
        if ops.conv_split_enabled():
            P["conv0"]["packed_split"] = ops.pack_conv3d_weight_split(P["conv0"]["weight"])
        if ops.conv0_f16_enabled():
            P["conv0"]["packed_f16x3"] = ops.pack_conv3d_weight_f16x3(P["conv0"]["weight"])
        w = sd[f"{prefix}prob.weight"].float().contiguous()
        bias = sd.get(f"{prefix}prob.bias")
        P["prob"] = dict(weight=w, scale=None, shift=None if bias is None else bias.float().contiguous(),
                         stride=1, transposed=False, packed=ops.pack_conv3d_weight(w, False, 1))
    return P


def costreg_forward(x_cl, P, impl=ops.IMPL_AUTO, in_c8=False, x_absmax=None):
    """x_cl [B,D,H,W,Cin] channels-last (in_c8: [B,D,H,Cin/8,W,8]) -> cost [B,D,H,W]
    (module.py:429-438)."""
    D, H, W = x_cl.shape[1], x_cl.shape[2], x_cl.shape[4 if in_c8 else 3]
    if D % 8 == 0 and H % 8 == 0 and W % 8 == 0:
        return ops.costreg_forward(x_cl, P, in_c8=in_c8, impl=impl, x_absmax=x_absmax)   # one C call: mvs_costreg_fwd2_f32

    def run(name, t, skip=None, relu=True, c8=False):
        p = P[name]
        if c8 and p.get("packed_f16x3") is not None and impl != ops.IMPL_DIRECT:
            return ops.conv3d_c8_f16x3(t, p["packed_f16x3"], x_absmax, p["scale"], p["shift"], skip, relu)
        if c8 and p.get("packed_split") is not None and impl != ops.IMPL_DIRECT:
            return ops.conv3d_c8_split(t, p["packed_split"], p["scale"], p["shift"], skip, relu)
        return ops.conv3d(t, p["weight"], p["scale"], p["shift"], skip, relu, p["transposed"],
                          p["stride"], channels_last=True, packed=p["packed"], impl=impl, in_c8=c8)

    c0 = run("conv0", x_cl, c8=in_c8)
    c2 = run("conv2", run("conv1", c0))
    c4 = run("conv4", run("conv3", c2))
    t = run("conv6", run("conv5", c4))
    t = run("conv7", t, c4)
    t = run("conv9", t, c2)
    t = run("conv11", t, c0)
    return run("prob", t, None, relu=False).squeeze(-1)


def depthnet_forward(features, cas_proj, depth_values, costreg_params, prob_volume_init=None,
                     proj_where="host", tag="", features_cl=False, rts_job=None):
    """features: list of V tensors [B,C,H,W] ([B,H,W,C] with features_cl); cas_proj [B,V,2,4,4]; depth_values [B,D,H,W]
    -> {"depth", "photometric_confidence"} as DepthNet.forward (cas_mvsnet.py:12-66)."""
    # like rot_trans, the K @ E composition is evaluated where the reference's CPU forward
    # evaluates it (host) unless told otherwise: the depth is sensitive to its rounding
    dev = cas_proj.device
    with ops.stage(tag + "rot_trans"):
        if rts_job is not None:          # a host hop started earlier (ops.HostRotTrans(cas_pairs=True))
            rts = rts_job.result()
        else:
            proj = compose_cas_proj(cas_proj.cpu() if proj_where == "host" else cas_proj)
            rts = ops.rot_trans_all(proj, proj_where, device=dev)
    with ops.stage(tag + "to_channels_last"):
        if features_cl:   # [B,H,W,C] already (the HIP FeatureNet's layout); a [V,B,H,W,C] tensor as is
            fcl = features if torch.is_tensor(features) else torch.stack([f for f in features])
        else:
            fcl = torch.stack([ops.nchw_to_nhwc(f) for f in features])      # [V,B,H,W,C]
        C = fcl.shape[-1]
        use_dma = C % 16 == 0    # the LDS/DMA sweep kernel takes 16-channel-blocked maps
        if use_dma:              # [V,B,H,W,C] -> [V,B,C/blk,H,W,blk] (a view when C = blk)
            # shared depth planes (the first stage): 4-channel blocks for the persistent kernel
            blk = 4 if ops.variance_persistent_supported(depth_values, fcl.shape[1], fcl.shape[0], C,
                                                         fcl.shape[2], fcl.shape[3]) else 16
            f16 = fcl.reshape(*fcl.shape[:4], C // blk, blk).permute(0, 1, 4, 2, 3, 5).contiguous()
    amax = None
    with ops.stage(tag + "costvol_variance"):
        if use_dma:
            if costreg_params["conv0"].get("packed_f16x3") is not None:   # conv0's operand scale, collected by the sweep kernel
                amax = ops.absmax_block(fcl.device)
            var = ops.costvol_variance_c16(f16[0], f16[1:], rts, depth_values, out_c8=True, absmax_out=amax)
        else:
            var = ops.costvol_variance_cl(fcl[0], fcl[1:], rts, depth_values, out_c8=True)
    with ops.stage(tag + "costreg"):
        cost = costreg_forward(var, costreg_params, in_c8=True, x_absmax=amax)
        if prob_volume_init is not None:
            cost = cost + prob_volume_init
    with ops.stage(tag + "softmax_regress_conf"):
        depth, conf, _ = ops.softmax_regress_conf(cost, depth_values, clamp_idx=True)
    return {"depth": depth, "photometric_confidence": conf}
