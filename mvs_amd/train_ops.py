"""Training-path (autograd) 3D convolution on the HIP kernels, channels-last.

MIOpen's fp32 3D convolutions dominate a training step of the reference model on this
stack (1.45 s per 640x512x192 sample: `naive_conv_*_ncdhw` and im2col GEMMs).  Here the
forward AND the input gradient of every CostRegNet layer run on the MFMA kernels of
libmvs_hip.so -- the input gradient of a convolution is again one of the supported
layer shapes:

    conv   k3 s1 (Ci->Co)   dgrad = conv   k3 s1 (Co->Ci), weights flipped + transposed
    conv   k3 s2 (Ci->Co)   dgrad = deconv k3 s2 (Co->Ci), same weight tensor
    deconv k3 s2 (Ci->Co)   dgrad = conv   k3 s2 (Co->Ci), same weight tensor

The weight gradient is its own MFMA kernel (csrc/conv3d_wgrad.hip); BatchNorm (batch
statistics) + ReLU + skip add are one fused op (csrc/bnorm.hip).
"""
import torch
import torch.nn.functional as F

from . import ops


import weakref

_derived = {}           # eager use
_derived_capture = {}   # while a HIP graph is being captured (see _cached)
_capture_state = {"id": 0}


def _capture_id():
    """0 when the current stream is not capturing, else the id of the capture it is in (hipStreamGetCaptureInfo through
    mvs_stream_capture_id): two captures with no eager call between them are still two captures."""
    import ctypes
    if not torch.cuda.is_current_stream_capturing():
        return 0
    cid = ctypes.c_ulonglong(0)
    ops.check(ops._lib.load().mvs_stream_capture_id(ops.stream(), ctypes.byref(cid)), "mvs_stream_capture_id")
    return int(cid.value) or 1


def _cached(weight, kind, make):
    """Tensors derived from a layer's weight (packed fragments, flipped / transposed / parity-class weights), made once
    per weight VERSION: a step uses each of them in the forward and the backward of every view, and re-deriving them
    (a flip, a permute, a pack launch, ...) per use was hundreds of tiny launches per step.

    An entry belongs to ONE tensor object: it holds a weak reference to the weight and a hit needs `ref() is weight` and
    the same `_version` (the storage address alone is recycled by the caching allocator when a model is freed
    and another one built, with the same version count after the same init sequence); the entry goes when the weight does.
    What bumps `_version` is an in-place op on the parameter itself (every torch optimizer's step).  Updates written
    through `p.data` (`p.data.copy_()`, `p.data.add_()`: hand-made EMA / clipping code) do NOT -- call `invalidate_derived()`
    after them.
    While the stream is capturing a HIP graph the eager entries are not used: the graph must CONTAIN the kernels that
    derive the packs from the weights it updates, or its replays would run with the packs of capture time.  Entries made
    during a capture live in their own table (the forward and the backward of the captured step still share them), which
    belongs to ONE capture: it is dropped as soon as a call arrives under another capture id."""
    cid = _capture_id() if weight.is_cuda else 0
    capturing = cid != 0
    if capturing and cid != _capture_state["id"]:
        _derived_capture.clear()
        _capture_state["id"] = cid
    table = _derived_capture if capturing else _derived
    key = (id(weight), kind)
    hit = table.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == weight._version:
        return hit[2]
    val = make()
    ref = weakref.ref(weight, lambda _r, k=key, t=table: t.pop(k, None))
    table[key] = (ref, weight._version, val)
    return val


def invalidate_derived():
    """Drop every cached derived tensor (after weight updates that bypass autograd's version counter, e.g. `p.data.copy_()`)."""
    _derived.clear()
    _derived_capture.clear()


def _pk3(weight, transposed, stride):
    """Forward pack of a 3D layer (split-operand bf16 kernels where the shape has one)."""
    return _cached(weight, ("pk3", transposed, stride),
                   lambda: ops.pack_conv3d_weight(weight.detach().contiguous(), transposed, stride, split=True, f16=False, lazy=True))


def _dgrad3(weight, transposed, stride):
    """What the input gradient of a 3D layer runs on: (conv weight, pack) of the flipped / transposed weights for a stride-1
    convolution, the pack of the weight read as a transposed convolution for a stride-2 one, as a convolution for a
    transposed layer."""
    w = weight.detach()
    if not transposed and stride == 1:
        return _cached(weight, "dgrad3_s1", lambda: (lambda t: (t, ops.pack_conv3d_weight(t, False, 1, split=True, f16=False, lazy=True)))(
            w.flip(2, 3, 4).permute(1, 0, 2, 3, 4).contiguous()))        # (Ci,Co,k) as a conv weight
    if not transposed:
        return _cached(weight, "dgrad3_s2", lambda: ops.pack_conv3d_weight(w.contiguous(), True, 2, split=True, f16=False, lazy=True))
    return _cached(weight, ("dgrad3_t", stride), lambda: ops.pack_conv3d_weight(w.contiguous(), False, stride))


def _dgrad2_s1(weight):
    w = weight.detach()
    return _cached(weight, "dgrad2_s1", lambda: ops.pack_conv2d_weight(
        w.flip(2, 3).permute(1, 0, 2, 3).contiguous(), 1, split=True, f16=False, lazy=True))


def _dgrad2_parity(weight):
    return _cached(weight, "dgrad2_parity", lambda: [ops.pack_conv2d_weight(wc, 1) for wc in _parity_weights(weight.detach())])


def _pk2(weight, stride):
    """Forward pack of a 2D layer."""
    return _cached(weight, ("pk2", stride),
                   lambda: ops.pack_conv2d_weight(weight.detach().contiguous(), stride, split=True, f16=False, lazy=True))


def prepare_step(feature, costreg):
    """Every weight-derived tensor a training step will ask for -- FeatureNet's and CostRegNet's forward packs, every
    layer's input-gradient weights and packs -- made up front with the bf16 split packs recorded and run as ONE launch
    (ops.pack_batch: 26 launches of a few microseconds of work otherwise).  Entries land in the same cache the layers look
    in, so a layer this pass does not know simply makes its own on first use.  MVS_TRAIN_PREPARE=0 switches it off."""
    import os
    if os.environ.get("MVS_TRAIN_PREPARE", "1") == "0":
        return
    with ops.pack_batch():
        for name, stride in feature._PLAN:
            w = getattr(feature, name).conv.weight
            cout, cin, k, _ = w.shape
            _pk2(w, stride)
            if name == feature._PLAN[0][0]:
                continue                       # the image needs no gradient
            if stride == 1 and ops.conv2d_supported(cout, cin, k, 1):
                _dgrad2_s1(w)
            elif stride == 2 and k == 5 and ops.conv2d_supported(cout, cin, 3, 1):
                _dgrad2_parity(w)
        w = feature.feature.weight
        _pk2(w, 1)
        if ops.conv2d_supported(w.shape[0], w.shape[1], w.shape[2], 1):
            _dgrad2_s1(w)
        for name in ("conv1", "conv2", "conv3", "conv4", "conv5", "conv6"):
            m = getattr(costreg, name)
            stride = m.conv.stride[0]
            _pk3(m.conv.weight, False, stride)
            _dgrad3(m.conv.weight, False, stride)
        for name in ("conv7", "conv9", "conv11"):
            w = getattr(costreg, name)[0].weight
            _pk3(w, True, 2)
            _dgrad3(w, True, 2)
        _pk3(costreg.prob.weight, False, 1)
        _dgrad3(costreg.prob.weight, False, 1)


class _Conv3dCL(torch.autograd.Function):
    """x [B,D,H,W,Ci] -> raw convolution output [B,Do,Ho,Wo,Co] (no bias, no affine)."""

    @staticmethod
    def forward(ctx, x, weight, transposed, stride, tag):
        x = x.contiguous()
        w = weight.detach().contiguous()
        # (split-operand bf16 kernels where the shape has one)
        packed = _pk3(weight, transposed, stride)
        if ops.split_companion(packed) is not None:
            ops.split_stage_names.add(f"train.{tag}.fwd")
        with ops.stage(f"train.{tag}.fwd"):
            out = ops.conv3d(x, w, None, None, None, False, transposed, stride, channels_last=True,
                             packed=packed)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (transposed, stride, tag)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        transposed, stride, tag = ctx.cfg
        g = g.contiguous()
        w = weight.detach()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if not transposed and stride == 1:
                wt, pk = _dgrad3(weight, False, 1)
                if ops.split_companion(pk) is not None:
                    ops.split_stage_names.add(f"train.{tag}.dgrad")
                with ops.stage(f"train.{tag}.dgrad"):
                    gx = ops.conv3d(g, wt, channels_last=True, packed=pk)
            elif not transposed:
                if any(s % 2 for s in x.shape[1:4]):
                    raise ops.MvsHipError("stride-2 conv backward needs even D, H, W")
                wc = w.contiguous()                                            # (Co,Ci,k) as a deconv weight
                pk = _dgrad3(weight, False, 2)
                if ops.split_companion(pk) is not None:
                    ops.split_stage_names.add(f"train.{tag}.dgrad")
                with ops.stage(f"train.{tag}.dgrad"):
                    gx = ops.conv3d(g, wc, transposed=True, stride=2, channels_last=True, packed=pk)
            else:
                wc = w.contiguous()                                            # (Ci,Co,k) as a conv weight
                pk = _dgrad3(weight, True, stride)
                with ops.stage(f"train.{tag}.dgrad"):
                    gx = ops.conv3d(g, wc, stride=stride, channels_last=True, packed=pk)
        if ctx.needs_input_grad[1]:
            with ops.stage(f"train.{tag}.wgrad"):
                gw = _wgrad(x, g, transposed, stride, weight.shape)
        return gx, gw, None, None, None


_SPLIT = 16384   # voxels per split-K slice


def _tall_gemm_t(a, b):
    """a^T @ b for a [N,Ma], b [N,Mb] with N in the millions and Ma, Mb <= 64: as one
    batched GEMM over N/_SPLIT slices + a sum (a single skinny GEMM with K = N runs on a
    handful of CUs: 2.9 ms per tap for conv0, 154 ms per training step in total)."""
    n = a.shape[0]
    pad = (-n) % _SPLIT
    if pad:
        a = F.pad(a, (0, 0, 0, pad))
        b = F.pad(b, (0, 0, 0, pad))
    nb = a.shape[0] // _SPLIT
    return torch.bmm(a.view(nb, _SPLIT, -1).transpose(1, 2), b.view(nb, _SPLIT, -1)).sum(0)


def _wgrad(x, g, transposed, stride, wshape):
    """dW on the HIP weight-gradient kernel (voxels as the MFMA reduction dimension); a shape
    it does not cover falls back to 27 split-K batched GEMMs over strided views in torch --
    still on the GPU, never on the host."""
    gw = ops.conv3d_wgrad(g, x, 2) if transposed else ops.conv3d_wgrad(x, g, stride)
    if gw is not None:
        return gw
    gw = torch.empty(wshape, device=x.device, dtype=torch.float32)
    if not transposed:
        _, Do, Ho, Wo, Co = g.shape
        xp = F.pad(x, (0, 0, 1, 1, 1, 1, 1, 1))
        g2 = g.reshape(-1, Co)
        for kz in range(3):
            for ky in range(3):
                for kx in range(3):
                    xv = xp[:, kz:kz + stride * Do:stride, ky:ky + stride * Ho:stride,
                            kx:kx + stride * Wo:stride, :]
                    gw[:, :, kz, ky, kx] = _tall_gemm_t(g2, xv.reshape(g2.shape[0], -1))
    else:   # out[o] += in[i] w[ci,co,k], o = 2i - 1 + k
        _, D, H, W, Ci = x.shape
        gp = F.pad(g, (0, 0, 1, 1, 1, 1, 1, 1))
        x2 = x.reshape(-1, Ci)
        for kz in range(3):
            for ky in range(3):
                for kx in range(3):
                    gv = gp[:, kz:kz + 2 * D:2, ky:ky + 2 * H:2, kx:kx + 2 * W:2, :]
                    gw[:, :, kz, ky, kx] = _tall_gemm_t(x2, gv.reshape(x2.shape[0], -1))
    return gw


def conv3d_cl(x, weight, transposed=False, stride=1, tag="conv3d"):
    return _Conv3dCL.apply(x, weight, transposed, stride, tag)


def conv_bn_relu_cl(x, conv, bn, transposed=False, stride=1, skip=None, tag="conv3d"):
    """ConvBnReLU3D / deconv block of the reference (module.py:26-33, mvsnet.py:66-79) in
    channels-last with batch statistics when bn.training (running stats are updated); `skip`
    is added after the ReLU (mvsnet.py:89-91).  BatchNorm + ReLU + skip are one fused HIP op
    when the channel count has a kernel, torch ops otherwise."""
    y = conv3d_cl(x, conv.weight, transposed, stride, tag)
    C = y.shape[-1]
    if bn.training and C in (8, 16, 32, 64) and bn.momentum is not None and bn.weight is not None:
        return ops.bn_relu_cl(y, bn, True, skip)
    out = _bn_relu_torch(y, bn)
    return out if skip is None else skip + out


def _bn_relu_torch(y, bn):
    C = y.shape[-1]
    y2 = F.batch_norm(y.reshape(-1, C), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                      bn.training, bn.momentum, bn.eps)
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return F.relu(y2).reshape(y.shape)


# ------------------------------------------------------------------ FeatureNet (2D)
_parity_index = {}


def _parity_weights(w):
    """The input gradient of a 5x5 stride-2 layer (padding 2) as four 3x3 stride-1 convolutions of the output gradient,
    one per parity class (py, px) of the input pixel: gx[2a+py][2b+px][ci] = sum_{co,ty,tx} Wc[ci][co][ty][tx] *
    g[a+ty-1][b+tx-1][co] with Wc[ci][co][ty][tx] = w[co][ci][py+4-2ty][px+4-2tx] (zero where that index leaves 0..4).
    -> [4, Cin, Cout, 3, 3] (one gather)."""
    idx = _parity_index.get(w.device)
    if idx is None:
        k = torch.tensor([[p + 4 - 2 * t for t in range(3)] for p in (0, 1)])          # [parity, tap]
        ok = ((k >= 0) & (k <= 4)).float()
        k = k.clamp(0, 4)
        ky = k[[0, 0, 1, 1]][:, :, None].expand(4, 3, 3)
        kx = k[[0, 1, 0, 1]][:, None, :].expand(4, 3, 3)
        mask = ok[[0, 0, 1, 1]][:, :, None] * ok[[0, 1, 0, 1]][:, None, :]
        idx = _parity_index[w.device] = (ky.to(w.device), kx.to(w.device), mask.to(w.device))
    ky, kx, mask = idx
    wt = w.permute(1, 0, 2, 3)                                   # [Cin, Cout, 5, 5]
    return (wt[:, :, ky, kx] * mask).permute(2, 0, 1, 3, 4).contiguous()


class _Conv2dCL(torch.autograd.Function):
    """x [N,H,W,Ci] channels-last (planar: the [N,3,H,W] image) -> raw convolution output [N,Ho,Wo,Co]: forward, input
    gradient and weight gradient on the HIP kernels (no MIOpen): the input gradient of a 3x3 stride-1 layer is the
    same layer shape with flipped, transposed weights; of a 5x5 stride-2 layer, four 3x3 stride-1 convolutions (one per
    output parity) + mvs_interleave2x2_f32; the weight gradient is mvs_conv2d_wgrad_f32."""

    @staticmethod
    def forward(ctx, x, weight, stride, planar):
        x = x.contiguous()
        w = weight.detach().contiguous()
        cout, cin, k, _ = w.shape
        pk = _pk2(weight, stride)
        with ops.stage("train.feature.fwd"):
            out = ops.conv2d(x, pk, cin, cout, k, stride, None, None, False, planar=planar)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, planar)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, planar = ctx.cfg
        g = g.contiguous()
        w = weight.detach()
        cout, cin, k, _ = w.shape
        gx = gw = None
        if ctx.needs_input_grad[0]:
            H, W = (x.shape[2], x.shape[3]) if planar else (x.shape[1], x.shape[2])
            with ops.stage("train.feature.dgrad"):
                if stride == 1 and ops.conv2d_supported(cout, cin, k, 1) and not planar:
                    pk = _dgrad2_s1(weight)
                    gx = ops.conv2d(g, pk, cout, cin, k, 1)
                elif stride == 2 and k == 5 and not planar and H % 2 == 0 and W % 2 == 0 and \
                        ops.conv2d_supported(cout, cin, 3, 1):
                    N, Ho, Wo, _ = g.shape
                    cls = torch.empty(4, N, Ho, Wo, cin, device=g.device, dtype=torch.float32)
                    pks = _dgrad2_parity(weight)
                    for i, pk in enumerate(pks):
                        ops.conv2d(g, pk, cout, cin, 3, 1, out=cls[i])
                    gx = ops.interleave2x2(cls)
                else:   # odd image sizes: transposed convolution through torch (NHWC views)
                    gx = F.conv_transpose2d(g.permute(0, 3, 1, 2), w, None, stride, k // 2, stride - 1)
                    gx = gx if planar else gx.permute(0, 2, 3, 1).contiguous()
        if ctx.needs_input_grad[1]:
            with ops.stage("train.feature.wgrad"):
                gw = ops.conv2d_wgrad(x, g, k, stride, planar)
        return gx, gw, None, None


def conv2d_cl(x, weight, stride=1, planar=False):
    return _Conv2dCL.apply(x, weight, stride, planar)


def conv2d_bn_relu_cl(x, conv, bn, stride=1, planar=False, groups=1):
    """ConvBnReLU of the reference (module.py:6-13) in channels-last: the HIP convolution + the fused HIP BatchNorm
    (batch statistics when bn.training, running statistics updated) + ReLU.  groups: the images of x are that many
    consecutive blocks (views), each with its own batch statistics -- `groups` calls of the reference's module as one."""
    y = conv2d_cl(x, conv.weight, stride, planar)
    C = y.shape[-1]
    if bn.training and C in (8, 16, 32, 64) and bn.momentum is not None and bn.weight is not None:
        return ops.bn_relu_cl(y, bn, groups=groups)
    outs = []
    for yg in y.chunk(groups, 0):       # torch ops: one module call per group, as the reference
        y2 = F.batch_norm(yg.reshape(-1, C), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                          bn.training, bn.momentum, bn.eps)
        if bn.training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        outs.append(F.relu(y2).reshape(yg.shape))
    return outs[0] if groups == 1 else torch.cat(outs, 0)
