"""Torch-facing wrappers of the libmvs_hip.so C ABI (include/mvs_hip.h).

Tensors in, tensors out; device memory and streams come from PyTorch-ROCm,
every FLOP of the cost-volume path runs in the hand-written HIP kernels.
There is no fallback: CPU tensors or a missing library raise MvsHipError.
"""
import contextlib
import ctypes
import threading

import torch

from . import _lib
from ._lib import (MVS_LAYOUT_C4, MVS_LAYOUT_C8, MVS_LAYOUT_C16, MVS_LAYOUT_NCHW, MVS_LAYOUT_NHWC, MvsHipError, check, ptr,
                   stream)

_I = ctypes.c_int


# ------------------------------------------------------------ stage timing
class StageTimer:
    """HIP-event timing of named stages on torch's current stream (the stream
    every kernel of this package is launched on).  bench.py installs one with
    set_timer(); when none is installed the hooks cost nothing."""

    def __init__(self, only=None):
        self.only = only          # None = every stage, else a set of names
        self.pairs = {}

    def begin(self, name):
        if self.only is not None and name not in self.only:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, name, start):
        if start is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.pairs.setdefault(name, []).append((start, ev))

    def summary_ms(self):
        """name -> (count, mean ms); call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v))
                for k, v in self.pairs.items()}

    def min_ms(self):
        """name -> fastest occurrence (ms): the kernel's own duration, without whatever a
        profiler or a busy host adds between the two events of one occurrence."""
        return {k: min(a.elapsed_time(b) for a, b in v) for k, v in self.pairs.items()}


_tls = threading.local()   # the timer belongs to the thread that installed it: nn.DataParallel runs
                           # one forward per device on worker threads, which must not share event lists


_process_timer = None      # set_timer(t, all_threads=True): also seen by autograd's backward thread


def _get_timer():
    return getattr(_tls, "timer", None) or _process_timer


def set_timer(t, all_threads=False):
    """Install (or with None remove) the stage timer of the calling thread; all_threads: of every thread that has none
    of its own -- autograd runs backward() on its own thread, so a training step is timed with all_threads=True (one
    model per process; nn.DataParallel's per-device worker threads must not share one)."""
    global _process_timer
    _tls.timer = t
    if all_threads or t is None:
        _process_timer = t


def timing_enabled():
    return _get_timer() is not None


class stage:
    """with ops.stage("name"): ...  -- no-op unless a StageTimer is installed."""

    def __init__(self, name):
        self.name = name
        self.ev = None

    def __enter__(self):
        self.timer = _get_timer()
        if self.timer is not None:
            self.ev = self.timer.begin(self.name)
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.end(self.name, self.ev)
        return False


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def rot_trans(src_proj, ref_proj, where="host"):
    """rows of (src_proj @ inverse(ref_proj))[:3,:4] -> [B,12] on src_proj's device.

    This is MVSNet/models/module.py:63-65.  The depth map is sensitive to the
    rounding of this 4x4 fp32 inverse (~2e-4 mm), so by default it is evaluated
    with the same ATen CPU (LAPACK) ops the reference's CPU forward uses
    (`where="host"`: bit-identical, costs one small D2H/H2D round trip);
    `where="device"` keeps it on the GPU (no sync, rounding differs in the
    last bits)."""
    dev = src_proj.device
    with torch.no_grad():
        if where == "host":
            s, r = src_proj.detach().float().cpu(), ref_proj.detach().float().cpu()
        else:
            s, r = src_proj.detach().float(), ref_proj.detach().float()
        m = torch.matmul(s, torch.inverse(r))
        return m[:, :3, :4].reshape(-1, 12).contiguous().to(dev, non_blocking=True)


def rot_trans_all(proj_matrices, where="host", device=None):
    """[B,V,4,4] -> [V-1,B,12]: rot_trans of every source view against view 0 with ONE
    device->host hop and one upload.  Each view's product keeps the reference's exact
    shapes ([B,4,4] @ inverse([B,4,4]), module.py:63), so the values are bit-identical
    to calling rot_trans per view."""
    dev = device if device is not None else proj_matrices.device
    if where == "device" and proj_matrices.is_cuda:
        # mvs_rot_trans_f32: float64 Gauss-Jordan on the device, no synchronisation (torch.inverse checks LAPACK's
        # info word on the host: it cannot be captured into a HIP graph and stalls the launching thread)
        P = _f32c(proj_matrices.detach())
        out = torch.empty(P.shape[1] - 1, P.shape[0], 12, device=P.device, dtype=torch.float32)
        check(_lib.load().mvs_rot_trans_f32(ptr(P), P.shape[0], P.shape[1], ptr(out), stream()), "mvs_rot_trans_f32")
        return out
    with torch.no_grad():
        P = proj_matrices.detach().float()
        if where == "host":
            P = P.cpu()
        inv_ref = torch.inverse(P[:, 0])
        out = torch.stack([torch.matmul(P[:, v], inv_ref)[:, :3, :4].reshape(-1, 12)
                           for v in range(1, P.shape[1])])
        return out.contiguous().to(dev, non_blocking=True)


_side_streams = {}
_state_lock = threading.Lock()   # guards the small per-device caches below (DataParallel worker threads)


def _side_stream(dev):
    key = (dev.type, dev.index)
    with _state_lock:
        side = _side_streams.get(key)
        if side is None:
            side = _side_streams[key] = torch.cuda.Stream(device=dev)
    return side


def _rot_trans_host_math(host):
    """[B,V,4,4] CPU fp32 -> [V-1,B,12]: the reference's own CPU ops (module.py:63-65)."""
    inv_ref = torch.inverse(host[:, 0])
    return torch.stack([torch.matmul(host[:, v], inv_ref)[:, :3, :4].reshape(-1, 12)
                        for v in range(1, host.shape[1])]).contiguous()


class _HostHop:
    """Stream-ordered host hop: D2H -> CPU function -> H2D enqueued on a side stream with
    hipLaunchHostFunc, so the launching thread never blocks on the GPU (it may run many
    reference views ahead; a blocking hop ties it to one, and every host hiccup then
    shows up as GPU idle time).  One instance per (device, shape); the side stream
    serialises the uses of its pinned buffers."""
    _hip = None
    _instances = {}

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import ctypes
            path = None
            with open("/proc/self/maps") as f:
                for line in f:
                    if "libamdhip64" in line:
                        path = line.split()[-1]
                        break
            lib = ctypes.CDLL(path) if path else None   # the runtime torch already mapped
            if lib is None or not hasattr(lib, "hipLaunchHostFunc"):
                cls._hip = False
            else:
                lib.hipLaunchHostFunc.restype = ctypes.c_int
                lib.hipLaunchHostFunc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
                cls._hip = lib
        return cls._hip

    def __init__(self, shape_in, shape_out, fn):
        import ctypes
        self.pin_in = torch.empty(shape_in, dtype=torch.float32).pin_memory()
        self.pin_out = torch.empty(shape_out, dtype=torch.float32).pin_memory()
        self.error = None

        def _cb(_):
            # Runs on a HIP runtime thread and needs the GIL: a Python thread that holds the GIL
            # while it waits for this stream (torch.cuda.synchronize() releases it; a hipFree from
            # torch.cuda.empty_cache() does not) can deadlock against it -- HostRotTrans(blocking=True)
            # is the hop without this hazard.
            try:
                with torch.no_grad():
                    self.pin_out.copy_(fn(self.pin_in))
            except BaseException as e:   # cannot propagate out of a runtime thread
                # the upload behind this callback still runs: poison it, so the forward that
                # consumes it yields NaN depths instead of the previous sample's matrices
                self.pin_out.fill_(float("nan"))
                self.error = e
        self._cb = ctypes.CFUNCTYPE(None, ctypes.c_void_p)(_cb)   # keep alive

    @classmethod
    def get(cls, dev, shape_in, shape_out, fn):
        key = (dev.type, dev.index, tuple(shape_in), fn)
        with _state_lock:
            inst = cls._instances.get(key)
            if inst is None:
                inst = cls._instances[key] = cls(shape_in, shape_out, fn)
        return inst

    def raise_pending(self):
        """Raise the failure of a hop whose callback has already run."""
        if self.error is not None:
            e, self.error = self.error, None
            raise MvsHipError(f"host hop failed (its output was poisoned with NaN): {e!r}")

    def enqueue(self, src, side):
        """On `side` (current stream = side): src (device) -> fn on the host -> new device tensor."""
        import ctypes
        self.raise_pending()
        self.pin_in.copy_(src, non_blocking=True)
        rc = self.hip().hipLaunchHostFunc(ctypes.c_void_p(side.cuda_stream),
                                          ctypes.cast(self._cb, ctypes.c_void_p), None)
        if rc != 0:
            raise MvsHipError(f"hipLaunchHostFunc failed (rc={rc})")
        out = torch.empty(self.pin_out.shape, dtype=torch.float32, device=src.device)
        out.copy_(self.pin_out, non_blocking=True)
        return out


def check_host_hops():
    """After a device synchronisation: raise if any stream-ordered host hop failed (also the
    last sample's, which no later enqueue would report)."""
    with _state_lock:
        hops = list(_HostHop._instances.values())
    for h in hops:
        h.raise_pending()


def _cas_rot_trans_host_math(host):
    """[B,V,2,4,4] (extrinsic, intrinsic) pairs, CPU fp32 -> [V-1,B,12]: K @ E[:3,:4] composed as
    CasMVSNet does (cas_mvsnet.py:30-33), then the reference's rot_trans algebra."""
    E, K = host[:, :, 0], host[:, :, 1]
    P = E.clone()
    P[:, :, :3, :4] = torch.matmul(K[:, :, :3, :3], E[:, :, :3, :4])
    return _rot_trans_host_math(P)


class HostRotTrans:
    """rot_trans_all(where="host") as a stream-ordered job that overlaps GPU work:

        job = HostRotTrans(proj_matrices)      # marks "inputs ready" on the current stream
        ... launch kernels that do not need the result (FeatureNet) ...
        rts = job.result()                     # current stream waits for the hop's upload

    The hop (D2H, the reference's CPU 4x4 algebra, H2D) is enqueued on a side stream that
    waits only for the event recorded at construction, so it does not queue behind the
    kernels launched in between, and -- with hipLaunchHostFunc -- the launching thread
    never waits for the GPU.  blocking=True (or a runtime without hipLaunchHostFunc) does
    the hop synchronously in result().  Values are bit-identical to rot_trans_all."""

    def __init__(self, proj_matrices, blocking=False, cas_pairs=False):
        self.P = proj_matrices.detach()
        self.dev = proj_matrices.device
        self.math = _cas_rot_trans_host_math if cas_pairs else _rot_trans_host_math
        ev = torch.cuda.Event()
        ev.record()
        side = _side_stream(self.dev)
        side.wait_event(ev)
        self.out = self.done = self.hop = None
        if not blocking and _HostHop.hip():
            B, V = self.P.shape[0], self.P.shape[1]
            hop = self.hop = _HostHop.get(self.dev, tuple(self.P.shape), (V - 1, B, 12), self.math)
            with torch.no_grad(), torch.cuda.stream(side):
                self.out = hop.enqueue(self.P.float(), side)
                self.done = torch.cuda.Event()
                self.done.record(side)

    def result(self):
        if self.out is not None:
            # a callback that has already failed is reported here; one that fails later has
            # poisoned this result with NaN and is reported by the next hop / check_host_hops()
            self.hop.raise_pending()
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(self.done)
            self.out.record_stream(cur)   # allocated on the side stream, consumed here
            return self.out
        side = _side_stream(self.dev)
        with torch.no_grad():
            with torch.cuda.stream(side):
                host = self.P.float().to("cpu", non_blocking=True)
            side.synchronize()
            return self.math(host).to(self.dev, non_blocking=True)


def _depth_mode(depth_values):
    if depth_values.dim() == 2:
        return 0
    if depth_values.dim() == 4:
        return 1
    raise MvsHipError(f"depth_values must be [B,D] or [B,D,H,W], got {tuple(depth_values.shape)}")


def nchw_to_nhwc(x):
    """[B,C,*spatial] -> [B,*spatial,C] (contiguous) on the HIP transpose kernel."""
    x = _f32c(x)
    B, C = x.shape[0], x.shape[1]
    S = x[0, 0].numel()
    out = torch.empty((B,) + tuple(x.shape[2:]) + (C,), device=x.device, dtype=torch.float32)
    check(_lib.load().mvs_nchw_to_nhwc_f32(ptr(x), ptr(out), B, C, S, stream()), "mvs_nchw_to_nhwc_f32")
    return out


def nhwc_to_nchw(x):
    x = _f32c(x)
    B, C = x.shape[0], x.shape[-1]
    S = x[0, ..., 0].numel()
    out = torch.empty((B, C) + tuple(x.shape[1:-1]), device=x.device, dtype=torch.float32)
    check(_lib.load().mvs_nhwc_to_nchw_f32(ptr(x), ptr(out), B, C, S, stream()), "mvs_nhwc_to_nchw_f32")
    return out


# ---------------------------------------------------------------- K1 warp
class _HomoWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src_fea, rt, depth_values, align_corners):
        src_fea, depth_values = _f32c(src_fea), _f32c(depth_values)
        B, C, H, W = src_fea.shape
        D = depth_values.shape[1]
        mode = _depth_mode(depth_values)
        out = torch.empty((B, C, D, H, W), device=src_fea.device, dtype=torch.float32)
        check(_lib.load().mvs_warp_fwd_f32(ptr(src_fea), ptr(rt), ptr(depth_values), mode, B, C, D,
                                           H, W, int(align_corners), ptr(out), stream()),
              "mvs_warp_fwd_f32")
        ctx.save_for_backward(rt, depth_values)
        ctx.meta = (B, C, D, H, W, mode, int(align_corners))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rt, depth_values = ctx.saved_tensors
        B, C, D, H, W, mode, ac = ctx.meta
        grad_out = _f32c(grad_out)
        g = torch.empty((B, C, H, W), device=grad_out.device, dtype=torch.float32)
        check(_lib.load().mvs_warp_bwd_f32(ptr(grad_out), ptr(rt), ptr(depth_values), mode, B, C, D,
                                           H, W, ac, ptr(g), stream()), "mvs_warp_bwd_f32")
        return g, None, None, None


def homo_warp(src_fea, rt, depth_values, align_corners=False):
    return _HomoWarp.apply(src_fea, rt, depth_values, align_corners)


# ------------------------------------------------------ K1+K2 fused variance
class _CostVolVariance(torch.autograd.Function):
    """Planar ([B,C,H,W] -> [B,C,D,H,W]) fused warp+variance with backward."""

    @staticmethod
    def forward(ctx, ref_fea, src_feas, rts, depth_values, align_corners, alias_quirk):
        ref_fea, src_feas, depth_values = _f32c(ref_fea), _f32c(src_feas), _f32c(depth_values)
        B, C, H, W = ref_fea.shape
        V = src_feas.shape[0] + 1
        D = depth_values.shape[1]
        mode = _depth_mode(depth_values)
        out = torch.empty((B, C, D, H, W), device=ref_fea.device, dtype=torch.float32)
        check(_lib.load().mvs_costvol_variance_fwd_f32(
            ptr(ref_fea), ptr(src_feas), ptr(rts), ptr(depth_values), mode, B, V, C, D, H, W,
            int(align_corners), int(alias_quirk), MVS_LAYOUT_NCHW, MVS_LAYOUT_NCHW, ptr(out),
            stream()), "mvs_costvol_variance_fwd_f32")
        ctx.save_for_backward(ref_fea, src_feas, rts, depth_values)
        ctx.meta = (B, V, C, D, H, W, mode, int(align_corners), int(alias_quirk))
        return out

    @staticmethod
    def backward(ctx, grad_var):
        ref_fea, src_feas, rts, depth_values = ctx.saved_tensors
        B, V, C, D, H, W, mode, ac, quirk = ctx.meta
        if quirk:
            raise MvsHipError("alias_quirk (CVP inference form) is not differentiable here")
        grad_var = _f32c(grad_var)
        g_ref = torch.empty_like(ref_fea)
        g_src = torch.empty_like(src_feas)
        check(_lib.load().mvs_costvol_variance_bwd_f32(
            ptr(grad_var), ptr(ref_fea), ptr(src_feas), ptr(rts), ptr(depth_values), mode, B, V, C,
            D, H, W, ac, MVS_LAYOUT_NCHW, MVS_LAYOUT_NCHW, ptr(g_ref), ptr(g_src), stream()),
            "mvs_costvol_variance_bwd_f32")
        return g_ref, g_src, None, None, None, None


class _CostVolVarianceC16(torch.autograd.Function):
    """Channels-last training form: 16-channel-blocked feature maps in, [B,D,H,W,C] volume out
    (DMA sweep kernel), backward on the LDS-accumulating kernel.  Shared depth planes only."""

    @staticmethod
    def forward(ctx, ref16, srcs16, rts, depth_values, align_corners):
        ref16, srcs16, depth_values = _f32c(ref16), _f32c(srcs16), _f32c(depth_values)
        if _depth_mode(depth_values) != 0:
            raise MvsHipError("the differentiable channels-last variance takes [B,D] depth planes")
        with stage("train.variance.fwd"):
            out = costvol_variance_c16(ref16, srcs16, rts, depth_values, align_corners)
        ctx.save_for_backward(ref16, srcs16, rts, depth_values)
        ctx.ac = int(align_corners)
        return out

    @staticmethod
    def backward(ctx, grad_var):
        ref16, srcs16, rts, depth_values = ctx.saved_tensors
        B, G, H, W, _ = ref16.shape
        V, D = srcs16.shape[0] + 1, depth_values.shape[1]
        grad_var = _f32c(grad_var)
        g_ref, g_src = torch.empty_like(ref16), torch.empty_like(srcs16)
        with stage("train.variance.bwd"):
            check(_lib.load().mvs_costvol_variance_bwd_f32(
                ptr(grad_var), ptr(ref16), ptr(srcs16), ptr(rts), ptr(depth_values), 0, B, V, G * 16, D, H, W,
                ctx.ac, MVS_LAYOUT_C16, MVS_LAYOUT_NHWC, ptr(g_ref), ptr(g_src), stream()),
                "mvs_costvol_variance_bwd_f32")
        return g_ref, g_src, None, None, None


def costvol_variance_c16_autograd(ref16, srcs16, rts, depth_values, align_corners=False):
    """ref16 [B,C/16,H,W,16]; srcs16 [V-1,B,C/16,H,W,16] -> [B,D,H,W,C], differentiable w.r.t.
    the feature maps."""
    return _CostVolVarianceC16.apply(ref16, srcs16, rts, depth_values, align_corners)


def costvol_variance(ref_fea, src_feas, rts, depth_values, align_corners=False, alias_quirk=False):
    """ref_fea [B,C,H,W]; src_feas [V-1,B,C,H,W]; rts [V-1,B,12] -> [B,C,D,H,W]
    (differentiable w.r.t. the feature maps)."""
    return _CostVolVariance.apply(ref_fea, src_feas, rts, depth_values, align_corners, alias_quirk)


def costvol_variance_cl(ref_fea_cl, src_feas_cl, rts, depth_values, align_corners=False,
                        alias_quirk=False, out_c8=False):
    """Channels-last inference form: ref [B,H,W,C]; srcs [V-1,B,H,W,C] -> [B,D,H,W,C],
    or with out_c8 the 8-channel-blocked volume [B,D,H,C/8,W,8] the conv0 kernel
    streams chunk by chunk."""
    ref_fea_cl, src_feas_cl, depth_values = _f32c(ref_fea_cl), _f32c(src_feas_cl), _f32c(depth_values)
    B, H, W, C = ref_fea_cl.shape
    V = src_feas_cl.shape[0] + 1
    D = depth_values.shape[1]
    if out_c8 and C % 8:
        raise MvsHipError("out_c8 needs C to be a multiple of 8")
    shape = (B, D, H, C // 8, W, 8) if out_c8 else (B, D, H, W, C)
    out = torch.empty(shape, device=ref_fea_cl.device, dtype=torch.float32)
    check(_lib.load().mvs_costvol_variance_fwd_f32(
        ptr(ref_fea_cl), ptr(src_feas_cl), ptr(rts), ptr(depth_values), _depth_mode(depth_values),
        B, V, C, D, H, W, int(align_corners), int(alias_quirk), MVS_LAYOUT_NHWC,
        MVS_LAYOUT_C8 if out_c8 else MVS_LAYOUT_NHWC, ptr(out), stream()),
        "mvs_costvol_variance_fwd_f32")
    return out


def nchw_to_c16(x):
    """[...,C,H,W] -> [...,C/16,H,W,16]: 16-channel blocked feature maps for the
    LDS-staged variance kernel (one HIP transpose launch)."""
    x = _f32c(x)
    *lead, C, H, W = x.shape
    if C % 16:
        raise MvsHipError(f"nchw_to_c16 needs C % 16 == 0, got {C}")
    n = 1
    for v in lead:
        n *= v
    out = torch.empty(tuple(lead) + (C // 16, H, W, 16), device=x.device, dtype=torch.float32)
    check(_lib.load().mvs_nchw_to_nhwc_f32(ptr(x), ptr(out), n * (C // 16), 16, H * W, stream()),
          "mvs_nchw_to_nhwc_f32")
    return out


_variance_ws = {}


def _variance_workspace(dev, nbytes):
    """Scratch of the persistent variance kernel's cold-path queue: one buffer per device and
    stream, grown on demand (kernels on one stream run in order, so they can share it)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _variance_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _variance_ws[key] = torch.empty(max(nbytes, 1 << 20), device=dev, dtype=torch.uint8)
    return ws


def costvol_variance_c16(ref16, srcs16, rts, depth_values, align_corners=False, alias_quirk=False,
                         out_c8=False, fast=False, absmax_out=None):
    """LDS-staged fused warp+variance.  ref16 [B,C/16,H,W,16]; srcs16 [V-1,B,C/16,H,W,16]
    -> [B,D,H,W,C] or (out_c8) [B,D,H,C/8,W,8].  Shared depth planes run the persistent
    kernel (mvs_costvol_variance_fwd_ws_f32); `fast` selects its fast-coordinate mode
    (MVS_SWEEP_FAST), otherwise the result is bit-identical to the reference's arithmetic.
    absmax_out: absmax block (absmax_block()) that receives the volume's largest magnitude
    (mvs_costvol_variance_fwd_ws2_f32; what conv3d_c8_f16x3 / costreg_forward scale conv0's operands by)."""
    ref16, srcs16, depth_values = _f32c(ref16), _f32c(srcs16), _f32c(depth_values)
    B, G, H, W, blk = ref16.shape
    if blk not in (4, 16):
        raise MvsHipError(f"blocked features come 16 or 4 channels to a block, got {blk}")
    layout = MVS_LAYOUT_C16 if blk == 16 else MVS_LAYOUT_C4
    C = G * blk
    V = srcs16.shape[0] + 1
    D = depth_values.shape[1]
    shape = (B, D, H, C // 8, W, 8) if out_c8 else (B, D, H, W, C)
    out = torch.empty(shape, device=ref16.device, dtype=torch.float32)
    lib = _lib.load()
    mode = _depth_mode(depth_values)
    need = lib.mvs_costvol_variance_workspace_bytes2(mode, B, V, C, D, H, W, layout, int(alias_quirk))
    ws = _variance_workspace(ref16.device, need) if need else None
    import os
    # (a forced kernel, MVS_SWEEP_PERSIST, runs no chooser: nothing to read back)
    hint = _SweepVerdict.get(ref16.device, (mode, B, V, C, D, H, W, layout)) if ws is not None and "MVS_SWEEP_PERSIST" not in os.environ else None
    flags = (1 if fast else 0) | (2 if hint is not None and hint.per_tile() else 0)      # MVS_SWEEP_FAST | MVS_SWEEP_TILE_CANDIDATE_PER_TILE
    check(lib.mvs_costvol_variance_fwd_ws2_f32(
        ptr(ref16), ptr(srcs16), ptr(rts), ptr(depth_values), mode, B, V, C,
        D, H, W, int(align_corners), int(alias_quirk), layout,
        MVS_LAYOUT_C8 if out_c8 else MVS_LAYOUT_NHWC, flags, ptr(out),
        ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, ws.numel() if ws is not None else 0,
        ctypes.c_void_p(absmax_out.data_ptr()) if absmax_out is not None else None,
        stream()), "mvs_costvol_variance_fwd_ws2_f32")
    if hint is not None:
        hint.observe(ws)
    return out


def handover_enabled():
    """False when MVS_HANDOVER=0 keeps the variance volume fp32 between the sweep and conv0 (A/B).  Default: the sweep hands it
    over as two fp16 pieces per value (mvs_costvol_variance_fwd_ws3_f32) wherever conv0 runs on the two-piece kernel."""
    import os
    return conv0_f16_enabled() and os.environ.get("MVS_HANDOVER", "1") != "0"


class HandedVolume:
    """What mvs_costvol_variance_fwd_ws3_f32 left on the device: `buf` = the volume -- MVS_LAYOUT_C8PH pieces scaled by the bound
    in `hand` if redo[0] == 0, plain fp32 MVS_LAYOUT_C8 if redo[0] == 1 (nobody on the host needs to know which) --, `absmax` =
    the block of its true largest magnitude, shape = (B, C, D, H, W)."""
    __slots__ = ("buf", "hand", "absmax", "redo", "shape")

    def __init__(self, buf, hand, absmax, redo, shape):
        self.buf, self.hand, self.absmax, self.redo, self.shape = buf, hand, absmax, redo, shape

    @property
    def device(self):
        return self.buf.device

    def to_c8(self):
        """The volume as fp32 [B,D,H,C/8,W,8] (tests, diagnostics: synchronises the device)."""
        B, C, D, H, W = self.shape
        if int(self.redo[0].item()) != 0:
            return self.buf[:B * D * H * W * C * 4].view(torch.float32).view(B, D, H, C // 8, W, 8).clone()
        e = (int(self.hand.max().item()) >> 23 & 255) - 127
        e = max(-100, min(127, e))
        main, _ = self.pieces()
        v = main[:, :, :, 0].double() + main[:, :, :, 1].double()               # hi + lo: [B,D,G,parity,H,W2,8]
        v = v.permute(0, 1, 4, 2, 5, 3, 6).reshape(B, D, H, C // 8, -1, 8)[:, :, :, :, :W]      # x = 2 i + parity
        return (v * 2.0 ** (e - 14)).float().contiguous()

    def pieces(self):
        """Views of the MVS_LAYOUT_C8PH buffer as fp16: main [B,D,C/8,part,parity,H,ceil(W/2),8], halo strips [B,D,C/8,part,tile,side,H,8]."""
        B, C, D, H, W = self.shape
        w2, tiles = (W + 1) // 2, (W + 31) // 32
        region, strip = H * w2 * 16, H * 16
        chunk = (4 * region + 4 * tiles * strip + 255) // 256 * 256
        blk = self.buf[:B * D * (C // 8) * chunk].view(B, D, C // 8, chunk)
        main = blk[..., :4 * region].view(torch.float16).view(B, D, C // 8, 2, 2, H, w2, 8)
        halo = blk[..., 4 * region:4 * region + 4 * tiles * strip].view(torch.float16).view(B, D, C // 8, 2, tiles, 2, H, 8)
        return main, halo


def costvol_variance_handover(ref16, srcs16, rts, depth_values, fea_absmax, align_corners=False, fast=False, veto=None):
    """The fused warp + variance sweep with the volume handed to conv0 as two fp16 pieces per value (include/mvs_hip.h:
    mvs_costvol_variance_fwd_ws3_f32).  ref16 / srcs16 as costvol_variance_c16; fea_absmax: the absmax block of ALL the feature
    maps (the pieces' scale comes from the bound var <= max|f|^2); veto: None, or a device address (int) of one float of the
    reader -- NaN there keeps the volume fp32.  -> HandedVolume, or None when this shape / environment has no hand-over
    (per-pixel hypotheses, a forced kernel, MVS_HANDOVER=0): call costvol_variance_c16 then."""
    import os
    ref16, srcs16, depth_values = _f32c(ref16), _f32c(srcs16), _f32c(depth_values)
    B, G, H, W, blk = ref16.shape
    if blk not in (4, 16) or not handover_enabled() or "MVS_SWEEP_PERSIST" in os.environ or _depth_mode(depth_values) != 0:
        return None
    layout = MVS_LAYOUT_C16 if blk == 16 else MVS_LAYOUT_C4
    C, V, D = G * blk, srcs16.shape[0] + 1, depth_values.shape[1]
    lib = _lib.load()
    need = lib.mvs_costvol_variance_workspace_bytes2(0, B, V, C, D, H, W, layout, 0)
    nbytes = lib.mvs_costvol_variance_handover_bytes(B, C, D, H, W)
    if need == 0 or nbytes == 0:      # (0 bytes: a plane beyond the 32-bit offsets of conv0's copies)
        return None
    ws = _variance_workspace(ref16.device, need)
    buf = torch.empty(nbytes, device=ref16.device, dtype=torch.uint8)
    words = torch.empty(2 * ABSMAX_WORDS + 4, device=ref16.device, dtype=torch.int32)      # hand, absmax, redo: all written by the kernels
    hv = HandedVolume(buf, words[:ABSMAX_WORDS], words[ABSMAX_WORDS:2 * ABSMAX_WORDS], words[2 * ABSMAX_WORDS:], (B, C, D, H, W))
    hint = _SweepVerdict.get(ref16.device, (0, B, V, C, D, H, W, layout))
    flags = (1 if fast else 0) | (2 if hint.per_tile() else 0)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    check(lib.mvs_costvol_variance_fwd_ws3_f32(
        ptr(ref16), ptr(srcs16), ptr(rts), ptr(depth_values), B, V, C, D, H, W, int(align_corners), layout, flags,
        vp(fea_absmax), ctypes.c_void_p(veto) if veto else None, vp(buf), vp(ws), ws.numel(), vp(hv.absmax), vp(hv.hand), vp(hv.redo),
        stream()), "mvs_costvol_variance_fwd_ws3_f32")
    hint.observe(ws)
    return hv


def conv0_veto_word(packed_f16x3, cin):
    """Device address of the word of a conv0 two-piece pack that is NaN when the layer's weights are not finite (what
    costvol_variance_handover takes as `veto`)."""
    return _lib.load().mvs_conv3d_f16x3_pack_veto_word(ctypes.c_void_p(packed_f16x3.data_ptr()), cin)


def conv3d_c8_handed(hv, packed, scale=None, shift=None, relu=False, out_absmax=None):
    """conv0 on whatever the hand-over sweep left (mvs_conv3d_c8_handed_f16x3_f32): the kernel on the pieces, then the fp32 kernel
    under the redo word -- exactly one of the two runs.  -> [B,D,H,W,8]."""
    B, C, D, H, W = hv.shape
    out = torch.empty(B, D, H, W, 8, device=hv.device, dtype=torch.float32)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    with stage("conv3d_split"):
        check(_lib.load().mvs_conv3d_c8_handed_f16x3_f32(
            vp(hv.buf), vp(hv.hand), vp(hv.redo), vp(hv.absmax), ptr(packed), ptr(_f32c(scale)) if scale is not None else None,
            ptr(_f32c(shift)) if shift is not None else None, None, int(bool(relu)), B, C, D, H, W, ptr(out), vp(out_absmax), stream()),
            "mvs_conv3d_c8_handed_f16x3_f32")
    return out


class _SweepVerdict:
    """Which kernel the device-side chooser of the sweep picked for a shape, read back WITHOUT synchronising: now and then the
    verdict word of the workspace is copied to pinned host memory behind the call, and looked at by a later call once its event
    has completed.  Only a speed hint comes of it (how the per-tile candidate is launched: include/mvs_hip.h,
    MVS_SWEEP_TILE_CANDIDATE_PER_TILE); a stale or missing verdict costs microseconds, never correctness.  Skipped while the
    stream is capturing a HIP graph."""
    _table = {}
    PERIOD = 8           # calls between two read-backs once a verdict is known (4 bytes, asynchronous)

    @classmethod
    def get(cls, device, key):
        k = (device.index, key)
        v = cls._table.get(k)
        if v is None:
            v = cls._table[k] = cls()
        return v

    def __init__(self):
        self.choice = None       # 16 / 8 / 0, or None: not known yet
        self.pending = None      # (pinned word, event)
        self.calls = 0

    def per_tile(self):
        # (an event query is not a legal call while a stream captures in the default 'global' mode: keep the last known choice)
        if self.pending is not None and not torch.cuda.is_current_stream_capturing() and self.pending[1].query():
            self.choice = int(self.pending[0].item())
            self.pending = None
        return self.choice == 0

    def observe(self, ws):
        self.calls += 1
        due = self.choice is None or self.calls % self.PERIOD == 0
        if not due or self.pending is not None or torch.cuda.is_current_stream_capturing():
            return
        word = torch.empty(1, dtype=torch.int32, pin_memory=True)
        word.copy_(ws[4:8].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = (word, ev)


def costvol_variance_nhwc_ws(ref_cl, srcs_cl, rts, depth_values, align_corners=False, out_c8=False, fast=False,
                             absmax_out=None):
    """Channels-last maps [B,H,W,C] / [V-1,B,H,W,C] through the workspace entry: the persistent kernel
    reads them in place (no re-blocking copy) when the shape is its own, else the gather kernel runs."""
    ref_cl, srcs_cl, depth_values = _f32c(ref_cl), _f32c(srcs_cl), _f32c(depth_values)
    B, H, W, C = ref_cl.shape
    V = srcs_cl.shape[0] + 1
    D = depth_values.shape[1]
    shape = (B, D, H, C // 8, W, 8) if out_c8 else (B, D, H, W, C)
    out = torch.empty(shape, device=ref_cl.device, dtype=torch.float32)
    lib = _lib.load()
    mode = _depth_mode(depth_values)
    need = lib.mvs_costvol_variance_workspace_bytes(mode, B, V, C, D, H, W, MVS_LAYOUT_NHWC)
    ws = _variance_workspace(ref_cl.device, need) if need else None
    check(lib.mvs_costvol_variance_fwd_ws2_f32(
        ptr(ref_cl), ptr(srcs_cl), ptr(rts), ptr(depth_values), mode, B, V, C, D, H, W, int(align_corners), 0,
        MVS_LAYOUT_NHWC, MVS_LAYOUT_C8 if out_c8 else MVS_LAYOUT_NHWC, 1 if fast else 0, ptr(out),
        ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, ws.numel() if ws is not None else 0,
        ctypes.c_void_p(absmax_out.data_ptr()) if absmax_out is not None else None,
        stream()), "mvs_costvol_variance_fwd_ws2_f32")
    return out


def variance_persistent_supported(depth_values, B, V, C, H, W, alias_quirk=False):
    """True when mvs_costvol_variance_fwd_ws_f32 serves this shape with the persistent kernel
    (then 4-channel-blocked features, nchw_to_c4, are its fastest input).  The same predicate as the launcher's."""
    depth_values = _f32c(depth_values)
    return _lib.load().mvs_costvol_variance_workspace_bytes2(
        _depth_mode(depth_values), B, V, C, depth_values.shape[1], H, W, MVS_LAYOUT_C4, int(alias_quirk)) > 0


def nchw_to_c4(x):
    """[...,C,H,W] -> [...,C/4,H,W,4]: 4-channel blocked feature maps (MVS_LAYOUT_C4)."""
    x = _f32c(x)
    *lead, C, H, W = x.shape
    if C % 4:
        raise MvsHipError(f"nchw_to_c4 needs C % 4 == 0, got {C}")
    n = 1
    for v in lead:
        n *= v
    out = torch.empty(tuple(lead) + (C // 4, H, W, 4), device=x.device, dtype=torch.float32)
    check(_lib.load().mvs_nchw_to_nhwc_f32(ptr(x), ptr(out), n * (C // 4), 4, H * W, stream()),
          "mvs_nchw_to_nhwc_f32")
    return out


def c8_to_nchw(x):
    """[B,D,H,C/8,W,8] -> [B,C,D,H,W] (torch ops; tests / debugging only)."""
    B, D, H, G, W, _ = x.shape
    return x.permute(0, 3, 5, 1, 2, 4).reshape(B, G * 8, D, H, W).contiguous()


def nchw_to_c8(x):
    """[B,C,D,H,W] -> [B,D,H,C/8,W,8] (torch ops; tests / debugging only)."""
    B, C, D, H, W = x.shape
    return x.reshape(B, C // 8, 8, D, H, W).permute(0, 3, 4, 1, 5, 2).contiguous()


# ---------------------------------------------------------------- K3 conv
IMPL_AUTO, IMPL_DIRECT, IMPL_MFMA = 0, 1, 2


def conv3d_mfma_supported(transposed, cin, cout, stride):
    return bool(_lib.load().mvs_conv3d_mfma_supported(int(transposed), cin, cout, stride))


# A packed fp32 weight tensor carries its derived packs as attributes of the tensor OBJECT: `_mvs_split` = (bf16 three-piece
# companion, fp16 two-piece companion), `_mvs_lazy_fill` = the callable that fills its fp32 fragments on demand.  Nothing
# module-level owns them (an id-keyed registry whose entry held a closure over `packed` pinned every training
# step's packs forever -- ~5 MB per eager step); they die with the tensor.
def _register_split(packed, split, f16=None):
    packed._mvs_split = (split, f16)


def _register_lazy(packed, fill):
    """fill(packed) writes the fp32 fragments; it must not close over `packed` itself (a tensor -> closure -> tensor cycle
    would leave the device memory to the cycle collector)."""
    packed._mvs_lazy_fill = fill


def materialize_packed(packed):
    """pack_conv*_weight(..., lazy=True) skips the fp32 fragment pack of a layer whose split-operand companion will run it (the
    training path packs every weight every step: ~35 tiny launches per step that nothing read).  A caller that does fall through to
    the fp32 MFMA kernels -- a volume beyond the split launcher's limits, an explicit impl -- fills the fragments here first."""
    fill = getattr(packed, "_mvs_lazy_fill", None) if packed is not None else None
    if fill is not None:
        packed._mvs_lazy_fill = None
        fill(packed)
    return packed


def split_companion(packed):
    """The bf16 hi/mid/lo pack registered for this packed weight tensor (pack_conv*_weight(..., split=True)), or None."""
    hit = getattr(packed, "_mvs_split", None) if packed is not None else None
    return hit[0] if hit is not None else None


def f16_companion(packed):
    """The scaled fp16 hi/lo pack registered beside it (two-piece form: the layer runs there when its caller hands the
    input's absmax block to conv3d / conv2d), or None."""
    hit = getattr(packed, "_mvs_split", None) if packed is not None else None
    return hit[1] if hit is not None else None


def split_f16_enabled():
    """False when MVS_SPLIT_F16=0 keeps the split-operand layers on the three-piece bf16 kernels (six products; A/B).
    Default: the two-piece fp16 kernels (three products) wherever the caller chains the absmax blocks."""
    import os
    return conv_split_enabled() and os.environ.get("MVS_SPLIT_F16", "1") != "0"


def pack_conv3d_weight(weight, transposed, stride, split=False, f16=True, lazy=False):
    """PyTorch-layout weight -> MFMA A-fragment order (None if the shape has no
    MFMA configuration).  split: also pack the layer for the split-operand bf16 kernel (mvs_conv_split_f32) where
    its shape has one and MVS_CONV_SPLIT is not 0; conv3d() then runs the layer there (inference paths opt in).  f16: with
    split, also the two-piece fp16 pack (taken when the caller hands conv3d the input's absmax block); the training layers,
    whose weights are packed every step and whose inputs carry no block, pass False."""
    weight = _f32c(weight)
    cin, cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
    n = _lib.load().mvs_conv3d_packed_weight_floats(int(transposed), cin, cout, stride)
    if n <= 0:
        return None
    packed = torch.empty(n, device=weight.device, dtype=torch.float32)

    def fill(dst):
        check(_lib.load().mvs_conv3d_pack_weights_f32(ptr(weight), int(transposed), cin, cout, stride,
                                                      ptr(dst), stream()),
              "mvs_conv3d_pack_weights_f32")
    # (the stride-2 layers have a split-operand kernel too; it wins only where one launch covers the layer -- conv1, 8 -> 16:
    # 0.28 vs 0.34 ms -- not where each 16 output channels re-read and re-split the input: conv3 0.17 vs 0.13, conv5
    # 0.15 vs 0.105: those are routed there only on request)
    import os
    want = stride == 1 or (stride == 2 and (cin == 8 or os.environ.get("MVS_CONV_SPLIT_STRIDE2") == "1"))
    if split and not transposed and conv_split_enabled() and want:
        sp = pack_conv_weight_split(weight, stride)
        if sp is not None:
            _register_split(packed, sp, pack_conv_weight_split_f16(weight, stride) if (f16 and split_f16_enabled()) else None)
    elif split and not transposed and stride == 2 and f16 and split_f16_enabled():
        # (round 4) conv3 / conv5: no bf16 companion (it loses to the fp32 kernel), but the TWO-piece form takes 32 output channels
        # per launch and wins: registered alone, taken when the caller hands over the input's absmax block
        pf = pack_conv_weight_split_f16(weight, stride)
        if pf is not None:
            _register_split(packed, None, pf)
    if split and transposed and stride == 2 and conv_split_enabled():
        sp = pack_deconv_weight_split(weight)
        if sp is not None:
            import os
            f16d = f16 and split_f16_enabled() and os.environ.get("MVS_DECONV_F16", "1") != "0"      # (A/B switch)
            _register_split(packed, sp, pack_deconv_weight_split_f16(weight) if f16d else None)
    # lazy: the fp32 fragments only if no split-operand companion will run the layer (materialize_packed() otherwise, on demand)
    if lazy and split_companion(packed) is not None:
        _register_lazy(packed, fill)
    else:
        fill(packed)
    return packed


def conv_split_enabled():
    """False when MVS_CONV_SPLIT=0 keeps conv0 on the fp32 MFMA kernel (A/B switch; default: the
    split-operand bf16 kernel, mvs_conv3d_c8_bf16x6_f32)."""
    import os
    return os.environ.get("MVS_CONV_SPLIT", "1") != "0"


def pack_conv3d_weight_split(weight):
    """(8, Cin, 3, 3, 3) weight -> the bf16 hi/mid/lo A fragments of conv3d_c8_split (None if the
    shape has no such kernel)."""
    weight = _f32c(weight)
    if weight.dim() != 5 or weight.shape[0] != 8 or tuple(weight.shape[2:]) != (3, 3, 3):
        return None
    n = _lib.load().mvs_conv3d_bf16x6_packed_bytes(int(weight.shape[1]))
    if n == 0:
        return None
    packed = torch.empty(n // 4, device=weight.device, dtype=torch.float32)   # opaque bytes
    check(_lib.load().mvs_conv3d_pack_weights_bf16x6_f32(ptr(weight), int(weight.shape[1]), ptr(packed), stream()),
          "mvs_conv3d_pack_weights_bf16x6_f32")
    return packed


def conv3d_c8_split(x_c8, packed_split, scale=None, shift=None, residual=None, relu=False):
    """conv0-class layer (3x3x3, Cout 8, stride 1) on the bf16 matrix pipe with exactly split fp32
    operands (mvs_conv3d_c8_bf16x6_f32): x_c8 [B,D,H,Cin/8,W,8] -> [B,D,H,W,8]."""
    x_c8 = _f32c(x_c8)
    B, D, H, G, W, _ = x_c8.shape
    out = torch.empty(B, D, H, W, 8, device=x_c8.device, dtype=torch.float32)
    with stage("conv3d_split"):
        check(_lib.load().mvs_conv3d_c8_bf16x6_f32(
            ptr(x_c8), ptr(packed_split), ptr(_f32c(scale)) if scale is not None else None,
            ptr(_f32c(shift)) if shift is not None else None,
            ptr(_f32c(residual)) if residual is not None else None, int(bool(relu)), B, G * 8, D, H, W,
            ptr(out), stream()), "mvs_conv3d_c8_bf16x6_f32")
    return out


def pack_conv3d_weight_f16x3(weight):
    """(8, Cin, 3, 3, 3) weight -> the scaled fp16 hi/lo A fragments of conv3d_c8_f16x3 + the trailer that undoes the
    scale (None if the shape has no such kernel)."""
    weight = _f32c(weight)
    if weight.dim() != 5 or weight.shape[0] != 8 or tuple(weight.shape[2:]) != (3, 3, 3):
        return None
    n = _lib.load().mvs_conv3d_f16x3_packed_bytes(int(weight.shape[1]))
    if n == 0:
        return None
    packed = torch.empty(n // 4, device=weight.device, dtype=torch.float32)   # opaque bytes
    check(_lib.load().mvs_conv3d_pack_weights_f16x3_f32(ptr(weight), int(weight.shape[1]), ptr(packed), stream()),
          "mvs_conv3d_pack_weights_f16x3_f32")
    return packed


ABSMAX_WORDS = 256      # MVS_ABSMAX_WORDS of include/mvs_hip.h


def absmax_block(device, zero=False, n=None):
    """An absmax block: the int32 words a producer kernel collects the largest magnitude of its output in (bit patterns of
    |x|; the maximum over the block is the value) and a two-piece fp16 convolution scales its input by.  n: that many
    blocks as rows of one tensor (one fill for a whole network's blocks)."""
    shape = ABSMAX_WORDS if n is None else (n, ABSMAX_WORDS)
    return (torch.zeros if zero else torch.empty)(shape, device=device, dtype=torch.int32)


def guard_fallback_count():
    """Launches of the current device whose two-piece fp16 layer judged its input unfit for one scale per tensor (non-finite,
    or outlier-dominated) and ran in plain fp32 instead (mvs_guard_fallback_count; synchronises the device: a diagnostic)."""
    n = ctypes.c_ulonglong(0)
    check(_lib.load().mvs_guard_fallback_count(ctypes.byref(n)), "mvs_guard_fallback_count")
    return int(n.value)


def absmax_value(block):
    """The float an absmax block holds."""
    return block.max().view(1).view(torch.float32).item()


def _with_absmax(out, block):
    if block is not None:
        absmax(out, block)
    return out


def absmax(x, out=None):
    """Largest magnitude of a device array into an absmax block (mvs_absmax_f32: resets the block, then one pass over x)."""
    x = _f32c(x)
    if out is None:
        out = absmax_block(x.device)
    with stage("absmax"):
        check(_lib.load().mvs_absmax_f32(ptr(x), x.numel(), ctypes.c_void_p(out.data_ptr()), stream()), "mvs_absmax_f32")
    return out


def conv3d_c8_f16x3(x_c8, packed, x_absmax=None, scale=None, shift=None, residual=None, relu=False, out_absmax=None):
    """conv0-class layer (3x3x3, Cout 8, stride 1) on the fp16 matrix pipe with two-piece operands, three products
    (mvs_conv3d_c8_f16x3_f32): x_c8 [B,D,H,Cin/8,W,8] -> [B,D,H,W,8].  x_absmax: the absmax block (absmax_block()) the producer of
    x_c8 filled (None: computed here by one more pass over x_c8)."""
    x_c8 = _f32c(x_c8)
    B, D, H, G, W, _ = x_c8.shape
    if x_absmax is None:
        x_absmax = absmax(x_c8)
    out = torch.empty(B, D, H, W, 8, device=x_c8.device, dtype=torch.float32)
    with stage("conv3d_split"):
        check(_lib.load().mvs_conv3d_c8_f16x3_f32(
            ptr(x_c8), ctypes.c_void_p(x_absmax.data_ptr()), ptr(packed), ptr(_f32c(scale)) if scale is not None else None,
            ptr(_f32c(shift)) if shift is not None else None,
            ptr(_f32c(residual)) if residual is not None else None, int(bool(relu)), B, G * 8, D, H, W,
            ptr(out), ctypes.c_void_p(out_absmax.data_ptr()) if out_absmax is not None else None, stream()),
            "mvs_conv3d_c8_f16x3_f32")
    return out


def c8_to_c8h(x_c8, block):
    """fp32 blocked volume [B,D,H,C/8,W,8] -> the fp16 pairs of MVS_LAYOUT_C8H (opaque int16 tensor) under the scale of an
    absmax block that bounds it (mvs_c8_to_c8h_f32)."""
    x_c8 = _f32c(x_c8)
    B, D, H, G, W, _ = x_c8.shape
    out = torch.empty(_lib.load().mvs_c8h_bytes(B, G * 8, D, H, W) // 2, device=x_c8.device, dtype=torch.int16)
    check(_lib.load().mvs_c8_to_c8h_f32(ptr(x_c8), ctypes.c_void_p(block.data_ptr()), B, G * 8, D, H, W,
                                        ctypes.c_void_p(out.data_ptr()), stream()), "mvs_c8_to_c8h_f32")
    return out


def conv3d_c8h_f16x3(x_pairs, shape, packed, x_absmax, scale=None, shift=None, residual=None, relu=False, out_absmax=None):
    """conv0-class layer on a volume that arrives as fp16 pairs (MVS_LAYOUT_C8H; mvs_conv3d_c8h_f16x3_f32).  shape = (B, Cin, D, H, W);
    x_absmax = the block the producer scaled by."""
    B, cin, D, H, W = shape
    out = torch.empty(B, D, H, W, 8, device=x_pairs.device, dtype=torch.float32)
    with stage("conv3d_split"):
        check(_lib.load().mvs_conv3d_c8h_f16x3_f32(
            ctypes.c_void_p(x_pairs.data_ptr()), ctypes.c_void_p(x_absmax.data_ptr()), ptr(packed),
            ptr(_f32c(scale)) if scale is not None else None, ptr(_f32c(shift)) if shift is not None else None,
            ptr(_f32c(residual)) if residual is not None else None, int(bool(relu)), B, cin, D, H, W,
            ptr(out), ctypes.c_void_p(out_absmax.data_ptr()) if out_absmax is not None else None, stream()),
            "mvs_conv3d_c8h_f16x3_f32")
    return out


_pack_batch_tls = threading.local()


def _pack_batch_state():
    st = getattr(_pack_batch_tls, "st", None)
    if st is None:
        st = _pack_batch_tls.st = {"open": False, "keep": []}
    return st


@contextlib.contextmanager
def pack_batch():
    """Within the block every bf16 split pack (pack_conv_weight_split) is recorded and all of them run as ONE launch at the
    end (mvs_pack_batch_begin / _end): the training step's prepare pass.  Source weights -- temporaries such as the flipped
    weights of an input-gradient layer included -- are kept alive until then.  Not re-entrant."""
    st = _pack_batch_state()
    if st["open"]:
        yield
        return
    check(_lib.load().mvs_pack_batch_begin(), "mvs_pack_batch_begin")
    st["open"] = True
    try:
        yield
    finally:
        st["open"] = False
        rc = _lib.load().mvs_pack_batch_end(stream())
        st["keep"].clear()
        check(rc, "mvs_pack_batch_end")


def pack_conv_weight_split(weight, stride=1):
    """(Cout, Cin, [3,] 3, 3) weight -> the bf16 hi/mid/lo A fragments of conv_split (None if the shape has no
    such kernel: stride 1 with Cin, Cout in {16, 32, 64}; 3D stride 2 with Cin in {8, 16, 32}; 2D stride 2 = the
    (16, 8, 5, 5) and (32, 16, 5, 5) layers of FeatureNet)."""
    weight = _f32c(weight)
    kd = 3 if weight.dim() == 5 else 1
    k55 = kd == 1 and stride == 2           # FeatureNet's 5x5 stride-2 layers
    if tuple(weight.shape[-2:]) != ((5, 5) if k55 else (3, 3)) or (kd == 3 and weight.shape[2] != 3):
        return None
    n = _lib.load().mvs_conv_split_packed_bytes(kd, int(weight.shape[1]), int(weight.shape[0]), stride)
    if n == 0:
        return None
    packed = torch.empty(n // 4, device=weight.device, dtype=torch.float32)   # opaque bytes
    st = _pack_batch_state()
    if st["open"]:
        st["keep"].append((weight, packed))
    check(_lib.load().mvs_conv_split_pack_weights_f32(ptr(weight), kd, int(weight.shape[1]), int(weight.shape[0]), stride,
                                                      ptr(packed), stream()), "mvs_conv_split_pack_weights_f32")
    return packed


def pack_conv_weight_split_f16(weight, stride=1):
    """The same layers' weights as scaled fp16 hi/lo A fragments + trailer (two-piece form, mvs_conv_split_f16_f32)."""
    weight = _f32c(weight)
    kd = 3 if weight.dim() == 5 else 1
    k55 = kd == 1 and stride == 2
    if tuple(weight.shape[-2:]) != ((5, 5) if k55 else (3, 3)) or (kd == 3 and weight.shape[2] != 3):
        return None
    n = _lib.load().mvs_conv_split_f16_packed_bytes(kd, int(weight.shape[1]), int(weight.shape[0]), stride)
    if n == 0:
        return None
    packed = torch.empty(n // 4, device=weight.device, dtype=torch.float32)   # opaque bytes
    check(_lib.load().mvs_conv_split_pack_weights_f16_f32(ptr(weight), kd, int(weight.shape[1]), int(weight.shape[0]), stride,
                                                          ptr(packed), stream()), "mvs_conv_split_pack_weights_f16_f32")
    return packed


def conv_split_f16(x_cl, packed_f16, cout, x_absmax=None, scale=None, shift=None, residual=None, relu=1, kd=3, out_c4=False,
                   stride=1, out_absmax=None, soft=False):
    """conv_split on the fp16 matrix pipe with two-piece operands (three products; mvs_conv_split_f16_f32).  x_absmax: the
    absmax block the producer of x_cl filled (None: one more pass over x_cl); out_absmax: ZEROED absmax block that receives
    the largest magnitude of the result (for the next two-piece layer)."""
    x_cl = _f32c(x_cl)
    if x_absmax is None:
        x_absmax = absmax(x_cl)
    if kd == 3:
        B, D, H, W, cin = x_cl.shape
        out = torch.empty(B, (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1, cout, device=x_cl.device,
                          dtype=torch.float32)
    else:
        D, H, W, cin = x_cl.shape
        B = 1
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        out = torch.empty((D, cout // 4, Ho, Wo, 4) if out_c4 else (D, Ho, Wo, cout), device=x_cl.device, dtype=torch.float32)
    with stage("conv_split"):
        rc = _lib.load().mvs_conv_split_f16_f32(
            ptr(x_cl), ctypes.c_void_p(x_absmax.data_ptr()), ptr(packed_f16), ptr(_f32c(scale)) if scale is not None else None,
            ptr(_f32c(shift)) if shift is not None else None,
            ptr(_f32c(residual)) if residual is not None else None, int(relu), kd, stride, B, cin, cout, D, H, W,
            int(bool(out_c4)), ptr(out), ctypes.c_void_p(out_absmax.data_ptr()) if out_absmax is not None else None, stream())
    if soft and rc == MVS_EUNSUPPORTED:
        return None
    check(rc, "mvs_conv_split_f16_f32")
    return out


MVS_EUNSUPPORTED = -2     # include/mvs_hip.h
split_stage_names = set()  # stage names whose kernel ran on the split-operand bf16 pipe (bench.py prices them against that ceiling)


def conv_split(x_cl, packed_split, cout, scale=None, shift=None, residual=None, relu=1, kd=3, out_c4=False, stride=1, soft=False):
    """3x3(x3) layer (kd = 1 with stride 2: 5x5) on the bf16 matrix pipe with exactly split fp32 operands
    (mvs_conv_split_f32).  kd = 3: x_cl [B,D,H,W,Cin] -> [B,D,H,W,cout]; kd = 1: images x_cl [N,H,W,Cin] -> [N,H,W,cout].
    relu: 0 none, 1 ReLU, 2 LeakyReLU(0.1).  soft: return None instead of raising when the launcher answers
    MVS_EUNSUPPORTED (a volume beyond its 32-bit halo offsets / tile count) -- conv3d() / conv2d() then run the layer on the
    fp32 MFMA kernels, whose limits are 4x wider."""
    x_cl = _f32c(x_cl)
    if kd == 3:
        B, D, H, W, cin = x_cl.shape
        out = torch.empty(B, (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1, cout, device=x_cl.device,
                          dtype=torch.float32)
    else:
        D, H, W, cin = x_cl.shape
        B = 1
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        out = torch.empty((D, cout // 4, Ho, Wo, 4) if out_c4 else (D, Ho, Wo, cout), device=x_cl.device, dtype=torch.float32)
    with stage("conv_split"):
        rc = _lib.load().mvs_conv_split_f32(
            ptr(x_cl), ptr(packed_split), ptr(_f32c(scale)) if scale is not None else None,
            ptr(_f32c(shift)) if shift is not None else None,
            ptr(_f32c(residual)) if residual is not None else None, int(relu), kd, stride, B, cin, cout, D, H, W,
            int(bool(out_c4)), ptr(out), stream())
    if soft and rc == MVS_EUNSUPPORTED:
        return None
    check(rc, "mvs_conv_split_f32")
    return out


def pack_deconv_weight_split(weight):
    """(Cin, Cout, 3, 3, 3) transposed-layer weight -> the bf16 hi/mid/lo A fragments of deconv_split (None if the
    shape has no such kernel: Cin in {16, 32, 64}, Cout in {8, 16, 32})."""
    weight = _f32c(weight)
    if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3):
        return None
    n = _lib.load().mvs_deconv_split_packed_bytes(int(weight.shape[0]), int(weight.shape[1]))
    if n == 0:
        return None
    packed = torch.empty(n // 4, device=weight.device, dtype=torch.float32)   # opaque bytes
    check(_lib.load().mvs_deconv_split_pack_weights_f32(ptr(weight), int(weight.shape[0]), int(weight.shape[1]),
                                                        ptr(packed), stream()), "mvs_deconv_split_pack_weights_f32")
    return packed


def pack_deconv_weight_split_f16(weight):
    """The same layers' weights as scaled fp16 hi/lo A fragments + trailer (two-piece form, mvs_deconv_split_f16_f32)."""
    weight = _f32c(weight)
    if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3):
        return None
    n = _lib.load().mvs_deconv_split_f16_packed_bytes(int(weight.shape[0]), int(weight.shape[1]))
    if n == 0:
        return None
    packed = torch.empty(n // 4, device=weight.device, dtype=torch.float32)   # opaque bytes
    check(_lib.load().mvs_deconv_split_pack_weights_f16_f32(ptr(weight), int(weight.shape[0]), int(weight.shape[1]),
                                                            ptr(packed), stream()), "mvs_deconv_split_pack_weights_f16_f32")
    return packed


def deconv_split_f16(x_cl, packed_f16, cout, x_absmax=None, scale=None, shift=None, residual=None, relu=True, out_absmax=None,
                     soft=False):
    """deconv_split on the fp16 matrix pipe with two-piece operands (mvs_deconv_split_f16_f32); x_absmax / out_absmax as
    conv_split_f16."""
    x_cl = _f32c(x_cl)
    if x_absmax is None:
        x_absmax = absmax(x_cl)
    B, D, H, W, cin = x_cl.shape
    out = torch.empty(B, 2 * D, 2 * H, 2 * W, cout, device=x_cl.device, dtype=torch.float32)
    with stage("deconv_split"):
        rc = _lib.load().mvs_deconv_split_f16_f32(
            ptr(x_cl), ctypes.c_void_p(x_absmax.data_ptr()), ptr(packed_f16), ptr(_f32c(scale)) if scale is not None else None,
            ptr(_f32c(shift)) if shift is not None else None,
            ptr(_f32c(residual)) if residual is not None else None, int(bool(relu)), B, cin, cout, D, H, W,
            ptr(out), ctypes.c_void_p(out_absmax.data_ptr()) if out_absmax is not None else None, stream())
    if soft and rc == MVS_EUNSUPPORTED:
        return None
    check(rc, "mvs_deconv_split_f16_f32")
    return out


def deconv_split(x_cl, packed_split, cout, scale=None, shift=None, residual=None, relu=True, soft=False):
    """Transposed 3x3x3 stride-2 layer on the bf16 matrix pipe with exactly split fp32 operands
    (mvs_deconv_split_f32): x_cl [B,D,H,W,Cin] -> [B,2D,2H,2W,cout]; residual is added after the ReLU.
    soft: None instead of an exception on MVS_EUNSUPPORTED (see conv_split)."""
    x_cl = _f32c(x_cl)
    B, D, H, W, cin = x_cl.shape
    out = torch.empty(B, 2 * D, 2 * H, 2 * W, cout, device=x_cl.device, dtype=torch.float32)
    with stage("deconv_split"):
        rc = _lib.load().mvs_deconv_split_f32(
            ptr(x_cl), ptr(packed_split), ptr(_f32c(scale)) if scale is not None else None,
            ptr(_f32c(shift)) if shift is not None else None,
            ptr(_f32c(residual)) if residual is not None else None, int(bool(relu)), B, cin, cout, D, H, W,
            ptr(out), stream())
    if soft and rc == MVS_EUNSUPPORTED:
        return None
    check(rc, "mvs_deconv_split_f32")
    return out


def conv3d(x, weight, scale=None, shift=None, residual=None, relu=False, transposed=False,
           stride=1, channels_last=False, packed=None, impl=IMPL_AUTO, in_c8=False, x_absmax=None, out_absmax=None):
    """3x3x3 (transposed) convolution + per-channel affine + ReLU + skip add.
    x: [B,Cin,D,H,W] or, with channels_last, [B,D,H,W,Cin]; with in_c8 the input
    is the 8-channel-blocked [B,D,H,Cin/8,W,8] (MFMA conv path only; the output
    is channels-last).
    x_absmax: the absmax block of x (filled by the layer that produced x) -- a layer with a two-piece fp16 pack then runs on
    those kernels (three products instead of six).  out_absmax: a ZEROED absmax block that receives the largest magnitude of
    the result for the next layer (collected in the epilogue of the split-operand kernels, by one more pass behind the
    others)."""
    x = _f32c(x)
    weight = _f32c(weight) if weight is not None else None
    if in_c8:
        B, D, H, G, W, _ = x.shape
        cin = G * 8
        channels_last = True
    elif channels_last:
        B, D, H, W, cin = x.shape
    else:
        B, cin, D, H, W = x.shape
    if weight is not None:
        cout = weight.shape[1] if transposed else weight.shape[0]
    else:
        raise MvsHipError("conv3d needs the PyTorch-layout weight (used for shape and the direct path)")
    if transposed:
        Do, Ho, Wo = D * stride, H * stride, W * stride
    else:
        Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    shape = (B, Do, Ho, Wo, cout) if channels_last else (B, cout, Do, Ho, Wo)
    if residual is not None:
        residual = _f32c(residual)
        if tuple(residual.shape) != shape:
            raise MvsHipError(f"residual shape {tuple(residual.shape)} != output shape {shape}")
    # the layer's split-operand pack was registered with its fp32 pack: bf16 matrix pipe, fp32 accuracy.  Only on
    # IMPL_AUTO (an explicit IMPL_MFMA / IMPL_DIRECT measures the kernel it names), and a volume beyond the split
    # launcher's limits falls through to the fp32 kernels below.
    sp = split_companion(packed) if impl == IMPL_AUTO else None
    f16 = f16_companion(packed) if (impl == IMPL_AUTO and x_absmax is not None) else None
    if (sp is not None or f16 is not None) and channels_last and not in_c8 and not transposed:
        if f16 is not None:
            out = conv_split_f16(x, f16, cout, x_absmax, scale, shift, residual, 1 if relu else 0, kd=3, stride=stride,
                                 out_absmax=out_absmax, soft=True)
            if out is not None:
                return out
        out = conv_split(x, sp, cout, scale, shift, residual, 1 if relu else 0, kd=3, stride=stride, soft=True) if sp is not None else None
        if out is not None:
            return _with_absmax(out, out_absmax)
    if sp is not None and channels_last and not in_c8 and transposed and stride == 2:
        if f16 is not None:
            out = deconv_split_f16(x, f16, cout, x_absmax, scale, shift, residual, relu, out_absmax=out_absmax, soft=True)
            if out is not None:
                return out
        out = deconv_split(x, sp, cout, scale, shift, residual, relu, soft=True)
        if out is not None:
            return _with_absmax(out, out_absmax)
    out = torch.empty(shape, device=x.device, dtype=torch.float32)
    materialize_packed(packed)
    check(_lib.load().mvs_conv3d_absmax_f32(
        ptr(x), ptr(weight), ptr(packed), ptr(_f32c(scale)) if scale is not None else None,
        ptr(_f32c(shift)) if shift is not None else None, ptr(residual), int(relu), int(transposed),
        B, cin, cout, D, H, W, stride,
        MVS_LAYOUT_C8 if in_c8 else (MVS_LAYOUT_NHWC if channels_last else MVS_LAYOUT_NCHW),
        impl, ptr(out), ctypes.c_void_p(out_absmax.data_ptr()) if out_absmax is not None else None, stream()),
        "mvs_conv3d_absmax_f32")
    return out


# ------------------------------------------------------------ input pipeline (device side)
def images_u8_to_planar(u8, H=None, W=None):
    """Decoded images [N,Hs,Ws,3] uint8 on the device -> [N,3,H,W] float32 = np.array(img, float32) / 255.
    cropped to the top-left H x W and transposed (dtu_yao_eval.py:60-67,102), bit-identical to the
    reference loader's tensors."""
    if not u8.is_cuda or u8.dtype != torch.uint8 or not u8.is_contiguous() or u8.dim() != 4 or u8.shape[3] != 3:
        raise MvsHipError("images_u8_to_planar needs a contiguous device uint8 tensor [N,Hs,Ws,3]")
    if u8.device.index != torch.cuda.current_device():
        raise MvsHipError("images_u8_to_planar: tensor is not on the current device")
    N, Hs, Ws, _ = u8.shape
    H, W = (Hs if H is None else H), (Ws if W is None else W)
    out = torch.empty((N, 3, H, W), device=u8.device, dtype=torch.float32)
    check(_lib.load().mvs_images_u8_to_planar_f32(ctypes.c_void_p(u8.data_ptr()), N, Hs, Ws, H, W, ptr(out), stream()),
          "mvs_images_u8_to_planar_f32")
    return out


def proj_matrices(K, E, intrinsics_div=4.0):
    """K [N,3,3], E [N,4,4] (device, as parsed from the cam files) -> [N,4,4] with top 3x4 =
    (K, rows 0-1 / intrinsics_div) @ E[:3,:4] (dtu_yao_eval.py:54,93-95), bit-identical to the loader's."""
    K, E = _f32c(K), _f32c(E)
    out = torch.empty_like(E)
    check(_lib.load().mvs_proj_matrices_f32(ptr(K), ptr(E), float(intrinsics_div), K.shape[0], ptr(out), stream()),
          "mvs_proj_matrices_f32")
    return out


# ------------------------------------------------------------ CVP-MVSNet glue (device side)
def downsample_bilinear_half(x):
    """F.interpolate(x, scale_factor=0.5, mode='bilinear') for [..., H, W] (net.py:45), bit-identical
    to ATen's CPU kernel."""
    x = _f32c(x)
    *lead, H, W = x.shape
    planes = 1
    for v in lead:
        planes *= v
    out = torch.empty(tuple(lead) + (H // 2, W // 2), device=x.device, dtype=torch.float32)
    check(_lib.load().mvs_downsample_bilinear_half_f32(ptr(x), planes, H, W, ptr(out), stream()),
          "mvs_downsample_bilinear_half_f32")
    return out


def upsample_bicubic2x(x):
    """F.interpolate(x, scale_factor=2, mode='bicubic') for [..., H, W] (net.py:171)."""
    x = _f32c(x)
    *lead, H, W = x.shape
    planes = 1
    for v in lead:
        planes *= v
    out = torch.empty(tuple(lead) + (2 * H, 2 * W), device=x.device, dtype=torch.float32)
    check(_lib.load().mvs_upsample_bicubic2x_f32(ptr(x), planes, H, W, ptr(out), stream()),
          "mvs_upsample_bicubic2x_f32")
    return out


def cvp_refine_hypotheses(depth_up, K_ref, K_src0, E_ref, E_src0, d=4, pixel_interval=1.0):
    """calDepthHypo in test mode (modules.py:147-219) entirely on the device: per batch item the fp64
    camera algebra (mvs_cvp_hypothesis_mats_f64), the per-pixel fp64 epipolar step and its mean
    (mvs_cvp_interval_sum_f64) and depth_up + k * interval (mvs_cvp_hypotheses_f32).
    depth_up [B,H,W] -> [B,2d,H,W]."""
    depth_up = _f32c(depth_up)
    B, H, W = depth_up.shape
    dev = depth_up.device
    lib = _lib.load()
    out = torch.empty((B, 2 * d, H, W), device=dev, dtype=torch.float32)
    mats = torch.empty((B, 59), device=dev, dtype=torch.float64)
    total = torch.empty((B,), device=dev, dtype=torch.float64)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    Kr, Ks, Er, Es = (_f32c(t) for t in (K_ref, K_src0, E_ref, E_src0))
    for b in range(B):
        check(lib.mvs_cvp_hypothesis_mats_f64(ptr(Kr[b]), ptr(Ks[b]), ptr(Er[b]), ptr(Es[b]), vp(mats[b]), stream()),
              "mvs_cvp_hypothesis_mats_f64")
        check(lib.mvs_cvp_interval_sum_f64(ptr(depth_up[b]), vp(mats[b]), H, W, float(pixel_interval), vp(total[b:]),
                                           stream()), "mvs_cvp_interval_sum_f64")
        check(lib.mvs_cvp_hypotheses_f32(ptr(depth_up[b]), vp(total[b:]), H, W, d, ptr(out[b]), stream()),
              "mvs_cvp_hypotheses_f32")
    return out


COSTREG_ORDER = ("conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11", "prob")
_costreg_ws = {}


def conv0_f16_enabled():
    """False when MVS_CONV0_F16=0 keeps conv0 on the three-piece bf16 kernel (six products; A/B switch).  Default: the
    two-piece fp16 kernel (three products, mvs_conv3d_c8_f16x3_f32)."""
    import os
    return conv_split_enabled() and os.environ.get("MVS_CONV0_F16", "1") != "0"


def tail_fused_enabled():
    """False when MVS_TAIL_FUSED=0 keeps conv11 and prob as two launches (A/B switch).  Default: the fused kernel
    (mvs_costreg_tail_f16_f32) inside mvs_costreg_fwd3_f32."""
    import os
    return split_f16_enabled() and os.environ.get("MVS_TAIL_FUSED", "1") != "0"


def pack_costreg_tail(conv11_weight):
    """conv11's (16, 8, 3, 3, 3) ConvTranspose3d weight -> the fragments of the fused conv11 + prob kernel."""
    w = _f32c(conv11_weight)
    lib = _lib.load()
    out = torch.empty(lib.mvs_costreg_tail_packed_bytes(), device=w.device, dtype=torch.uint8)
    check(lib.mvs_costreg_tail_pack_weights_f32(ptr(w), ctypes.c_void_p(out.data_ptr()), stream()), "mvs_costreg_tail_pack_weights_f32")
    return out


def costreg_tail(x, x_absmax, skip, skip_absmax, packed_tail, scale, shift, prob_weight, prob_scale, prob_shift):
    """conv11 + prob as one kernel: x [B,Di,Hi,Wi,16], skip [B,2Di,2Hi,2Wi,8] -> (cost [B,2Di,2Hi,2Wi], flag); flag[0] = 1 when
    the launch declined (range guard): the cost is then NOT written and the caller runs the unfused layers."""
    x, skip = _f32c(x), _f32c(skip)
    B, Di, Hi, Wi, _ = x.shape
    out = torch.empty((B, 2 * Di, 2 * Hi, 2 * Wi), device=x.device, dtype=torch.float32)
    flag = torch.zeros(1, device=x.device, dtype=torch.int32)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    check(_lib.load().mvs_costreg_tail_f16_f32(ptr(x), vp(x_absmax), ptr(skip), vp(skip_absmax), vp(packed_tail),
                                               ptr(scale) if scale is not None else None, ptr(shift) if shift is not None else None,
                                               ptr(_f32c(prob_weight)), ptr(prob_scale) if prob_scale is not None else None,
                                               ptr(prob_shift) if prob_shift is not None else None, B, Di, Hi, Wi, ptr(out), vp(flag),
                                               stream()), "mvs_costreg_tail_f16_f32")
    return out, flag


def costreg_tail_guarded(x, x_absmax, skip, skip_absmax, p11, pprob, flag=None, d11=None):
    """conv11 + prob: the fused kernel with the two unfused layers enqueued behind it under its flag (no host synchronisation):
    whatever the range guard decides, the cost comes back.  p11 / pprob: the layers' parameter dicts (weight, packed, scale,
    shift; p11 with its two-piece companion and 'packed_tail')."""
    x, skip = _f32c(x), _f32c(skip)
    B, Di, Hi, Wi, _ = x.shape
    out = torch.empty((B, 2 * Di, 2 * Hi, 2 * Wi), device=x.device, dtype=torch.float32)
    if d11 is None:         # conv11's output for the unfused path: touched only if the guard declines
        d11 = torch.empty((B, 2 * Di, 2 * Hi, 2 * Wi, 8), device=x.device, dtype=torch.float32)
    if flag is None:        # (or a zeroed word of the caller's: one fill for a whole network's blocks and flags)
        flag = torch.zeros(1, device=x.device, dtype=torch.int32)
    layers = (_lib.ConvLayer * 2)()
    keep = []
    for i, p in enumerate((p11, pprob)):
        for field in ("weight", "packed", "scale", "shift"):
            t = p.get(field)
            if t is not None:
                t = _f32c(t)
                keep.append(t)
                setattr(layers[i], field, t.data_ptr())
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    check(_lib.load().mvs_costreg_tail_guarded_f16_f32(
        ptr(x), vp(x_absmax), ptr(skip), vp(skip_absmax), vp(p11["packed_tail"]), ctypes.cast(ctypes.byref(layers[0]), ctypes.c_void_p),
        vp(f16_companion(p11["packed"])), ctypes.cast(ctypes.byref(layers[1]), ctypes.c_void_p), B, Di, Hi, Wi, ptr(d11), ptr(out),
        vp(flag), stream()), "mvs_costreg_tail_guarded_f16_f32")
    return out


def costreg_forward(x, params, in_c8=False, impl=IMPL_AUTO, x_absmax=None):
    """The whole 3D U-Net in one C call (mvs_costreg_fwd_f32; mvsnet.py:83-93).  x: variance
    volume [B,D,H,W,Cin] or (in_c8) [B,D,H,Cin/8,W,8]; params: name -> dict(weight, packed, scale,
    shift) for the eleven layers of COSTREG_ORDER.  -> cost [B,D,H,W].  The activations live in a
    per-(device, stream) workspace that the next call on the same stream reuses."""
    handed = isinstance(x, HandedVolume)
    if handed:
        hv, x = x, x.buf
        B, cin, D, H, W = hv.shape
        in_c8 = True
    else:
        x = _f32c(x)
        if in_c8:
            B, D, H, G, W, _ = x.shape
            cin = G * 8
        else:
            B, D, H, W, cin = x.shape
    base = params["conv0"]["weight"].shape[0]
    lib = _lib.load()
    need = lib.mvs_costreg_workspace_bytes(B, base, D, H, W)
    if need == 0:
        raise MvsHipError(f"mvs_costreg_fwd_f32 needs D, H, W multiples of 8, got {D}, {H}, {W}")
    key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _costreg_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _costreg_ws[key] = torch.empty(need, device=x.device, dtype=torch.uint8)
    layers = (_lib.ConvLayer * 11)()
    keep = []
    for i, name in enumerate(COSTREG_ORDER):
        p = params[name]
        for field in ("weight", "packed", "scale", "shift", "packed_split"):
            t = p.get(field)
            if field == "packed_split" and t is None:
                t = split_companion(p.get("packed"))
            if t is not None:
                t = _f32c(t)
                keep.append(t)
                if not t.is_cuda:
                    raise MvsHipError("costreg_forward needs device tensors")
                setattr(layers[i], field, t.data_ptr())
    out = torch.empty((B, D, H, W), device=x.device, dtype=torch.float32)
    # two-piece fp16 packs (conv0: packed_f16x3, taken for a blocked input; the other layers: the companion registered with
    # their fp32 pack); x_absmax: the block the variance op filled, else collected inside the call
    f16 = (ctypes.c_void_p * 11)()
    any_f16 = False
    for i, name in enumerate(COSTREG_ORDER):
        t = (params[name].get("packed_f16x3") if in_c8 else None) if i == 0 else f16_companion(params[name].get("packed"))
        # (each switch on its own: MVS_CONV0_F16 decides conv0, MVS_SPLIT_F16 the other layers -- as the per-layer path does)
        if t is not None and (conv0_f16_enabled() if i == 0 else split_f16_enabled()):
            keep.append(t)
            f16[i] = t.data_ptr()
            any_f16 = True
    if any_f16:
        # conv11 -> prob as one kernel (tail_fused.hip) where both run two-piece on a base-8 net; its pack is made once per
        # params dict (the dict is rebuilt with the weights)
        tail = None
        if tail_fused_enabled() and base == 8 and f16[9] and params["conv11"]["weight"].shape[:2] == (16, 8):
            tail = params["conv11"].get("packed_tail")
            if tail is None:
                tail = params["conv11"]["packed_tail"] = pack_costreg_tail(params["conv11"]["weight"])
            keep.append(tail)
        if handed:
            if not f16[0]:
                raise MvsHipError("costreg_forward: a handed-over volume needs conv0's two-piece pack (packed_f16x3) and MVS_CONV0_F16 != 0")
            vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
            check(lib.mvs_costreg_fwd4_f32(vp(hv.buf), vp(hv.hand), vp(hv.redo), vp(hv.absmax), ctypes.cast(layers, ctypes.c_void_p),
                                           ctypes.cast(f16, ctypes.c_void_p), vp(tail) if tail is not None else None,
                                           B, cin, base, D, H, W, impl, vp(ws), ws.numel(), ptr(out), stream()), "mvs_costreg_fwd4_f32")
            return out
        check(lib.mvs_costreg_fwd3_f32(ptr(x), MVS_LAYOUT_C8 if in_c8 else MVS_LAYOUT_NHWC,
                                       ctypes.cast(layers, ctypes.c_void_p), ctypes.cast(f16, ctypes.c_void_p),
                                       ctypes.c_void_p(tail.data_ptr()) if tail is not None else None,
                                       B, cin, base, D, H, W, impl, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                       ctypes.c_void_p(x_absmax.data_ptr()) if x_absmax is not None else None,
                                       ptr(out), stream()), "mvs_costreg_fwd3_f32")
    elif handed:
        raise MvsHipError("costreg_forward: a handed-over volume needs the two-piece packs")
    else:
        check(lib.mvs_costreg_fwd_f32(ptr(x), MVS_LAYOUT_C8 if in_c8 else MVS_LAYOUT_NHWC,
                                      ctypes.cast(layers, ctypes.c_void_p), B, cin, base, D, H, W, impl,
                                      ctypes.c_void_p(ws.data_ptr()), ws.numel(), ptr(out), stream()), "mvs_costreg_fwd_f32")
    return out


# ------------------------------------------------------------ FeatureNet convs
def conv2d_persistent_enabled():
    """False when MVS_CONV2D_PERSISTENT=0 selects the per-tile 2D kernels (which write channels-last only)."""
    import os
    return os.environ.get("MVS_CONV2D_PERSISTENT", "1") != "0"


def conv2d_supported(cin, cout, ksize, stride):
    return bool(_lib.load().mvs_conv2d_supported(cin, cout, ksize, stride))


def pack_conv2d_weight(weight, stride, split=False, f16=True, lazy=False):
    """(Cout,Cin,k,k) -> MFMA A-fragment order, or None if the layer shape has no kernel.  split, lazy: as pack_conv3d_weight."""
    weight = _f32c(weight)
    cout, cin, k, _ = weight.shape
    n = _lib.load().mvs_conv2d_packed_weight_floats(cin, cout, k, stride)
    if n <= 0:
        return None
    packed = torch.empty(n, device=weight.device, dtype=torch.float32)

    def fill(dst):
        check(_lib.load().mvs_conv2d_pack_weights_f32(ptr(weight), cin, cout, k, stride, ptr(dst),
                                                      stream()), "mvs_conv2d_pack_weights_f32")
    import os
    k55 = stride == 2 and k == 5 and os.environ.get("MVS_CONV_SPLIT_55", "1") != "0"   # (A/B switch)
    if split and ((stride == 1 and k == 3) or k55) and conv_split_enabled():
        sp = pack_conv_weight_split(weight, stride)
        if sp is not None:
            _register_split(packed, sp, pack_conv_weight_split_f16(weight, stride) if (f16 and split_f16_enabled()) else None)
    if lazy and split_companion(packed) is not None:
        _register_lazy(packed, fill)
    else:
        fill(packed)
    return packed


def conv3d_wgrad(x_cl, g_cl, stride):
    """Weight gradient of a 3x3x3 layer on the matrix cores: x_cl [B,D,H,W,Cin], g_cl
    [B,Do,Ho,Wo,Cout] channels-last -> (Cout,Cin,3,3,3), or None when the shape has no kernel.
    For a transposed layer pass (grad_out, input, 2): the result is its (Cin,Cout,3,3,3)."""
    x_cl, g_cl = _f32c(x_cl), _f32c(g_cl)
    B, D, H, W, cin = x_cl.shape
    cout = g_cl.shape[-1]
    lib = _lib.load()
    if not lib.mvs_conv3d_wgrad_supported(cin, cout, stride):
        return None
    gw = torch.zeros((cout, cin, 3, 3, 3), device=x_cl.device, dtype=torch.float32)
    nbytes = int(lib.mvs_conv3d_wgrad_workspace_bytes(B, cin, cout, D, H, W, stride))
    ws = torch.empty((nbytes // 4,), device=x_cl.device, dtype=torch.float32)   # per-workgroup partial sums
    check(lib.mvs_conv3d_wgrad_f32(ptr(x_cl), ptr(g_cl), B, cin, cout, D, H, W, stride, ptr(gw), ptr(ws), nbytes,
                                   stream()), "mvs_conv3d_wgrad_f32")
    return gw


def conv3d_wgrad_c8(x_c8, g_cl):
    """Weight gradient of a conv0-class layer (Cout 8, stride 1) from its 8-channel-blocked input [B,D,H,Cin/8,W,8] and
    the output gradient [B,D,H,W,8] -> (8,Cin,3,3,3)."""
    x_c8, g_cl = _f32c(x_c8), _f32c(g_cl)
    B, D, H, G, W, _ = x_c8.shape
    cin = G * 8
    lib = _lib.load()
    gw = torch.zeros((8, cin, 3, 3, 3), device=x_c8.device, dtype=torch.float32)   # (the reduction adds into it)
    nbytes = int(lib.mvs_conv3d_wgrad_workspace_bytes(B, cin, 8, D, H, W, 1))
    ws = torch.empty((max(nbytes, 4) // 4,), device=x_c8.device, dtype=torch.float32)
    check(lib.mvs_conv3d_wgrad_c8_f32(ptr(x_c8), ptr(g_cl), B, cin, D, H, W, ptr(gw), ptr(ws), nbytes, stream()),
          "mvs_conv3d_wgrad_c8_f32")
    return gw


def wgrad_f16_enabled():
    """False when MVS_WGRAD_F16=0 keeps conv0's weight gradient on the fp32 matrix pipe (A/B switch; default: the two-piece
    fp16 kernel, mvs_conv3d_wgrad_c8_f16_f32, wherever the training node has both operands' absmax blocks)."""
    import os
    return os.environ.get("MVS_WGRAD_F16", "1") != "0"


def conv3d_wgrad_c8_f16(x_c8, x_absmax, g_cl, g_absmax):
    """conv0's weight gradient on the 16-bit matrix pipe (mvs_conv3d_wgrad_c8_f16_f32): x_c8 [B,D,H,4,W,8] with its absmax
    block, g_cl [B,D,H,W,8] with its absmax block -> (8,32,3,3,3); None if the shape has no such kernel."""
    x_c8, g_cl = _f32c(x_c8), _f32c(g_cl)
    B, D, H, G, W, _ = x_c8.shape
    cin = G * 8
    lib = _lib.load()
    if not lib.mvs_conv3d_wgrad_c8_f16_supported(B, cin, D, H, W):
        return None
    if tuple(g_cl.shape) != (B, D, H, W, 8):
        raise MvsHipError(f"conv3d_wgrad_c8_f16: grad_out {tuple(g_cl.shape)} does not match the volume {tuple(x_c8.shape)}")
    gw = torch.zeros((8, cin, 3, 3, 3), device=x_c8.device, dtype=torch.float32)   # (the reduction adds into it)
    nbytes = int(lib.mvs_conv3d_wgrad_c8_f16_workspace_bytes(B, cin, D, H, W))
    ws = torch.empty((nbytes // 4,), device=x_c8.device, dtype=torch.float32)     # one partial per wave
    check(lib.mvs_conv3d_wgrad_c8_f16_f32(ptr(x_c8), ctypes.c_void_p(x_absmax.data_ptr()), ptr(g_cl),
                                          ctypes.c_void_p(g_absmax.data_ptr()), B, cin, D, H, W, ptr(gw), ptr(ws), nbytes,
                                          stream()), "mvs_conv3d_wgrad_c8_f16_f32")
    return gw


class _VarianceConv0(torch.autograd.Function):
    """Training path: fused warp + variance -> conv0 (raw output, before BatchNorm) as ONE autograd node, so that the
    variance volume can live in the 8-channel-blocked layout conv0's split-operand bf16 kernel reads (0.87 -> 0.4 ms at
    640x512, D=192) without a mislabelled tensor leaving the node: backward = conv0's weight gradient from the blocked
    volume (mvs_conv3d_wgrad_c8_f32), its input gradient as an 8 -> 32 convolution (channels-last), and the variance
    kernel's backward on that."""

    @staticmethod
    def forward(ctx, ref16, srcs16, rts, depth_values, weight, align_corners):
        ref16, srcs16, depth_values = _f32c(ref16), _f32c(srcs16), _f32c(depth_values)
        if _depth_mode(depth_values) != 0:
            raise MvsHipError("the fused variance -> conv0 training op takes [B,D] depth planes")
        f16 = conv0_f16_enabled()      # conv0 on the two-piece fp16 kernel: its operand scale rides out of the sweep kernel
        amax = absmax_block(ref16.device) if f16 else None
        with stage("train.variance.fwd"):
            var = costvol_variance_c16(ref16, srcs16, rts, depth_values, align_corners, out_c8=True, absmax_out=amax)
        w = weight.detach().contiguous()
        pks = pack_conv3d_weight_f16x3(w) if f16 else pack_conv3d_weight_split(w)
        if pks is None:
            raise MvsHipError(f"no split-operand conv0 kernel for a {tuple(w.shape)} weight")
        split_stage_names.add("train.conv0.fwd")
        with stage("train.conv0.fwd"):
            out = conv3d_c8_f16x3(var, pks, amax) if f16 else conv3d_c8_split(var, pks, None, None, None, False)
        ctx.save_for_backward(ref16, srcs16, rts, depth_values, var, weight)
        ctx.ac = int(align_corners)
        ctx.var_absmax = amax          # (None on the bf16 path)
        return out

    @staticmethod
    def backward(ctx, g):
        ref16, srcs16, rts, depth_values, var, weight = ctx.saved_tensors
        g = _f32c(g)
        w = weight.detach()
        gw = g_ref = g_src = None
        g_absmax = None                # max |g|: the operand scale of both two-piece kernels below, one pass
        if ctx.needs_input_grad[4]:
            with stage("train.conv0.wgrad"):
                if ctx.var_absmax is not None and wgrad_f16_enabled():
                    g_absmax = absmax(g)
                    gw = conv3d_wgrad_c8_f16(var, ctx.var_absmax, g, g_absmax)
                if gw is None:
                    gw = conv3d_wgrad_c8(var, g)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            wt = w.flip(2, 3, 4).permute(1, 0, 2, 3, 4).contiguous()      # (32, 8, k) as a conv weight
            pk = pack_conv3d_weight(wt, False, 1, split=True)
            if g_absmax is None and f16_companion(pk) is not None:
                g_absmax = absmax(g)
            with stage("train.conv0.dgrad"):     # (8 -> 32 on the split-operand kernel: two pieces when the pack has them, scale = max |g|)
                gvar = conv3d(g, wt, channels_last=True, packed=pk, x_absmax=g_absmax if f16_companion(pk) is not None else None)   # [B,D,H,W,32]
            if split_companion(pk) is not None:
                split_stage_names.add("train.conv0.dgrad")
            B, G, H, W, _ = ref16.shape
            V, D = srcs16.shape[0] + 1, depth_values.shape[1]
            g_ref, g_src = torch.empty_like(ref16), torch.empty_like(srcs16)
            with stage("train.variance.bwd"):
                check(_lib.load().mvs_costvol_variance_bwd_f32(
                    ptr(gvar), ptr(ref16), ptr(srcs16), ptr(rts), ptr(depth_values), 0, B, V, G * 16, D, H, W,
                    ctx.ac, MVS_LAYOUT_C16, MVS_LAYOUT_NHWC, ptr(g_ref), ptr(g_src), stream()),
                    "mvs_costvol_variance_bwd_f32")
        return g_ref, g_src, None, None, gw, None


def variance_conv0_autograd(ref16, srcs16, rts, depth_values, conv0_weight, align_corners=False):
    """ref16 [B,C/16,H,W,16]; srcs16 [V-1,B,C/16,H,W,16]; conv0_weight (8,C,3,3,3) -> conv0's raw output [B,D,H,W,8]."""
    return _VarianceConv0.apply(ref16, srcs16, rts, depth_values, conv0_weight, align_corners)


def conv2d_wgrad(x, g_cl, ksize, stride, planar=False):
    """Weight gradient of a 2D layer on the matrix cores (mvs_conv2d_wgrad_f32): x [N,H,W,Cin] channels-last (planar:
    [N,Cin,H,W]), g_cl [N,Ho,Wo,Cout] -> (Cout,Cin,k,k)."""
    x, g_cl = _f32c(x), _f32c(g_cl)
    if planar:
        N, cin, H, W = x.shape
    else:
        N, H, W, cin = x.shape
    cout = g_cl.shape[-1]
    lib = _lib.load()
    nbytes = int(lib.mvs_conv2d_wgrad_workspace_bytes(N, cin, cout, H, W, ksize, stride))
    if nbytes == 0:
        raise MvsHipError(f"conv2d_wgrad: no kernel for k={ksize} stride={stride} Cin={cin} Cout={cout}")
    gw = torch.empty((cout, cin, ksize, ksize), device=x.device, dtype=torch.float32)
    ws = torch.empty((nbytes // 4,), device=x.device, dtype=torch.float32)
    check(lib.mvs_conv2d_wgrad_f32(ptr(x), ptr(g_cl), N, cin, cout, H, W, ksize, stride, int(planar), ptr(gw), ptr(ws),
                                   nbytes, stream()), "mvs_conv2d_wgrad_f32")
    return gw


def interleave2x2(classes):
    """[4,N,H,W,C] (class py*2+px holds the pixels (2y+py, 2x+px)) -> [N,2H,2W,C]."""
    classes = _f32c(classes)
    _, N, H, W, C = classes.shape
    out = torch.empty((N, 2 * H, 2 * W, C), device=classes.device, dtype=torch.float32)
    check(_lib.load().mvs_interleave2x2_f32(ptr(classes), N, H, W, C, ptr(out), stream()), "mvs_interleave2x2_f32")
    return out


def geo_consistency_matrices(K_ref, E_ref, src_Ks, src_Es):
    """The fp32 matrices of mvs_geo_consistency_f32, composed on the host with the numpy calls
    the reference makes per pair (eval.py:155-178): inverse(K_ref), K_ref, then per source view
    K_src, inverse(K_src), E_src @ inverse(E_ref), E_ref @ inverse(E_src)."""
    import numpy as np
    K_ref, E_ref = np.asarray(K_ref, dtype=np.float32), np.asarray(E_ref, dtype=np.float32)
    parts = [np.linalg.inv(K_ref).ravel(), K_ref.ravel()]
    for K, E in zip(src_Ks, src_Es):
        K, E = np.asarray(K, dtype=np.float32), np.asarray(E, dtype=np.float32)
        parts += [K.ravel(), np.linalg.inv(K).ravel(), np.matmul(E, np.linalg.inv(E_ref)).ravel(),
                  np.matmul(E_ref, np.linalg.inv(E)).ravel()]
    return np.concatenate(parts).astype(np.float32)


def geo_consistency(depth_ref, K_ref, E_ref, src_depths, src_Ks, src_Es, per_view=True):
    """Geometric-consistency check of one reference depth map [H,W] against S source depth maps
    [S,H,W] (device tensors; cameras as numpy) -- check_geometric_consistency for every source view
    and the sums of filter_depth (eval.py:190-262) in one kernel.  Returns a dict: geo_mask_sum
    [H,W] int32, depth_averaged [H,W] float64 and, with per_view, mask [S,H,W] bool,
    depth_reprojected [S,H,W], x_src / y_src [S,H,W]."""
    depth_ref, src_depths = _f32c(depth_ref), _f32c(src_depths)
    S, H, W = src_depths.shape
    if tuple(depth_ref.shape) != (H, W) or len(src_Ks) != S or len(src_Es) != S:
        raise MvsHipError("geo_consistency: depth_ref [H,W], src_depths [S,H,W], S cameras")
    dev = depth_ref.device
    mats = torch.from_numpy(geo_consistency_matrices(K_ref, E_ref, src_Ks, src_Es)).to(dev)
    geo_sum = torch.empty((H, W), device=dev, dtype=torch.int32)
    avg = torch.empty((H, W), device=dev, dtype=torch.float64)
    mask = torch.empty((S, H, W), device=dev, dtype=torch.uint8) if per_view else None
    drep = torch.empty((S, H, W), device=dev, dtype=torch.float32) if per_view else None
    xy = torch.empty((S, 2, H, W), device=dev, dtype=torch.float32) if per_view else None
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    check(_lib.load().mvs_geo_consistency_f32(ptr(depth_ref), ptr(src_depths), ptr(mats), S, H, W, vp(mask), ptr(drep),
                                              ptr(xy), vp(geo_sum), vp(avg), stream()), "mvs_geo_consistency_f32")
    out = {"geo_mask_sum": geo_sum, "depth_averaged": avg}
    if per_view:
        out.update(mask=mask.bool(), depth_reprojected=drep, x_src=xy[:, 0], y_src=xy[:, 1])
    return out


def cas_depth_hypotheses(prev_depth, ndepth, interval, full_hw, stage_hw):
    """Hypothesis volume [B,ndepth,Hs,Ws] of a cascade stage after the first from the previous
    stage's depth map [B,hp,wp] (cas_mvsnet.py:129-152, module.py:485-502) in one kernel."""
    prev_depth = _f32c(prev_depth)
    B, hp, wp = prev_depth.shape
    (H, W), (Hs, Ws) = full_hw, stage_hw
    out = torch.empty((B, ndepth, Hs, Ws), device=prev_depth.device, dtype=torch.float32)
    check(_lib.load().mvs_cas_depth_hypotheses_f32(ptr(prev_depth), B, hp, wp, H, W, Hs, Ws, ndepth,
                                                   float(ndepth / 2 * interval), ptr(out), stream()),
          "mvs_cas_depth_hypotheses_f32")
    return out


class _BnReluCL(torch.autograd.Function):
    """Training-mode BatchNorm + optional ReLU + optional skip add on channels-last rows
    (mvs_bn_train_fwd_f32 / mvs_bn_train_bwd_f32); running statistics are updated in place."""

    @staticmethod
    def forward(ctx, x, weight, bias, skip, running_mean, running_var, nbt, momentum, eps, relu, groups=1):
        x = _f32c(x)
        C = x.shape[-1]
        if x.shape[0] % groups:
            raise MvsHipError(f"bn_relu_cl: leading dimension {x.shape[0]} is not a multiple of groups={groups}")
        N = x.numel() // C // groups        # rows of one group
        lib = _lib.load()
        nbytes = int(lib.mvs_bn_train_workspace_bytes(C))
        if nbytes == 0:
            raise MvsHipError(f"bn_relu_cl: C={C} not supported")
        ws = torch.empty((nbytes // 4,), device=x.device, dtype=torch.float32)
        w, b = _f32c(weight.detach()), _f32c(bias.detach())
        sk = _f32c(skip) if skip is not None else None
        mean = torch.empty((groups, C), device=x.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        y = torch.empty_like(x)
        check(lib.mvs_bn_train_fwd_groups_f32(ptr(x), ptr(w), ptr(b), ptr(sk), int(groups), N, C, float(eps), float(momentum),
                                              int(relu), ptr(running_mean), ptr(running_var),
                                              ctypes.c_void_p(nbt.data_ptr()) if nbt is not None else None,
                                              ptr(mean), ptr(invstd), ptr(y),
                                              ptr(ws), nbytes, stream()), "mvs_bn_train_fwd_groups_f32")
        ctx.save_for_backward(x, w, b, mean, invstd)
        ctx.relu, ctx.has_skip, ctx.groups = int(relu), skip is not None, int(groups)
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, nbt) if t is not None])
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, b, mean, invstd = ctx.saved_tensors
        gy = _f32c(gy)
        C = x.shape[-1]
        N = x.numel() // C // ctx.groups
        lib = _lib.load()
        nbytes = int(lib.mvs_bn_train_workspace_bytes(C))
        ws = torch.empty((nbytes // 4,), device=x.device, dtype=torch.float32)
        gx = torch.empty_like(x)
        gw, gb = torch.empty_like(w), torch.empty_like(b)
        check(lib.mvs_bn_train_bwd_groups_f32(ptr(gy), ptr(x), ptr(w), ptr(b), ptr(mean), ptr(invstd), ctx.groups, N, C, ctx.relu,
                                              ptr(gx), ptr(gw), ptr(gb), ptr(ws), nbytes, stream()), "mvs_bn_train_bwd_groups_f32")
        return gx, gw, gb, (gy if ctx.has_skip else None), None, None, None, None, None, None, None


def bn_relu_cl(x, bn, relu=True, skip=None, groups=1):
    """relu(bn(x)) [+ skip] with batch statistics for a channels-last tensor [..., C] and an
    nn.BatchNorm module in training mode (momentum must be a number, affine + running stats).
    groups: x[0] splits into that many consecutive blocks, each normalised with its OWN statistics; the running statistics are
    updated once per block, in order -- what `groups` separate calls of the module would do (the reference's per-view FeatureNet
    calls, mvsnet.py:146, as one batch)."""
    if bn.momentum is None or bn.weight is None:
        raise MvsHipError("bn_relu_cl: needs an affine BatchNorm with a numeric momentum")
    return _BnReluCL.apply(x, bn.weight, bn.bias, skip, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                           bn.momentum, bn.eps, relu, groups)


def fpn_tail_supported(H, W):
    import os
    return os.environ.get("MVS_FPN_TAIL", "1") != "0" and bool(_lib.load().mvs_fpn_tail_supported(int(H), int(W)))


def pack_fpn_tail_weight(w_out):
    """out3's (8, 32, 3, 3) weight -> the bf16 hi/mid/lo A fragments of fpn_tail (opaque bytes)."""
    w_out = _f32c(w_out)
    if tuple(w_out.shape) != (8, 32, 3, 3):
        raise MvsHipError(f"pack_fpn_tail_weight: out3 weight {tuple(w_out.shape)} is not (8, 32, 3, 3)")
    packed = torch.empty(_lib.load().mvs_fpn_tail_packed_bytes() // 4, device=w_out.device, dtype=torch.float32)
    check(_lib.load().mvs_fpn_tail_pack_weights_f32(ptr(w_out), ptr(packed), stream()), "mvs_fpn_tail_pack_weights_f32")
    return packed


def fpn_tail(fine, coarse, w_inner, b_inner, packed_out, b_out=None):
    """out3(nearest_x2(coarse) + inner2(fine)) of CasMVSNet's FPN (module.py:396-398) in one kernel: fine [N,H,W,8],
    coarse [N,H/2,W/2,32] channels-last -> [N,H,W,8]."""
    fine, coarse = _f32c(fine), _f32c(coarse)
    N, H, W, C = fine.shape
    if C != 8 or tuple(coarse.shape) != (N, H // 2, W // 2, 32) or w_inner.numel() != 256:
        raise MvsHipError(f"fpn_tail: fine {tuple(fine.shape)} / coarse {tuple(coarse.shape)} are not the 8- and 32-channel maps")
    out = torch.empty((N, H, W, 8), device=fine.device, dtype=torch.float32)
    check(_lib.load().mvs_fpn_tail_f32(ptr(fine), ptr(coarse), ptr(_f32c(w_inner)), ptr(b_inner), ptr(packed_out), ptr(b_out),
                                       N, H, W, ptr(out), stream()), "mvs_fpn_tail_f32")
    return out


def feature_head_enabled():
    """False when MVS_FEATURE_HEAD=0 keeps FeatureNet's first two layers as two launches (A/B switch)."""
    import os
    return os.environ.get("MVS_FEATURE_HEAD", "1") != "0"


def feature_head_supported(H, W):
    return bool(_lib.load().mvs_feature_head_supported(int(H), int(W)))


def pack_feature_head_weight(w1):
    """conv1's (8, 8, 3, 3) weight -> the bf16 hi/mid/lo A fragments of feature_head (opaque bytes)."""
    w1 = _f32c(w1)
    if tuple(w1.shape) != (8, 8, 3, 3):
        raise MvsHipError(f"pack_feature_head_weight: conv1 weight {tuple(w1.shape)} is not (8, 8, 3, 3)")
    packed = torch.empty(_lib.load().mvs_feature_head_packed_bytes() // 4, device=w1.device, dtype=torch.float32)
    check(_lib.load().mvs_feature_head_pack_weights_f32(ptr(w1), ptr(packed), stream()), "mvs_feature_head_pack_weights_f32")
    return packed


def feature_head(img_nchw, w0, scale0, shift0, packed1, scale1, shift1, out_absmax=None):
    """FeatureNet's conv0 + BN + ReLU + conv1 + BN + ReLU (mvsnet.py:11-12) in one kernel: the [N,3,H,W] image batch ->
    [N,H,W,8] channels-last.  w0: conv0's weight in PyTorch layout (8,3,3,3); packed1: pack_feature_head_weight(w1);
    scale / shift: the folded BatchNorm affines."""
    x = _f32c(img_nchw)
    N, C, H, W = x.shape
    if C != 3 or tuple(w0.shape) != (8, 3, 3, 3):
        raise MvsHipError(f"feature_head: image {tuple(x.shape)} / conv0 weight {tuple(w0.shape)} are not the 3 -> 8 head")
    out = torch.empty((N, H, W, 8), device=x.device, dtype=torch.float32)
    check(_lib.load().mvs_feature_head_absmax_f32(
        ptr(x), ptr(_f32c(w0)), ptr(scale0), ptr(shift0), ptr(packed1), ptr(scale1), ptr(shift1), N, H, W, ptr(out),
        ctypes.c_void_p(out_absmax.data_ptr()) if out_absmax is not None else None, stream()), "mvs_feature_head_absmax_f32")
    return out


def conv2d(x, packed, cin, cout, ksize, stride, scale=None, shift=None, relu=False, planar=False, coarse=None,
           out_c4=False, out=None, x_absmax=None, out_absmax=None):
    """FeatureNet convolution.  x: [B,H,W,cin] channels-last, or (planar) the [B,3,H,W]
    image.  relu: False/True, or 2 for LeakyReLU(0.1).  coarse: [B,Ho/2,Wo/2,cout], added through
    a nearest x2 upsample (FPN top-down step).  Returns [B,Ho,Wo,cout] channels-last, or with out_c4
    the 4-channel blocked [B,cout/4,Ho,Wo,4] (MVS_LAYOUT_C4)."""
    x = _f32c(x)
    if planar:
        B, _, H, W = x.shape
    else:
        B, H, W, _ = x.shape
    pad = ksize // 2
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    sp = split_companion(packed)
    if sp is not None and (ksize, stride) in ((3, 1), (5, 2)) and not planar and coarse is None and out is None:
        f16 = f16_companion(packed) if x_absmax is not None else None      # x_absmax / out_absmax: as conv3d
        if f16 is not None:
            res = conv_split_f16(x, f16, cout, x_absmax, scale, shift, None, int(relu), kd=1, out_c4=out_c4, stride=stride,
                                 out_absmax=out_absmax, soft=True)
            if res is not None:
                return res
        res = conv_split(x, sp, cout, scale, shift, None, int(relu), kd=1, out_c4=out_c4, stride=stride, soft=True)
        if res is not None:       # else: beyond the split launcher's 32-bit halo offsets -> the fp32 MFMA kernel
            if out_absmax is not None:
                absmax(res, out_absmax)
            return res
    shape = (B, cout // 4, Ho, Wo, 4) if out_c4 else (B, Ho, Wo, cout)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != shape or not out.is_contiguous() or out.dtype != torch.float32:
        raise MvsHipError(f"conv2d: out must be a contiguous float32 tensor of shape {shape}")
    if coarse is not None:
        coarse = _f32c(coarse)
        if tuple(coarse.shape) != (B, Ho // 2, Wo // 2, cout) or Ho % 2 or Wo % 2:
            raise MvsHipError(f"conv2d: coarse {tuple(coarse.shape)} is not half of {(B, Ho, Wo, cout)}")
    materialize_packed(packed)
    check(_lib.load().mvs_conv2d_absmax_f32(ptr(x), ptr(packed), ptr(scale), ptr(shift), ptr(coarse), int(relu), B, cin,
                                            cout, H, W, ksize, stride, int(planar) | (2 if out_c4 else 0), ptr(out),
                                            ctypes.c_void_p(out_absmax.data_ptr()) if out_absmax is not None else None, stream()),
          "mvs_conv2d_absmax_f32")
    return out


def conv2d_pair_enabled():
    """False when MVS_CONV2D_PAIR=0 keeps FeatureNet's conv3 / conv4 as two launches (A/B).  Default: one kernel
    (mvs_conv2d_pair_guarded_f16_f32)."""
    import os
    return split_f16_enabled() and os.environ.get("MVS_CONV2D_PAIR", "1") != "0"


def pack_conv2d_pair(w1, w2):
    """Two (C, C, 3, 3) weights -> the fragments of the fused pair kernel, or None when the shape has none (C = 16 only)."""
    w1, w2 = _f32c(w1), _f32c(w2)
    C = w1.shape[0]
    lib = _lib.load()
    if tuple(w1.shape) != (C, C, 3, 3) or tuple(w2.shape) != (C, C, 3, 3):
        return None
    n = lib.mvs_conv2d_pair_packed_bytes(C)
    if n == 0:
        return None
    out = torch.empty(n, device=w1.device, dtype=torch.uint8)
    check(lib.mvs_conv2d_pair_pack_weights_f32(ptr(w1), ptr(w2), C, ctypes.c_void_p(out.data_ptr()), stream()),
          "mvs_conv2d_pair_pack_weights_f32")
    return out


def conv2d_pair(x, x_absmax, packed_pair, p1, p2, out_c4=False, out_absmax=None, flag=None):
    """Two consecutive 3x3 stride-1 layers (parameter dicts p1, p2 with 'packed' -- whose two-piece companions serve the unfused
    path --, 'scale', 'shift', 'relu') as one kernel; the two layers of conv_split are enqueued behind it and run only if its range
    guard declined (no host synchronisation).  x [N,H,W,C] -> [N,H,W,C] (or out_c4 [N,C/4,H,W,4])."""
    x = _f32c(x)
    N, H, W, C = x.shape
    out = torch.empty((N, C // 4, H, W, 4) if out_c4 else (N, H, W, C), device=x.device, dtype=torch.float32)
    mid = torch.empty((N, H, W, C), device=x.device, dtype=torch.float32)       # touched only if the guard declines
    if flag is None:        # (a caller that zeroes its absmax blocks in one fill hands over ABSMAX_WORDS + 64 of those words)
        flag = torch.zeros(ABSMAX_WORDS + 64, device=x.device, dtype=torch.int32)
    elif flag.numel() < ABSMAX_WORDS + 64 or not flag.is_contiguous():
        raise MvsHipError("conv2d_pair: flag = ABSMAX_WORDS + 64 zeroed contiguous int32 words")
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    check(_lib.load().mvs_conv2d_pair_guarded_f16_f32(
        ptr(x), vp(x_absmax), vp(packed_pair), vp(f16_companion(p1["packed"])), vp(f16_companion(p2["packed"])),
        ptr(p1["scale"]), ptr(p1["shift"]), ptr(p2["scale"]), ptr(p2["shift"]), int(bool(p2["relu"])), N, C, H, W, int(bool(out_c4)),
        ptr(mid), ptr(out), vp(out_absmax), vp(flag), stream()), "mvs_conv2d_pair_guarded_f16_f32")
    return out


# ------------------------------------------------------------- K4+K5 regress
class _SoftmaxRegress(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, depth_values, clamp_idx, want_prob):
        cost, depth_values = _f32c(cost), _f32c(depth_values)
        B, D, H, W = cost.shape
        mode = _depth_mode(depth_values)
        depth = torch.empty((B, H, W), device=cost.device, dtype=torch.float32)
        conf = torch.empty((B, H, W), device=cost.device, dtype=torch.float32)
        prob = torch.empty_like(cost) if want_prob else None
        check(_lib.load().mvs_softmax_regress_conf_f32(
            ptr(cost), ptr(depth_values), mode, int(clamp_idx), B, D, H, W, ptr(depth), ptr(conf),
            ptr(prob), stream()), "mvs_softmax_regress_conf_f32")
        ctx.save_for_backward(cost, depth_values)
        ctx.meta = (B, D, H, W, mode)
        ctx.mark_non_differentiable(conf)
        if want_prob:
            ctx.mark_non_differentiable(prob)
            return depth, conf, prob
        return depth, conf, None

    @staticmethod
    def backward(ctx, g_depth, g_conf, g_prob):
        cost, depth_values = ctx.saved_tensors
        B, D, H, W, mode = ctx.meta
        g_depth = _f32c(g_depth)
        g_cost = torch.empty_like(cost)
        check(_lib.load().mvs_softmax_regress_bwd_f32(
            ptr(cost), ptr(depth_values), mode, ptr(g_depth), B, D, H, W, ptr(g_cost), stream()),
            "mvs_softmax_regress_bwd_f32")
        return g_cost, None, None, None


def softmax_regress_conf(cost, depth_values, clamp_idx=False, want_prob=False):
    """cost [B,D,H,W] -> (depth [B,H,W], photometric confidence [B,H,W], prob or None)."""
    return _SoftmaxRegress.apply(cost, depth_values, clamp_idx, want_prob)
