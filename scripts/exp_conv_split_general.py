"""mvs_conv_split_f32 (3x3(x3) layers with 16/32/64 channels on the bf16 matrix pipe, operands split exactly) against
an fp64 convolution and the fp32 MFMA kernels; times at layer shapes of configs 2 and 4.
  python scripts/exp_conv_split_general.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops  # noqa: E402


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)


def case(kd, cin, cout, shape, reps=0, check=True, relu=1, seed=0):
    g = torch.Generator().manual_seed(seed)
    if kd == 3:
        B, D, H, W = shape
        x = torch.randn(B, cin, D, H, W, generator=g) * torch.rand(B, cin, D, H, W, generator=g) ** 2
        w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    else:
        N, H, W = shape
        x = torch.randn(N, cin, H, W, generator=g) * torch.rand(N, cin, H, W, generator=g) ** 2
        w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    xd, wd = x.cuda(), w.cuda()
    x_cl = xd.permute(0, 2, 3, 4, 1).contiguous() if kd == 3 else xd.permute(0, 2, 3, 1).contiguous()
    pks = ops.pack_conv_weight_split(wd)
    res = torch.randn(x_cl.shape[:-1] + (cout,), generator=g).cuda() if check else None
    y = ops.conv_split(x_cl, pks, cout, scale.cuda(), shift.cuda(), res, relu, kd)
    torch.cuda.synchronize()
    line = f"kd{kd} {cin}->{cout} {shape}:"
    if check:
        conv = torch.nn.functional.conv3d if kd == 3 else torch.nn.functional.conv2d
        ref = conv(x.double(), w.double(), padding=1)
        vs = (1, cout) + (1,) * (kd == 3 and 3 or 2)
        ref = ref * scale.double().view(vs) + shift.double().view(vs)
        ref = torch.relu(ref) if relu == 1 else (torch.where(ref > 0, ref, ref * 0.1) if relu == 2 else ref)
        ref = (ref.permute(0, 2, 3, 4, 1) if kd == 3 else ref.permute(0, 2, 3, 1)) + res.cpu().double()
        err = (y.cpu().double() - ref).abs().max().item()
        line += f" |ref|max {ref.abs().max().item():.3g} split err {err:.3g}"
        if kd == 3:
            pk = ops.pack_conv3d_weight(wd, False, 1)
            y32 = ops.conv3d(x_cl, wd, scale.cuda(), shift.cuda(), res, relu == 1, False, 1, channels_last=True, packed=pk, impl=ops.IMPL_MFMA)
            line += f" fp32-mfma err {(y32.cpu().double() - ref).abs().max().item():.3g}"
    if reps:
        line += f"  split {timeit(lambda: ops.conv_split(x_cl, pks, cout, None, None, None, relu, kd), reps):.3f} ms"
        if kd == 3:
            pk = ops.pack_conv3d_weight(wd, False, 1)
            line += f"  fp32 {timeit(lambda: ops.conv3d(x_cl, wd, None, None, None, True, False, 1, channels_last=True, packed=pk, impl=ops.IMPL_MFMA), reps):.3f} ms"
        else:
            pk2 = ops.pack_conv2d_weight(wd, 1)
            line += f"  fp32 {timeit(lambda: ops.conv2d(x_cl, pk2, cin, cout, 3, 1, None, None, 1), reps):.3f} ms"
    print(line, flush=True)


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    case(3, 16, 16, (1, 5, 9, 21))
    case(3, 32, 32, (2, 6, 7, 33), seed=1)
    case(3, 64, 64, (1, 4, 10, 18), seed=2, relu=0)
    case(3, 16, 32, (1, 9, 17, 40), seed=3)
    case(1, 16, 16, (3, 21, 45), seed=4, relu=2)
    case(1, 32, 32, (2, 40, 70), seed=5)
    case(1, 64, 32, (1, 33, 35), seed=6, relu=2)
    case(1, 64, 64, (1, 18, 50), seed=7, relu=2)
    if reps:
        case(3, 16, 16, (1, 96, 148, 200), reps, check=False)     # config 2 conv2
        case(3, 32, 32, (1, 48, 74, 100), reps, check=False)      # conv4
        case(3, 64, 64, (1, 24, 37, 50), reps, check=False)       # conv6
        case(1, 16, 16, (5, 592, 800), reps, check=False)         # FeatureNet conv3 / conv4
        case(1, 32, 32, (5, 296, 400), reps, check=False)         # FeatureNet conv6 / feature
        case(1, 64, 64, (7, 1056, 1920), reps, check=False)       # CVP pyramid, full resolution
        case(3, 16, 16, (1, 8, 528, 960), reps, check=False)      # CVP cost regularisation, level 1
        for cin, cout, shp in ((8, 16, (1, 192, 296, 400)), (16, 32, (1, 96, 148, 200)), (32, 64, (1, 48, 74, 100))):   # conv1, conv3, conv5
            x = torch.randn(*shp, cin, device="cuda")
            w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
            pks, pk = ops.pack_conv_weight_split(w, 2), ops.pack_conv3d_weight(w, False, 2)
            print(f"stride 2 {cin}->{cout} {shp}: split {timeit(lambda: ops.conv_split(x, pks, cout, None, None, None, 1, 3, stride=2), reps):.3f} ms"
                  f"  fp32 {timeit(lambda: ops.conv3d(x, w, None, None, None, True, False, 2, channels_last=True, packed=pk, impl=ops.IMPL_MFMA), reps):.3f} ms", flush=True)
