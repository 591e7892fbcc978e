"""mvs_deconv_split_f32 (transposed 3x3x3 stride-2 layers on the bf16 matrix pipe, operands split exactly) against an fp64
transposed convolution and the fp32 MFMA kernel; times at config 2's conv7 / conv9 / conv11.
  python scripts/exp_deconv_split.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)


def case(cin, cout, shape, reps=0, check=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    B, D, H, W = shape
    x = torch.randn(B, cin, D, H, W, generator=g) * torch.rand(B, cin, D, H, W, generator=g) ** 2
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) / (27 * cin / 8) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    xd, wd = x.cuda(), w.cuda()
    x_cl = xd.permute(0, 2, 3, 4, 1).contiguous()
    pks = ops.pack_deconv_weight_split(wd)
    res = torch.randn(B, 2 * D, 2 * H, 2 * W, cout, generator=g).cuda()
    y = ops.deconv_split(x_cl, pks, cout, scale.cuda(), shift.cuda(), res, True)
    torch.cuda.synchronize()
    line = f"deconv {cin}->{cout} {shape}:"
    pk = ops.pack_conv3d_weight(wd, True, 2)
    if check:
        ref = torch.nn.functional.conv_transpose3d(x.double(), w.double(), stride=2, padding=1, output_padding=1)
        ref = torch.relu(ref * scale.double().view(1, cout, 1, 1, 1) + shift.double().view(1, cout, 1, 1, 1))
        ref = ref.permute(0, 2, 3, 4, 1) + res.cpu().double()
        y32 = ops.conv3d(x_cl, wd, scale.cuda(), shift.cuda(), res, True, True, 2, channels_last=True, packed=pk, impl=ops.IMPL_MFMA)
        line += f" |ref|max {ref.abs().max().item():.3g} split err {(y.cpu().double() - ref).abs().max().item():.3g}"
        line += f" fp32-mfma err {(y32.cpu().double() - ref).abs().max().item():.3g}"
    if reps:
        line += f"  split {timeit(lambda: ops.deconv_split(x_cl, pks, cout, None, None, res, True), reps):.3f} ms"
        line += f"  fp32 {timeit(lambda: ops.conv3d(x_cl, wd, None, None, res, True, True, 2, channels_last=True, packed=pk, impl=ops.IMPL_MFMA), reps):.3f} ms"
    print(line, flush=True)


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    case(16, 8, (1, 3, 5, 9))
    case(16, 8, (2, 5, 6, 21), seed=1)
    case(32, 16, (1, 4, 7, 19), seed=2)
    case(64, 32, (1, 3, 5, 17), seed=3)
    if reps:
        case(64, 32, (1, 24, 37, 50), reps, check=False)       # conv7
        case(32, 16, (1, 48, 74, 100), reps, check=False)      # conv9
        case(16, 8, (1, 96, 148, 200), reps, check=False)      # conv11
