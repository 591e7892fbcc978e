"""conv0-class layer on the bf16 matrix pipe with exactly split fp32 operands (mvs_conv3d_c8_bf16x6_f32)
against (a) an fp64 convolution, (b) the fp32 MFMA kernel; and its time at the headline shape.

  python scripts/exp_conv_split.py [reps]
Prints one line per case: max |err| of both kernels against fp64 and the split kernel's time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops  # noqa: E402


def one(B, Cin, D, H, W, reps, seed=0, check=True):
    g = torch.Generator().manual_seed(seed)
    # variance-volume-like input: non-negative, wide dynamic range
    x = (torch.randn(B, Cin, D, H, W, generator=g) * torch.rand(B, Cin, D, H, W, generator=g) ** 4).square()
    w = torch.randn(8, Cin, 3, 3, 3, generator=g) * 0.1
    scale = torch.rand(8, generator=g) + 0.5
    shift = torch.randn(8, generator=g) * 0.1
    xd, wd = x.cuda(), w.cuda()
    x_c8 = ops.nchw_to_c8(xd)
    pk = ops.pack_conv3d_weight(wd, False, 1)
    pks = ops.pack_conv3d_weight_split(wd)
    y_f32 = ops.conv3d(x_c8, wd, scale.cuda(), shift.cuda(), relu=True, packed=pk, impl=ops.IMPL_MFMA, in_c8=True)
    y_spl = ops.conv3d_c8_split(x_c8, pks, scale.cuda(), shift.cuda(), relu=True)
    torch.cuda.synchronize()
    line = f"B{B} Cin{Cin} D{D} H{H} W{W}:"
    if check:
        ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
        ref = torch.relu(ref * scale.double().view(1, 8, 1, 1, 1) + shift.double().view(1, 8, 1, 1, 1))
        ref = ref.permute(0, 2, 3, 4, 1)
        e32 = (y_f32.cpu().double() - ref).abs().max().item()
        esp = (y_spl.cpu().double() - ref).abs().max().item()
        line += f" |ref|max {ref.abs().max().item():.3g}  fp32-mfma err {e32:.3g}  split err {esp:.3g}"
    line += f"  split-vs-fp32 {(y_spl - y_f32).abs().max().item():.3g}"
    if reps:
        for fn, name in ((lambda: ops.conv3d(x_c8, wd, None, None, relu=True, packed=pk, impl=ops.IMPL_MFMA, in_c8=True), "fp32"),
                         (lambda: ops.conv3d_c8_split(x_c8, pks, None, None, relu=True), "split")):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            line += f"  {name} {(time.perf_counter() - t0) / reps * 1e3:.3f} ms"
    print(line, flush=True)


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    one(1, 32, 8, 12, 40, 0)
    one(2, 16, 5, 7, 33, 0, seed=1)
    one(1, 8, 9, 6, 70, 0, seed=2)
    one(1, 32, 48, 32, 40, reps, seed=3)
    one(1, 32, 192, 128, 160, reps, seed=4, check=False)
