"""conv0 on a volume that arrives as fp16 pairs (mvs_conv3d_c8h_f16x3_f32) against the kernel that splits an fp32 volume itself
(mvs_conv3d_c8_f16x3_f32): bit-equality on ragged shapes and the time at BASELINE configs[1]'s volume."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvs_amd import ops

res = {"bit_equal": {}, "time": {}}
for B, D, H, W, C in ((1, 12, 20, 70, 32), (2, 7, 9, 37, 16), (1, 5, 30, 33, 8), (1, 17, 6, 64, 32), (1, 4, 4, 32, 32), (1, 40, 12, 31, 32)):
    g = torch.Generator().manual_seed(D * 7 + W)
    x = ((torch.randn(B, D, H, C // 8, W, 8, generator=g) * torch.rand(B, D, H, C // 8, W, 8, generator=g) ** 4).square()).cuda()
    w = (torch.randn(8, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5).cuda()
    sc, sh = (torch.rand(8, generator=g) + 0.5).cuda(), (torch.randn(8, generator=g) * 0.1).cuda()
    r = torch.randn(B, D, H, W, 8, generator=g).cuda()
    pf = ops.pack_conv3d_weight_f16x3(w)
    blk = ops.absmax(x)
    a = ops.conv3d_c8_f16x3(x, pf, blk, sc, sh, r, True)
    xp = ops.c8_to_c8h(x, blk)
    b = ops.conv3d_c8h_f16x3(xp, (B, C, D, H, W), pf, blk, sc, sh, r, True)
    res["bit_equal"][str((B, D, H, W, C))] = bool(torch.equal(a, b))
    print((B, D, H, W, C), torch.equal(a, b), float((a - b).abs().max()), flush=True)

B, D, H, W, C = 1, 192, 296, 400, 32
g = torch.Generator().manual_seed(5)
x = ((torch.randn(B, D, H, C // 8, W, 8, generator=g) * torch.rand(B, D, H, C // 8, W, 8, generator=g) ** 4).square()).cuda()
w = (torch.randn(8, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5).cuda()
sc, sh = (torch.rand(8, generator=g) + 0.5).cuda(), (torch.randn(8, generator=g) * 0.1).cuda()
pf = ops.pack_conv3d_weight_f16x3(w)
blk = ops.absmax(x)
xp = ops.c8_to_c8h(x, blk)
fns = {"f16x3 (splits fp32 itself)": lambda: ops.conv3d_c8_f16x3(x, pf, blk, sc, sh, None, True),
       "pairs (pre-split input)": lambda: ops.conv3d_c8h_f16x3(xp, (B, C, D, H, W), pf, blk, sc, sh, None, True),
       "c8_to_c8h": lambda: ops.c8_to_c8h(x, blk)}
for rep in range(2):
    for k, fn in fns.items():
        fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
        for a_, b_ in ev:
            a_.record(); fn(); b_.record()
        torch.cuda.synchronize()
        t = sorted(a_.elapsed_time(b_) for a_, b_ in ev)
        res["time"].setdefault(k, []).append({"min": round(t[0], 4), "med": round(t[4], 4)})
        print(k, t[0], t[4], flush=True)
res["fullsize_bit_equal"] = bool(torch.equal(fns["f16x3 (splits fp32 itself)"](), fns["pairs (pre-split input)"]()))
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/conv0_pairs.json", "w"), indent=1)
