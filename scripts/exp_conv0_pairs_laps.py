"""Phase cycles (tuning build, MVS_CONV_SPLIT_ABL=128) or the time with the MFMA phase removed (=2) of the pre-split-input conv0."""
import os as _os; _os.environ["MVS_HIP_TUNING"] = "1"
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops
D, H, W = 192, 296, 400
g = torch.Generator().manual_seed(5)
x = ((torch.randn(1, D, H, 4, W, 8, generator=g) * torch.rand(1, D, H, 4, W, 8, generator=g) ** 4).square()).cuda()
w = torch.randn(8, 32, 3, 3, 3, device="cuda") * 0.1
pf = ops.pack_conv3d_weight_f16x3(w)
blk = ops.absmax(x)
xp = ops.c8_to_c8h(x, blk)
abl = int(os.environ.get("MVS_CONV_SPLIT_ABL", "0"))
if abl & 128:
    dbg = torch.zeros(256 * 8 * 8 * 2, device="cuda")
    ops.conv3d_c8h_f16x3(xp, (1, 32, D, H, W), pf, blk, None, None, dbg, True)
    torch.cuda.synchronize()
    tt = dbg.view(torch.int64).view(256, 8, 8).double()
    steps = -(-D // 4) * -(-H // 4) * -(-W // 32) * 4 / 256
    names = ["barrier", "MFMA phase", "loop", "epilogue"]
    print(json.dumps({"cycles_per_step": round(tt[:, :, :4].sum(-1).mean().item() / steps),
                      "phases": {nm: {"mean": round(tt[:, :, k].mean().item() / steps), "slowest_wave": round(tt[:, :, k].max(1).values.mean().item() / steps),
                                      "fastest_wave": round(tt[:, :, k].min(1).values.mean().item() / steps)} for k, nm in enumerate(names)}}))
else:
    fn = lambda: ops.conv3d_c8h_f16x3(xp, (1, 32, D, H, W), pf, blk, None, None, None, True)
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    print("ABL", abl, "min %.3f med %.3f ms" % (t[0], t[3]))
