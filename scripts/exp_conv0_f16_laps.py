"""Phase cycles of the two-piece fp16 conv0 kernel at BASELINE configs[1]'s volume (tuning build: clock64() laps around the
phases of a step = one 8-channel chunk of one tile; MVS_LAP in csrc/conv_f16x3.hip).  python -m mvs_amd.build --tuning first."""
import os as _os; _os.environ["MVS_HIP_TUNING"] = "1"; _os.environ["MVS_CONV_SPLIT_ABL"] = "128"
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops
D, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (192, 296, 400)
x = torch.randn(1, D, H, 4, W, 8, device="cuda").square()
w = torch.randn(8, 32, 3, 3, 3, device="cuda") * 0.1
pf = ops.pack_conv3d_weight_f16x3(w)
dbg = torch.zeros(256 * 8 * 8 * 2, device="cuda")     # int64 [256 workgroups][8 waves][8]
ops.conv3d_c8_f16x3(x, pf, None, None, None, dbg, relu=True)
torch.cuda.synchronize()
tt = dbg.view(torch.int64).view(256, 8, 8).double()
names = ["-", "barrier A", "split pass", "barrier B", "-", "MFMA phase", "epilogue"]
steps = -(-D // 4) * -(-H // 4) * -(-W // 32) * 4 / 256
out = {"shape": [D, H, W], "steps_per_workgroup": steps, "cycles_per_step": round(tt[:, :, :7].sum(-1).mean().item() / steps),
       "phases_per_step": {nm: {"mean": round(tt[:, :, k].mean().item() / steps), "slowest_wave": round(tt[:, :, k].max(1).values.mean().item() / steps),
                                "fastest_wave": round(tt[:, :, k].min(1).values.mean().item() / steps)} for k, nm in enumerate(names) if nm != "-"}}
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/conv0_f16_laps.json", "w"), indent=1)
