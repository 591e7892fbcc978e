"""Per-phase cycle counts of the conv_split kernel (tuning build, MVS_CONV_SPLIT_LAPS=1).
  python scripts/exp_conv_split_laps.py kd cin cout  N|B,D H W"""
import os as _os; _os.environ.setdefault("MVS_HIP_TUNING", "1")   # needs python -m mvs_amd.build --tuning
import json, os, sys, torch
os.environ["MVS_CONV_SPLIT_LAPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops
kd, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dims = [int(v) for v in sys.argv[4:]]
x = torch.randn(*(dims + [cin]), device="cuda")
w = torch.randn(*([cout, cin] + [3] * (3 if kd == 3 else 2)), device="cuda") * 0.05
pks = ops.pack_conv_weight_split(w)
dbg = torch.zeros(256 * 8 * 8 * 2, device="cuda")
ops.conv_split(x, pks, cout, None, None, dbg, 1, kd)          # (Cout 64 = two launches: the second overwrites the first's counts)
torch.cuda.synchronize()
t = dbg.view(torch.int64).view(256, 8, 8).double()
names = ["barrier A", "split pass", "barrier B", "MFMA phase", "epilogue"]
tot = t[:, :, :5].sum(-1).mean().item()
print(json.dumps({"kd": kd, "cin": cin, "cout": cout, "dims": dims, "cycles_per_wave": round(tot),
                  "phases": {nm: {"share": round(t[:, :, k].mean().item() / tot, 4), "waves_0_3": round(t[:, :4, k].mean().item()),
                                  "waves_4_7": round(t[:, 4:, k].mean().item())} for k, nm in enumerate(names)}}, indent=1))
