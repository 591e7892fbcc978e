"""time of the split-operand conv0 kernel at one shape (tuning: MVS_CONV_SPLIT_ABL, MVS_CONV_SPLIT_DOT2)"""
import os as _os; _os.environ.setdefault("MVS_HIP_TUNING", "1")   # needs python -m mvs_amd.build --tuning
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops
D, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (192, 296, 400)
x = torch.randn(1, D, H, 4, W, 8, device="cuda")
w = torch.randn(8, 32, 3, 3, 3, device="cuda") * 0.1
pk, pks = ops.pack_conv3d_weight(w, False, 1), ops.pack_conv3d_weight_split(w)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)
ABL = int(os.environ.get("MVS_CONV_SPLIT_ABL", "0"))
if ABL & 128:
    # tuning build with clock64() laps around the phases of a step (mvs_amd/csrc/conv_bf16x6.hip, MVS_LAP)
    import json
    dbg = torch.zeros(256 * 8 * 8 * 2, device="cuda")     # int64 [256 workgroups][8 waves][8]
    ops.conv3d_c8_split(x, pks, None, None, dbg, relu=True)
    torch.cuda.synchronize()
    tt = dbg.view(torch.int64).view(256, 8, 8).double()
    names = ["copy wait", "barrier A", "split pass", "barrier B", "copy issue", "MFMA phase", "epilogue"]
    tot = tt[:, :, :7].sum(-1).mean().item()
    tiles = -(-D // 4) * -(-H // 4) * -(-W // 32)
    out = {"shape": [D, H, W], "steps_per_workgroup": tiles * 4 / 256, "cycles_per_wave": round(tot),
           "phases": {nm: {"mean": round(tt[:, :, k].mean().item()), "share": round(tt[:, :, k].mean().item() / tot, 4),
                           "min": round(tt[:, :, k].min().item()), "max": round(tt[:, :, k].max().item())} for k, nm in enumerate(names)}}
    print(json.dumps(out, indent=1))
    sys.exit(0)
print("ABL", os.environ.get("MVS_CONV_SPLIT_ABL", "0"), "DOT2", os.environ.get("MVS_CONV_SPLIT_DOT2", "1"),
      "fp32 %.3f ms" % t(lambda: ops.conv3d(x, w, None, None, relu=True, packed=pk, impl=ops.IMPL_MFMA, in_c8=True)),
      "split %.3f ms" % t(lambda: ops.conv3d_c8_split(x, pks, None, None, relu=True)), flush=True)
