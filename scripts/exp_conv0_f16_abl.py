"""conv0 fp16 kernel with parts of a step removed (tuning build, wrong results): what the copies and barriers alone cost."""
import os as _os; _os.environ["MVS_HIP_TUNING"] = "1"
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops
x = torch.randn(1, 192, 296, 4, 400, 8, device="cuda").square()
w = torch.randn(8, 32, 3, 3, 3, device="cuda") * 0.1
pf = ops.pack_conv3d_weight_f16x3(w)
mx = ops.absmax(x)
fn = lambda: ops.conv3d_c8_f16x3(x, pf, mx, None, None, None, relu=True)
fn(); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
for a, b in ev:
    a.record(); fn(); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev)
print("ABL", os.environ.get("MVS_CONV_SPLIT_ABL", "0"), "min %.3f med %.3f ms" % (t[0], t[3]))
