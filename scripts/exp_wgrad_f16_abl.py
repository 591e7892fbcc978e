"""Times ops.conv3d_wgrad_c8_f16 (kernel + reduce + zero fill) at the training step's volume (1 x 192 x 128 x 160 voxels, 32 -> 8).
The ablation numbers in conv3d_wgrad_f16.hip came from a build whose kernel skipped MFMAs / splits / loads on MVS_WGRAD_F16_ABL bits."""
import os, sys, torch, time
sys.path.insert(0, '/root/repo')
from mvs_amd import ops
dev = torch.device('cuda:0')
B, D, H, W = 1, 192, 128, 160
x = torch.randn(B, D, H, 4, W, 8, device=dev)
g = torch.randn(B, D, H, W, 8, device=dev)
ax, ag = ops.absmax(x), ops.absmax(g)
for _ in range(3): ops.conv3d_wgrad_c8_f16(x, ax, g, ag)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.conv3d_wgrad_c8_f16(x, ax, g, ag)
e1.record(); torch.cuda.synchronize()
print(round(e0.elapsed_time(e1) / 20, 4), 'ms per call (kernel + reduce + zero fill)')
