"""FeatureNet on the HIP kernels only, a few forwards at config-2 size (a target for scripts/prof_kernel_sq.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import synth
from mvs_amd.models import MVSNet
m = MVSNet(refine=False); m.load_state_dict(synth.random_state_dict(0)); m = m.cuda().eval()
x = torch.rand(5, 3, 1184, 1600, device="cuda")
with torch.no_grad():
    for _ in range(3):
        m.feature.forward_hip(x, out_c4=True)
torch.cuda.synchronize()
