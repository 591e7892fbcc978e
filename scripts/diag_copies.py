#!/usr/bin/env python3
"""Which host-side ops of one inference step launch runtime copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer) and fills:
torch.profiler with Python stacks, grouped by the innermost mvs_amd frame."""
import os, sys, collections
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import synth
from mvs_amd.models import MVSNet
dev = torch.device("cuda:0")
H, W, V, D = 1184, 1600, 5, 192
rng = np.random.default_rng(0)
imgs = torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev)
proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
model = MVSNet(refine=False)
model.load_state_dict(synth.random_state_dict(seed=0))
model = model.to(dev).eval()
def step():
    with torch.no_grad():
        return model(imgs, proj, dv)
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
groups = collections.Counter()
for e in prof.events():
    if e.device_type.name != "CPU":
        continue
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "hipMemcpyAsync", "hipMemsetAsync", "aten::cat", "aten::stack"):
        frame = next((f for f in e.stack if "mvs_amd" in f or "bench" in f), e.stack[0] if e.stack else "?")
        groups[(e.name, str(e.input_shapes)[:80] + ' ' + frame.strip()[-60:])] += 1
for (name, frame), n in sorted(groups.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} {name:18s} {frame}")
print("device kernels:")
kc = collections.Counter()
for e in prof.key_averages():
    if e.device_time_total > 0 and e.count and ("copy" in e.key.lower() or "fill" in e.key.lower() or "Memcpy" in e.key or "Memset" in e.key):
        print(f"  {e.key[:100]:100s} n={e.count} us={e.device_time_total:.1f}")
