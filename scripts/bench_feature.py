#!/usr/bin/env python3
"""FeatureNet (PyTorch-ROCm / MIOpen) timing variants at config-2 size."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import synth
from mvs_amd.models import MVSNet

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e3

dev = torch.device("cuda:0")
m = MVSNet(refine=False); m.load_state_dict(synth.random_state_dict(0)); m = m.to(dev).eval()
x = torch.rand(5, 3, 1184, 1600, device=dev)
with torch.no_grad():
    from mvs_amd import ops
    tm = ops.StageTimer(); ops.set_timer(tm)
    print("HIP 2D MFMA kernels    ms", round(t(lambda: m.feature.forward_hip(x)), 3))
    torch.cuda.synchronize(); ops.set_timer(None)
    print({k: round(v[1], 4) for k, v in tm.summary_ms().items()})
    print("default nchw           ms", round(t(lambda: m.feature(x)), 3))
    torch.backends.cudnn.benchmark = True
    t0 = time.time(); m.feature(x); torch.cuda.synchronize(); print("benchmark first call s", round(time.time() - t0, 2))
    print("cudnn.benchmark nchw   ms", round(t(lambda: m.feature(x)), 3))
    xcl = x.contiguous(memory_format=torch.channels_last); mcl = m.feature.to(memory_format=torch.channels_last)
    t0 = time.time(); mcl(xcl); torch.cuda.synchronize(); print("cl first call s", round(time.time() - t0, 2))
    print("benchmark channels_last ms", round(t(lambda: mcl(xcl)), 3))
    torch.backends.cudnn.benchmark = False
    print("channels_last no-bench ms", round(t(lambda: mcl(xcl)), 3))
