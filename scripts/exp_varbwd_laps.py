#!/usr/bin/env python3
"""The training path's variance backward at configs[4]'s shape (B 1, V 3, C 32, D 192, 128 x 160): time per launch (HIP events) and,
in the tuning build (MVS_HIP_TUNING=1, python -m mvs_amd.build --tuning), cycles of wave 0 per phase (mvs_tuning_varbwd_laps)."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops, synth, _lib
dev = torch.device("cuda:0")
B, V, C, D, H, W = 1, 3, 32, 192, 128, 160
rng = np.random.default_rng(0)
proj = torch.from_numpy(synth.proj_matrices(V, H, W, batch=B)).to(dev)
dv = torch.from_numpy(synth.depth_values(D, batch=B)).to(dev)
f16 = torch.from_numpy(synth.smooth_features(rng, (V, B, C // 16, H, W, 16))).to(dev).requires_grad_(True)
rts = ops.rot_trans_all(proj)
var = ops.costvol_variance_c16_autograd(f16[0], f16[1:], rts, dv)
go = torch.randn_like(var)
def bwd():
    f16.grad = None
    var.backward(go, retain_graph=True)
for _ in range(3): bwd()
torch.cuda.synchronize()
lib = _lib.load()
tuning = _lib.tuning_build_loaded()
if tuning:
    lib.mvs_tuning_varbwd_laps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.mvs_tuning_varbwd_laps(None, 1)
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): bwd()
e1.record(); torch.cuda.synchronize()
out = {"ms_per_backward_op": round(e0.elapsed_time(e1) / n, 4), "tuning": tuning, "grad_checksum": float(f16.grad.double().abs().sum())}
if tuning:
    buf = (ctypes.c_ulonglong * 8)()
    lib.mvs_tuning_varbwd_laps(buf, 0)
    names = ["setup", "clear+staging", "bound", "accumulate", "flush"]
    blocks = (D // 4) * (H // 8) * (W // 8) * n
    out["cycles_per_block"] = {nm: round(buf[i] / blocks) for i, nm in enumerate(names)}
    out["passes_per_block"] = buf[5] / blocks
print(json.dumps(out, indent=1))
