"""Times the channels-last variance backward (training shape 160x128, D=192, C=32; V from the env, default 3)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from mvs_amd import ops, synth

def run():
    dev = torch.device("cuda:0")
    H, W, V, D, C = 128, 160, int(os.environ.get("V", 3)), 192, 32
    rng = np.random.default_rng(0)
    proj = torch.from_numpy(synth.proj_matrices(V, H, W)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D)).to(dev)
    rts = ops.rot_trans_all(proj)
    ref16 = torch.from_numpy(synth.smooth_features(rng, (1, C // 16, H, W, 16))).to(dev).requires_grad_()
    srcs16 = torch.from_numpy(synth.smooth_features(rng, (V - 1, 1, C // 16, H, W, 16))).to(dev).requires_grad_()
    out = ops.costvol_variance_c16_autograd(ref16, srcs16, rts, dv)
    g = torch.randn_like(out)
    for _ in range(3):
        out.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out.backward(g, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    print(f"V={V}: {e0.elapsed_time(e1) / 10:.3f} ms per backward (incl. the two memsets)")

if __name__ == "__main__":
    run()
