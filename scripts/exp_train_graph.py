"""Training step of BASELINE configs[4] (640x512, V=3, D=192, one reference view) eager vs captured in a HIP graph
(torch.cuda.CUDAGraph): forward(train) -> mvsnet_loss -> backward [-> Adam] as one graph launch per step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import synth
from mvs_amd.models import MVSNet, mvsnet_loss
dev = torch.device("cuda:0")
torch.manual_seed(1)
H, W, V, D = 512, 640, 3, 192
model = MVSNet(refine=False).to(dev).train()
model.proj_where = "device"
opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
rng = np.random.default_rng(0)
proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
imgs = torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev)
gt = torch.full((1, H // 4, W // 4), 680.0, device=dev)
mask = torch.ones_like(gt)

def step():
    opt.zero_grad(set_to_none=False)
    out = model(imgs, proj, dv)
    loss = mvsnet_loss(out["depth"], gt, mask)
    loss.backward()
    opt.step()
    return loss

def timed(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): l = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, l

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
ms_e, l = timed(step)
print("eager ms/step", round(ms_e, 3), "loss", float(l.detach()), flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
if mode == "fwd":
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        out = model(imgs, proj, dv)
        static_loss = mvsnet_loss(out["depth"], gt, mask)
    print("forward captured", flush=True)
    g.replay(); torch.cuda.synchronize(); print("forward replayed", float(static_loss.detach()), flush=True)
    sys.exit(0)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="relaxed"):
    static_loss = step()
print("captured", flush=True)
def replay():
    g.replay(); return static_loss
ms_g, l = timed(replay)
print("graph ms/step", round(ms_g, 3), "loss", float(l))
