"""The training step of BASELINE configs[4] (and its parts) captured into a HIP graph (torch.cuda.CUDAGraph): does every
kernel of the HIP training path capture, and what does one graph launch per step buy over ~450 eager launches?
    FULL=1 python scripts/exp_train_graph.py full      (parts: feature | costreg | regress | fwdbwd | adam)
Measured (640x512, V=3, D=192): 9.41 ms per replayed step against 9.95 ms eager on a fast host -- the step is bound by its
kernels there; on a slow or busy host the eager step stretches to 12-16 ms while the replay does not."""
import os, sys, faulthandler
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops, synth
from mvs_amd.models import MVSNet, mvsnet_loss
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = MVSNet(refine=False).to(dev).train()
model.proj_where = "device"
H, W, V, D = (512, 640, 3, 192) if os.environ.get('FULL') else (128, 160, 3, 16)

def try_capture(name, fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print("capturing", name, flush=True)
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        fn()
    print("  captured", name, flush=True)
    g.replay(); torch.cuda.synchronize()
    print("  ok", name, flush=True)
    import time
    for label, f in (("graph", g.replay), ("eager", fn)):      # (no replay after eager steps: they re-allocate cached scratch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize(); print("  %s ms/step %.3f" % (label, (time.perf_counter() - t0) / 20 * 1e3), flush=True)

imgs = torch.rand(1, V, 3, H, W, device=dev)
proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
dv = torch.from_numpy(synth.depth_values(D, interval=synth.sweep_interval(D))).to(dev)
if os.environ.get('TPW'): model.train_proj_where = os.environ['TPW']
if os.environ.get('NOFUSE'): model.train_conv0_fused = False
gt = torch.full((1, H // 4, W // 4), 680.0, device=dev); mask = torch.ones_like(gt)
def zero():
    for p in model.parameters(): p.grad = None
what = sys.argv[1] if len(sys.argv) > 1 else "full"
if what == "feature":
    def fn():
        zero(); model.feature.forward_train_hip(imgs[:, 0]).sum().backward()
elif what == "costreg":
    x = torch.randn(1, D, H // 4, W // 4, 32, device=dev, requires_grad=True)
    def fn():
        zero(); x.grad = None; model.cost_regularization.forward_train_hip(x).sum().backward()
elif what == "regress":
    c = torch.randn(1, D, H // 4, W // 4, device=dev, requires_grad=True)
    def fn():
        c.grad = None
        d, _, _ = ops.softmax_regress_conf(c, dv)
        mvsnet_loss(d, gt, mask).backward()
elif what == "fwdbwd":
    def fn():
        zero(); out = model(imgs, proj, dv); mvsnet_loss(out["depth"], gt, mask).backward()
elif what == "full":
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
    def fn():
        opt.zero_grad(set_to_none=False); out = model(imgs, proj, dv); mvsnet_loss(out["depth"], gt, mask).backward(); opt.step()
elif what == "adam":
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
    out = model(imgs, proj, dv); mvsnet_loss(out["depth"], gt, mask).backward()
    def fn():
        opt.step()
try_capture(what, fn)
