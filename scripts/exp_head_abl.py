import os, sys, torch
sys.path.insert(0, '/root/repo')
from mvs_amd import synth, ops
from mvs_amd.models import MVSNet
dev = torch.device("cuda:0")
m = MVSNet(refine=False); m.load_state_dict(synth.random_state_dict(0)); m = m.to(dev).eval()
x = torch.rand(5, 3, 1184, 1600, device=dev)
with torch.no_grad():
    for _ in range(3): m.feature.forward_hip(x)
    tm = ops.StageTimer(only={"feature.head"}); ops.set_timer(tm)
    for _ in range(10): m.feature.forward_hip(x)
    torch.cuda.synchronize(); ops.set_timer(None)
    print(os.environ.get("MVS_HEAD_ABL"), round(tm.min_ms()["feature.head"], 4))
