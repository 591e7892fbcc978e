"""Three Adam steps on the fixed batch of g27 under the training path's switches: losses and parameter distances to the reference's
float32 and float64 trajectories."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import config_cases as cc
from fullsize_cases import GOLDEN, _dev
from mvs_amd.models import MVSNet, mvsnet_loss

def run(**attrs):
    dev = torch.device("cuda:0")
    c = cc.train_case()
    g = dict(np.load(os.path.join(GOLDEN, "g27_train_3steps.npz")))
    model = MVSNet(refine=False); model.load_state_dict(c["sd"]); model = model.to(dev).train()
    for k, v in attrs.items():
        setattr(model, k, v)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    imgs, proj, dv, gt, mask = (_dev(c[k], dev) for k in ("imgs", "proj", "depth_values", "gt", "mask"))
    losses = []
    for _ in range(3):
        opt.zero_grad(); out = model(imgs, proj, dv); loss = mvsnet_loss(out["depth"], gt, mask); loss.backward(); opt.step(); losses.append(float(loss.detach()))
    sd = {k: v.detach().double().cpu().numpy() for k, v in model.state_dict().items()}
    rms = lambda a: float(np.sqrt((a ** 2).mean()))
    rep = {"losses": losses}
    for k in (x[7:] for x in g if x.startswith("param__")):
        p32, p64 = g["param__" + k].astype(np.float64), g["param64__" + k]
        rep[k] = {"to_ref": rms(sd[k] - p32), "to_f64": rms(sd[k] - p64), "ref_to_f64": rms(p32 - p64)}
    return rep

if __name__ == "__main__":
    out = {"ref_losses": [112.71920013, 97.49016571, 89.49617004], "f64_losses": [112.71930321, 97.53409507, 89.54069698]}
    out["default"] = run()
    out["proj_host"] = run(train_proj_where="host")
    out["torch_impl"] = run(train_impl="torch")
    print(json.dumps(out, indent=1))
