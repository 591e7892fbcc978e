"""Device-side kernel choice of the fused warp+variance sweep (mvs_costvol_variance_fwd_ws_f32): for depth sweeps of
192 planes at interval scales x1 ... x4 (and CasMVSNet's 48 planes at x4), the time of the automatic choice against each
forced kernel (MVS_SWEEP_PERSIST = 16 | 8 | 0), the footprint statistics the choice was made from, and bit-equality of
the variance volumes (EXACT coordinates).  python scripts/exp_sweep_select.py [--json out.json]"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import ops, synth


def timeit(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)


def run(D, scale, h=296, w=400, V=5, fast=True, rig=0):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    feats = torch.from_numpy(synth.smooth_features(rng, (V, 1, 32, h, w))).to(dev)
    proj = torch.from_numpy(synth.proj_matrices(V, h, w, rig=rig)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D, interval=synth.DTU_INTERVAL * scale)).to(dev)
    rts = ops.rot_trans_all(proj, "device")
    c4 = ops.nchw_to_c4(feats)            # [V,1,8,h,w,4]
    res = {}
    outs = {}
    for tag, env in (("16", "16"), ("8", "8"), ("tile", "0"), ("auto", None), ("auto2", None)):
        if env is None:
            os.environ.pop("MVS_SWEEP_PERSIST", None)
        else:
            os.environ["MVS_SWEEP_PERSIST"] = env
        if tag == "tile":    # the per-tile kernel reads 16-channel blocks
            c16 = ops.nchw_to_c16(feats)
            fn = lambda: ops.costvol_variance_c16(c16[0], c16[1:], rts, dv, out_c8=True, fast=False)
            fx = fn
        else:
            fn = lambda: ops.costvol_variance_c16(c4[0], c4[1:], rts, dv, out_c8=True, fast=fast)
            fx = lambda: ops.costvol_variance_c16(c4[0], c4[1:], rts, dv, out_c8=True, fast=False)
        res[tag] = round(timeit(fn), 4)
        outs[tag] = fx().clone()
        if tag == "auto2":
            res["auto"] = min(res["auto"], res.pop("auto2"))
            outs.pop("auto2")
        if tag == "auto":
            torch.cuda.synchronize()
            ws = next(iter(ops._variance_ws.values()))
            hdr = ws[:32].view(torch.int32).cpu().tolist()
            res["choice"] = hdr[1]
            res["box16_max_mean"] = hdr[2:4]
            res["box8_max_mean"] = hdr[4:6]
            res["cap"] = hdr[7]
    os.environ.pop("MVS_SWEEP_PERSIST", None)
    res["bit_equal_exact"] = all(torch.equal(outs["auto"], outs[k]) for k in ("16", "8", "tile"))
    best = min(res[k] for k in ("16", "8", "tile"))
    res["auto_over_best"] = round(res["auto"] / best, 3)
    return res


if __name__ == "__main__":
    table = {}
    for D, scale in ((192, 1.0), (192, 1.5), (192, 2.0), (192, 3.0), (192, 4.0), (48, 4.0), (96, 2.0)):
        for rig in (0, 1):
            r = run(D, scale, rig=rig)
            table[f"D={D} x{scale} rig{rig}"] = r
            print(f"D={D} x{scale} rig{rig}", json.dumps(r), flush=True)
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(table, f, indent=1)
