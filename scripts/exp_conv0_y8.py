"""conv0 with eight-row tiles (csrc/conv_f16x3_y8.hip) against the committed kernel (csrc/conv_f16x3.hip).

    python scripts/exp_conv0_y8.py            # every variant in its own process: bits, times, tuning-build ablations

MVS_CONV0_Y8 = 0 (committed (4,4,32) tiles), 2 (two barriers per step), 1 (one barrier per step) is read once per
process, so each variant runs as a child; outputs on seven shapes (ragged, batch 2, Cin 8 / 16 / 32, residual) are
hashed and compared -- the kernels share the MFMA order per accumulator, so they must agree bit for bit.  Times at
BASELINE configs[1]'s volume (1 x 32 x 192 x 296 x 400).  With the tuning build (MVS_CONV_SPLIT_ABL): 1 = no split
work, 2 = no MFMA phase, 3 = copies and barriers only (VERDICT r05 item 1's kill criterion: <= 0.95 ms)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(1, 12, 20, 70, 32), (2, 7, 9, 37, 16), (1, 5, 30, 33, 8), (1, 33, 17, 64, 32), (1, 2, 8, 32, 32), (1, 1, 3, 5, 8),
          (2, 19, 41, 100, 32)]


def child(tag):
    import torch
    from mvs_amd import ops

    def data(B, D, H, W, C, seed):
        g = torch.Generator().manual_seed(seed)
        x = (torch.randn(B, D, H, C // 8, W, 8, generator=g) * torch.rand(B, D, H, C // 8, W, 8, generator=g) ** 4).square()
        w = torch.randn(8, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5
        return x, w, torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1

    res = {"hash": {}, "time": {}}
    abl = int(os.environ.get("MVS_CONV_SPLIT_ABL", "0"))
    if not abl:
        for i, (B, D, H, W, C) in enumerate(SHAPES):
            x, w, sc, sh = data(B, D, H, W, C, 100 + i)
            xd = x.cuda()
            resid = torch.randn(B, D, H, W, 8, generator=torch.Generator().manual_seed(i)).cuda() if i % 2 else None
            omx = ops.absmax_block(xd.device, zero=True)
            y = ops.conv3d_c8_f16x3(xd, ops.pack_conv3d_weight_f16x3(w.cuda()), None, sc.cuda(), sh.cuda(), resid, i != 2, out_absmax=omx)
            torch.cuda.synchronize()
            res["hash"][str((B, D, H, W, C))] = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16] + ":" + str(int(omx.max()))
    _, w, sc, sh = data(1, 1, 1, 1, 32, 5)
    gg = torch.Generator(device="cuda").manual_seed(5)       # the full-size volume is drawn on the device (same bits in every child)
    xd = (torch.randn(1, 192, 296, 4, 400, 8, generator=gg, device="cuda") * torch.rand(1, 192, 296, 4, 400, 8, generator=gg, device="cuda") ** 4).square()
    wd, sc, sh = w.cuda(), sc.cuda(), sh.cuda()
    pf = ops.pack_conv3d_weight_f16x3(wd)
    mx = ops.absmax(xd)
    fn = lambda: ops.conv3d_c8_f16x3(xd, pf, mx, sc, sh, None, True)
    for rep in range(2):
        fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(11)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        res["time"]["min"] = round(t[0], 4)
        res["time"]["med"] = round(t[5], 4)
    if not abl:
        res["hash"]["full"] = hashlib.sha1(fn().cpu().numpy().tobytes()).hexdigest()[:16]
    print("RESULT " + json.dumps(res), flush=True)


def laps():
    """tuning build, MVS_CONV_SPLIT_ABL=128: clock64() laps of the multiplying waves (cycles per step, mean / slowest / fastest wave)"""
    import torch
    from mvs_amd import ops
    D, H, W = 192, 296, 400
    x = torch.randn(1, D, H, 4, W, 8, device="cuda").square()
    w = torch.randn(8, 32, 3, 3, 3, device="cuda") * 0.1
    pf = ops.pack_conv3d_weight_f16x3(w)
    dbg = torch.zeros(256 * 8 * 8 * 2, device="cuda")     # int64 [256 workgroups][8 waves][8]
    ops.conv3d_c8_f16x3(x, pf, None, None, None, dbg, relu=True)
    torch.cuda.synchronize()
    tt = dbg.view(torch.int64).view(256, 8, 8).double()
    y8 = os.environ.get("MVS_CONV0_Y8", "0")
    if y8 == "0":
        names = ["-", "barrier 1", "split pass", "barrier 2", "-", "MFMA phase", "rest"]
        steps = -(-D // 4) * -(-H // 4) * -(-W // 32) * 4 / 256
    else:
        names = ["barrier 1", "split pass", "barrier 2", "MFMA phase", "rest"]
        steps = -(-D // 16) * 9 * -(-H // 8) * -(-W // 32) * 4 / 256
    n = len(names)
    out = {"steps_per_workgroup": steps, "cycles_per_step": round(tt[:, :, :n].sum(-1).mean().item() / steps),
           "phases_per_step": {nm: [round(tt[:, :, k].mean().item() / steps), round(tt[:, :, k].max(1).values.mean().item() / steps),
                                    round(tt[:, :, k].min(1).values.mean().item() / steps)] for k, nm in enumerate(names) if nm != "-"}}
    print("RESULT " + json.dumps(out), flush=True)


def pairs_child():
    """tuning build: the hand-over kernel (csrc/conv_f16x3_y8p.hip) on pairs made by mvs_c8_to_c8p_f32 -- hashes on the test shapes
    (same seeds as child(): must equal the staged kernels'), time at configs[1]'s volume; MVS_CONV_SPLIT_ABL = 2: copies only."""
    import ctypes
    import torch
    from mvs_amd import ops, _lib
    lib = _lib.load()
    layout, npair = int(os.environ["MVS_EXP_LAYOUT"]), int(os.environ["MVS_EXP_NPAIR"])
    abl = int(os.environ.get("MVS_CONV_SPLIT_ABL", "0"))
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def to_pairs(x, mx):
        B, D, H, G, W, _ = x.shape
        out = torch.zeros((lib.mvs_c8p_bytes(B, G * 8, D, H, W, layout) + 1) // 2, device=x.device, dtype=torch.int16)
        _lib.check(lib.mvs_c8_to_c8p_f32(vp(x), vp(mx), B, G * 8, D, H, W, layout, vp(out), st()), "mvs_c8_to_c8p_f32")
        return out

    def conv(xp, mx, pk, sc, sh, resid, relu, shape, omx=None):
        B, D, H, G, W, _ = shape
        out = torch.empty(B, D, H, W, 8, device=xp.device, dtype=torch.float32)
        _lib.check(lib.mvs_conv3d_c8p_f16x3_f32(vp(xp), vp(mx), None, vp(pk), vp(sc), vp(sh), vp(resid), int(relu), B, G * 8, D, H, W, layout, npair,
                                                vp(out), vp(omx), st()), "mvs_conv3d_c8p_f16x3_f32")
        return out

    def data(B, D, H, W, C, seed):
        g = torch.Generator().manual_seed(seed)
        x = (torch.randn(B, D, H, C // 8, W, 8, generator=g) * torch.rand(B, D, H, C // 8, W, 8, generator=g) ** 4).square()
        w = torch.randn(8, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5
        return x, w, torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1

    res = {"hash": {}, "time": {}}
    if not abl:
        for i, (B, D, H, W, C) in enumerate(SHAPES):
            x, w, sc, sh = data(B, D, H, W, C, 100 + i)
            xd = x.cuda()
            resid = torch.randn(B, D, H, W, 8, generator=torch.Generator().manual_seed(i)).cuda() if i % 2 else None
            omx = ops.absmax_block(xd.device, zero=True)
            mx = ops.absmax(xd)
            y = conv(to_pairs(xd, mx), mx, ops.pack_conv3d_weight_f16x3(w.cuda()), sc.cuda(), sh.cuda(), resid, i != 2, xd.shape, omx)
            torch.cuda.synchronize()
            res["hash"][str((B, D, H, W, C))] = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16] + ":" + str(int(omx.max()))
    _, w, sc, sh = data(1, 1, 1, 1, 32, 5)
    gg = torch.Generator(device="cuda").manual_seed(5)
    xd = (torch.randn(1, 192, 296, 4, 400, 8, generator=gg, device="cuda") * torch.rand(1, 192, 296, 4, 400, 8, generator=gg, device="cuda") ** 4).square()
    wd, sc, sh = w.cuda(), sc.cuda(), sh.cuda()
    pf = ops.pack_conv3d_weight_f16x3(wd)
    mx = ops.absmax(xd)
    xp = to_pairs(xd, mx)
    shape = xd.shape
    del xd
    if abl & 128:
        dbg = torch.zeros(256 * 8 * 8 * 2, device="cuda")
        conv(xp, mx, pf, sc, sh, dbg, True, shape)
        torch.cuda.synchronize()
        tt = dbg.view(torch.int64).view(256, 8, 8).double()
        steps = 12 * 9 * 37 * 13 * 4 / 256
        names = ["barrier", "MFMA phase", "rest"]
        res = {"cycles_per_step": round(tt[:, :, :3].sum(-1).mean().item() / steps),
               "phases_per_step": {nm: [round(tt[:, :, k].mean().item() / steps), round(tt[:, :, k].max(1).values.mean().item() / steps),
                                        round(tt[:, :, k].min(1).values.mean().item() / steps)] for k, nm in enumerate(names)}}
        print("RESULT " + json.dumps(res), flush=True)
        return
    fn = lambda: conv(xp, mx, pf, sc, sh, None, True, shape)
    for rep in range(2):
        fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(11)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        res["time"]["min"] = round(t[0], 4)
        res["time"]["med"] = round(t[5], 4)
    if not abl:
        res["hash"]["full"] = hashlib.sha1(fn().cpu().numpy().tobytes()).hexdigest()[:16]
    print("RESULT " + json.dumps(res), flush=True)


def run(env_extra, what="child"):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), what], env=env, capture_output=True, text=True, timeout=900)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    return {"error": (r.stdout + r.stderr)[-1500:]}


def main():
    out = {}
    for y8 in ("0", "2", "1"):
        out["y8=" + y8] = run({"MVS_CONV0_Y8": y8})
        print("y8=" + y8, json.dumps(out["y8=" + y8]), flush=True)
    base = out["y8=0"].get("hash")
    for y8 in ("2", "1"):
        h = out["y8=" + y8].get("hash")
        out["bit_identical_y8=" + y8] = bool(base) and h == base
        print("bit-identical to the committed kernel, y8 =", y8, ":", out["bit_identical_y8=" + y8], flush=True)
    if os.path.exists(os.path.join(ROOT, "mvs_amd", "csrc", "libmvs_hip_tuning.so")):
        for y8 in (("0", "2", "1") if os.environ.get("MVS_EXP_STAGED_ABL") else ()):
            for abl in ("1", "2", "3"):
                k = f"abl{abl}_y8={y8}"
                out[k] = run({"MVS_CONV0_Y8": y8, "MVS_HIP_TUNING": "1", "MVS_CONV_SPLIT_ABL": abl}).get("time")
                print(k, out[k], flush=True)
            out["laps_y8=" + y8] = run({"MVS_CONV0_Y8": y8, "MVS_HIP_TUNING": "1", "MVS_CONV_SPLIT_ABL": "128"}, "laps")
            print("laps y8=" + y8, json.dumps(out["laps_y8=" + y8]), flush=True)
        for layout in ("6", "7", "8"):
            for npair in ("4", "5"):
                env = {"MVS_HIP_TUNING": "1", "MVS_EXP_LAYOUT": layout, "MVS_EXP_NPAIR": npair}
                k = f"pairs_layout{layout}_npair{npair}"
                out[k] = run(env, "pairs")
                out[k + "_bit_identical"] = bool(base) and out[k].get("hash") == base
                print(k, json.dumps(out[k]), "bit-identical:", out[k + "_bit_identical"], flush=True)
                out[k + "_copies_only"] = run(dict(env, MVS_CONV_SPLIT_ABL="2"), "pairs").get("time")
                print(k, "copies only", out[k + "_copies_only"], flush=True)
                out[k + "_laps"] = run(dict(env, MVS_CONV_SPLIT_ABL="128"), "pairs")
                print(k, "laps", json.dumps(out[k + "_laps"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "conv0_y8.json"), "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(os.environ.get("MVS_CONV0_Y8", "0"))
    elif len(sys.argv) > 1 and sys.argv[1] == "laps":
        laps()
    elif len(sys.argv) > 1 and sys.argv[1] == "pairs":
        pairs_child()
    else:
        main()
