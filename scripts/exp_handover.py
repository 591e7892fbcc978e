"""The sweep with its volume handed over as fp16 pieces (mvs_costvol_variance_fwd_ws3_f32) against the fp32 sweep, at BASELINE
configs[1]'s shape (V = 5, 32 channels, 296 x 400, 192 planes): what the pieces cost the producer.

    python scripts/exp_handover.py            # children: fp32, pieces (rows + halo strips), and with the tuning build: the other layouts, no halo copies, no stores

Each variant in its own process (the switches are read once)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from mvs_amd import ops, synth
    dev = torch.device("cuda:0")
    V, h, w, D = 5, 296, 400, 192
    g = torch.Generator(device=dev).manual_seed(1)
    f = torch.randn(V, 1, 8, h, w, 4, device=dev, generator=g)
    rts = ops.rot_trans_all(torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev), "device")
    dv = torch.from_numpy(synth.depth_values(D)).to(dev)
    fa = ops.absmax(f)
    mode = os.environ.get("EXP_MODE", "fp32")
    blk = ops.absmax_block(dev)
    if mode == "fp32":
        fn = lambda: ops.costvol_variance_c16(f[0], f[1:], rts, dv, out_c8=True, fast=True, absmax_out=blk)
    else:
        fn = lambda: ops.costvol_variance_handover(f[0], f[1:], rts, dv, fa, fast=True)
    res = {}
    for rep in range(2):
        out = fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(11)]
        for a, b in ev:
            a.record(); out = fn(); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        res = {"min": round(t[0], 4), "med": round(t[5], 4)}
    if mode != "fp32":
        res["redo"] = int(out.redo[0].item())
        res["loose_bits"] = (int(out.hand.max().item()) >> 23) - (int(out.absmax.max().item()) >> 23)
    print("RESULT " + json.dumps(res), flush=True)


def run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    return {"error": (r.stdout + r.stderr)[-1500:]}


def main():
    out = {"fp32": run({"EXP_MODE": "fp32"}), "pieces_strips": run({"EXP_MODE": "hand"})}
    if os.path.exists(os.path.join(ROOT, "mvs_amd", "csrc", "libmvs_hip_tuning.so")):
        t = {"MVS_HIP_TUNING": "1", "EXP_MODE": "hand"}
        out["tuning_fp32"] = run({"MVS_HIP_TUNING": "1", "EXP_MODE": "fp32"})
        out["tuning_pieces_strips"] = run(t)
        out["tuning_pieces_rows"] = run(dict(t, MVS_HANDOVER_LAYOUT="6"))
        out["tuning_pieces_xtiled"] = run(dict(t, MVS_HANDOVER_LAYOUT="7"))
        out["tuning_pieces_strips_no_halo_copies"] = run(dict(t, MVS_HANDOVER_FLAGS="128"))
        out["tuning_pieces_strips_no_stores"] = run(dict(t, MVS_HANDOVER_FLAGS="2"))
    for k, v in out.items():
        print(k, json.dumps(v), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "handover_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "child" else main()
