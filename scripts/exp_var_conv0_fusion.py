#!/usr/bin/env python3
"""Evidence for the variance -> conv0 fusion decision:
  (a) conv0 on a 16-plane slab whose blocked input (242 MB) stays in the 256 MB Infinity Cache vs on the
      full 192-plane volume (2.9 GB from HBM): is conv0 limited by where its input comes from?
  (b) the sweep kernel and conv0 of DIFFERENT reference views issued on two streams: how much of the
      sweep's VALU/LDS time hides under conv0's MFMA time when both are resident on the chip together
      (the upper bound of what co-scheduling without sharing LDS tiles can give).
python scripts/exp_var_conv0_fusion.py"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")


def ev_time(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return round(min(ts), 4)


def main():
    D, h, w, V = 192, 296, 400, 5
    g = torch.Generator(device=dev).manual_seed(0)
    wt = torch.randn((8, 32, 3, 3, 3), device=dev, generator=g) * 0.05
    sc = torch.rand(8, device=dev, generator=g) + 0.5
    sh = torch.randn(8, device=dev, generator=g) * 0.1
    pk = ops.pack_conv3d_weight(wt, False, 1)
    res = {}
    x_full = torch.randn(1, D, h, 4, w, 8, device=dev, generator=g)
    x_slab = x_full[:, :16].contiguous()
    conv = lambda x: ops.conv3d(x, wt, sc, sh, None, True, False, 1, packed=pk, impl=ops.IMPL_MFMA, in_c8=True)   # noqa: E731
    t_full = ev_time(lambda: conv(x_full))
    t_slab = ev_time(lambda: [conv(x_slab) for _ in range(12)])      # the same 16 planes 12 times: cache-resident input
    res["conv0_full_volume_ms"] = t_full
    res["conv0_12x_cache_resident_16_plane_slab_ms"] = t_slab
    res["conv0_slab_note"] = ("12 x 16 planes = the FLOPs of 192 planes minus the slab's z-halo planes' share; input from the "
                              "Infinity Cache instead of HBM")
    feats = torch.randn(V, 1, 32, h, w, device=dev, generator=g)
    f4 = ops.nchw_to_c4(feats)
    proj = torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D)).to(dev)
    rts = ops.rot_trans_all(proj)
    var = lambda: ops.costvol_variance_c16(f4[0], f4[1:], rts, dv, out_c8=True, fast=True)   # noqa: E731
    t_var = ev_time(var)
    res["sweep_ms"] = t_var
    res["sum_serial_ms"] = round(t_var + t_full, 4)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        e = torch.cuda.Event(); e.record()
        s1.wait_event(e); s2.wait_event(e)
        with torch.cuda.stream(s1):
            var()
        with torch.cuda.stream(s2):
            conv(x_full)
        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
        e1.record(s1); e2.record(s2)
        torch.cuda.current_stream().wait_event(e1); torch.cuda.current_stream().wait_event(e2)

    res["two_streams_concurrent_ms"] = ev_time(both)
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, "gpurun_out", "exp_var_conv0_fusion.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
