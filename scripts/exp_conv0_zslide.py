"""conv0 (32 -> 8, bf16x6 split operands): the z-sliding-window kernel against the per-tile kernel of round 2
(MVS_CONV0_ZSLIDE=0) -- time at BASELINE configs[1]'s volume and bit-equality of the outputs on that and on
ragged shapes.  Each variant runs in its own process (the switch is read once per process)."""
import hashlib, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(1, 192, 296, 400, 32), (2, 37, 30, 70, 32), (1, 5, 9, 21, 16), (1, 48, 74, 100, 8), (1, 8, 296, 400, 8)]


def child():
    import torch
    from mvs_amd import ops
    out = {}
    for B, D, H, W, C in SHAPES:
        g = torch.Generator().manual_seed(D * 7 + W)
        x = (torch.randn(B, D, H, C // 8, W, 8, generator=g) * torch.rand(B, D, H, C // 8, W, 8, generator=g) ** 4).square().cuda()
        w = (torch.randn(8, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5).cuda()
        sc, sh = (torch.rand(8, generator=g) + 0.5).cuda(), (torch.randn(8, generator=g) * 0.1).cuda()
        pks = ops.pack_conv3d_weight_split(w)
        fn = lambda: ops.conv3d_c8_split(x, pks, sc, sh, None, True)
        y = fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        out[str((B, D, H, W, C))] = {"ms_min": round(min(a.elapsed_time(b) for a, b in ev), 4),
                                     "ms_med": round(sorted(a.elapsed_time(b) for a, b in ev)[3], 4),
                                     "sha": hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16],
                                     "finite": bool(torch.isfinite(y).all())}
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    res = {}
    for tag, env in (("zslide", "1"), ("per_tile", "0")):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, MVS_CONV0_ZSLIDE=env),
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-4000:]); sys.exit(1)
        res[tag] = json.loads(line[0][7:])
    for k in res["zslide"]:
        a, b = res["zslide"][k], res["per_tile"][k]
        print(k, "zslide %.3f ms (med %.3f)  per-tile %.3f ms (med %.3f)  bit-equal: %s" %
              (a["ms_min"], a["ms_med"], b["ms_min"], b["ms_med"], a["sha"] == b["sha"] and a["finite"]))
    print(json.dumps(res))
