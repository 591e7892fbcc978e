#!/usr/bin/env python3
"""End-to-end depth-maps/s of the eval driver on a synthetic DTU-format scan (49 viewpoints of
1600x1200 JPEGs, 10 sources listed per view, nviews = 5, D = 192), host loader vs device pipeline,
beside the MVSNet.forward-only rate of bench.py.   python scripts/bench_pipeline.py [nviews_in_scan]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
from io_fixture import cam_text  # noqa: E402
from mvs_amd import synth  # noqa: E402
from mvs_amd.tools import eval_depth  # noqa: E402


def build(root, nview, scan="scan9", listed=True):
    from PIL import Image
    os.makedirs(os.path.join(root, scan, "cams"))
    os.makedirs(os.path.join(root, scan, "images"))
    rng = np.random.default_rng(0)
    base = (rng.random((150, 200, 3)) * 255).astype(np.uint8)
    for v in range(nview):
        with open(os.path.join(root, scan, "cams", f"{v:0>8}_cam.txt"), "w") as f:
            f.write(cam_text(v % 3))
        img = np.asarray(Image.fromarray(np.roll(base, v * 3, 1)).resize((1600, 1200), Image.BICUBIC))
        Image.fromarray(img).save(os.path.join(root, scan, "images", f"{v:0>8}.jpg"), quality=92)
    with open(os.path.join(root, scan, "pair.txt"), "w") as f:
        f.write(f"{nview}\n")
        for v in range(nview):
            src = [(v + k) % nview for k in range(1, 11)]
            f.write(f"{v}\n10 " + " ".join(f"{s} {1000 - k:.2f}" for k, s in enumerate(src)) + " \n")
    with open(os.path.join(root, "test.txt"), "a") as f:
        f.write(scan + "\n")
    return os.path.join(root, "test.txt")


def main():
    nview = int(sys.argv[1]) if len(sys.argv) > 1 else 49
    nscan = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    root = tempfile.mkdtemp(prefix="dtu_synth_")
    for i in range(nscan):
        lf = build(root, nview, scan=f"scan{i + 1}")
    ckpt = os.path.join(root, "m.ckpt")
    torch.save({"model": synth.random_state_dict(0)}, ckpt)
    common = ["--testpath", root, "--testlist", lf, "--loadckpt", ckpt, "--nviews", "5", "--numdepth", "192", "--quiet"]
    res = {"views_in_scan": nview, "scans": nscan}
    nview = nview * nscan
    for name, extra in (("device_pipeline", ["--device_pipeline"]), ("host_loader_4_workers", ["--num_workers", "4"]),
                        ("device_pipeline_again", ["--device_pipeline"]),
                        ("device_pipeline_featurenet_per_sample", ["--device_pipeline", "--no_feature_cache"])):
        out = os.path.join(root, "out_" + name)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eval_depth.main(common + ["--outdir", out] + extra)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"seconds": round(dt, 2), "depth_maps_per_s": round(nview / dt, 2)}
        print(name, res[name], flush=True)
    # FeatureNet once per image vs once per (sample, view): the files are the same bytes
    a, b = os.path.join(root, "out_device_pipeline_again"), os.path.join(root, "out_device_pipeline_featurenet_per_sample")
    same = 0
    for dp, _, fs in os.walk(a):
        for fn in fs:
            pa = os.path.join(dp, fn)
            with open(pa, "rb") as fa, open(os.path.join(b, os.path.relpath(pa, a)), "rb") as fb:
                assert fa.read() == fb.read(), pa
            same += 1
    res["files_identical_with_and_without_feature_cache"] = same
    print(json.dumps(res))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "bench_pipeline.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
