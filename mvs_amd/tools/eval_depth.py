#!/usr/bin/env python3
"""Depth-map stage of the reference's eval.py (MVSNet/eval.py:95-131) on the MI355X path:

    python -m mvs_amd.tools.eval_depth --testpath DTU/ --testlist lists/dtu/test.txt \\
        --loadckpt model_000014.ckpt --outdir outputs

reads each scan's pair.txt / cams / images, runs MVSNet (refine=False) on cuda:0 and writes
{outdir}/{scan}/depth_est/{ref:08d}.pfm and .../confidence/{ref:08d}.pfm.  The filter /
fusion stage after it (eval.py:136-342) is `python -m mvs_amd.tools.fuse_depth`.
"""
import argparse
import os

import numpy as np
import torch

from ..datasets import find_dataset_def, save_pfm, save_pfm_rows_bottom_up
from ..models import MVSNet, load_reference_checkpoint


def _write(outdir, names, depth, conf):
    for name, d, c in zip(names, depth, conf):
        for kind, arr in (("depth_est", d), ("confidence", c)):
            path = os.path.join(outdir, name.format(kind, ".pfm"))
            os.makedirs(os.path.dirname(path), exist_ok=True)
            save_pfm(path, arr)


def save_depth(args):
    model = MVSNet(refine=False)
    if args.loadckpt:
        load_reference_checkpoint(model, torch.load(args.loadckpt, map_location="cpu"))
    model = model.cuda().eval()
    if args.device_pipeline:
        # images decoded once per scan, uploaded as uint8, normalised / cropped / transposed and the
        # projection matrices composed on the GPU; PFM writing on a worker thread (datasets/device_pipeline.py)
        from concurrent.futures import ThreadPoolExecutor
        from ..datasets.device_pipeline import DeviceScanPipeline
        pipe = DeviceScanPipeline(args.testpath, args.testlist, args.nviews, args.numdepth, args.interval_scale,
                                  decode_workers=args.decode_workers or None)
        writer, ring, RING = ThreadPoolExecutor(max_workers=2), [], 8

        def flush(slot):
            # The worker threads share the interpreter lock with the thread that launches the kernels, and whatever
            # they do under it is GPU idle time (scripts/pipe_probe.py: 9.2 ms per depth map with the copies and
            # flips of save_pfm here and 32 decoder threads, 7.4 with neither, the model alone 7.5).  So the rows are
            # flipped on the GPU and the file is the pinned buffer itself.
            ev, buf, names = slot
            ev.synchronize()
            if os.environ.get("MVS_EVAL_NO_WRITE") == "1":     # (scripts/pipe_probe.py: the driver without its file writes)
                return
            for b, name in enumerate(names):
                for k, kind in enumerate(("depth_est", "confidence")):
                    path = os.path.join(args.outdir, name.format(kind, ".pfm"))
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    save_pfm_rows_bottom_up(path, buf[k][b].numpy())

        pending = []
        scan_imgs, scan_feats = None, None
        with torch.no_grad():
            for it, sample in enumerate(pipe):
                feats = None
                if args.feature_cache:
                    # FeatureNet once per image of a scan instead of once per (sample, view): every image is the
                    # reference view of one sample and a source view of about nviews - 1 others
                    if sample["scan_imgs"] is not scan_imgs:
                        scan_imgs = sample["scan_imgs"]
                        scan_feats = model.extract_features(scan_imgs)
                    if scan_feats is not None:
                        feats = scan_feats.index_select(0, sample["view_slots"]).unsqueeze(0)
                # with cached features forward() only needs the SHAPE of the images: a stride-0 view of one resident image
                imgs = sample["imgs"] if feats is None else \
                    sample["scan_imgs"][:1].unsqueeze(0).expand(1, sample["view_slots"].numel(), -1, -1, -1)
                out = model(imgs, sample["proj_matrices"], sample["depth_values"], features=feats)
                # results leave through a ring of pinned buffers: the launching thread never waits
                # for the GPU, it runs whole reference views ahead of it
                if len(ring) < RING:
                    ring.append(torch.empty((2,) + tuple(out["depth"].shape), dtype=torch.float32).pin_memory())
                    buf = ring[-1]
                else:
                    pending[it - RING].result()          # that slot's file has been written
                    buf = ring[it % RING]
                buf[0].copy_(out["depth"].flip(-2), non_blocking=True)                     # PFM stores the bottom row first
                buf[1].copy_(out["photometric_confidence"].flip(-2), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending.append(writer.submit(flush, (ev, buf, sample["filename"])))
                if not args.quiet:
                    print(f"Iter {it}/{len(pipe)}")
        for p in pending:
            p.result()
        return
    dataset = find_dataset_def(args.dataset)(args.testpath, args.testlist, "test", args.nviews, args.numdepth,
                                             args.interval_scale)
    loader = torch.utils.data.DataLoader(dataset, args.batch_size, shuffle=False, num_workers=args.num_workers,
                                         drop_last=False)
    with torch.no_grad():
        for it, sample in enumerate(loader):
            out = model(sample["imgs"].cuda(non_blocking=True), sample["proj_matrices"].cuda(non_blocking=True),
                        sample["depth_values"].cuda(non_blocking=True))
            depth = out["depth"].cpu().numpy().astype(np.float32)
            conf = out["photometric_confidence"].cpu().numpy().astype(np.float32)
            if not args.quiet:
                print(f"Iter {it}/{len(loader)}")
            _write(args.outdir, sample["filename"], depth, conf)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Predict depth + confidence maps (MVSNet/eval.py save_depth)")
    ap.add_argument("--dataset", default="dtu_yao_eval")
    ap.add_argument("--testpath", required=True)
    ap.add_argument("--testlist", required=True)
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--numdepth", type=int, default=192)
    ap.add_argument("--interval_scale", type=float, default=1.06)
    ap.add_argument("--nviews", type=int, default=5)
    ap.add_argument("--num_workers", type=int, default=4)
    ap.add_argument("--loadckpt", default=None)
    ap.add_argument("--outdir", default="./outputs")
    ap.add_argument("--device_pipeline", action="store_true",
                    help="decode each image once per scan, normalise / crop / transpose on the GPU (batch size 1)")
    ap.add_argument("--decode_workers", type=int, default=4,
                    help="JPEG decoder threads of the device pipeline (a scan's 49 images take ~0.15 s on 4; more threads "
                         "only take the interpreter lock away from the launching thread)")
    ap.add_argument("--no_feature_cache", dest="feature_cache", action="store_false",
                    help="device pipeline: run FeatureNet on the views of every sample as the reference does, instead of "
                         "once per image of a scan (same bits either way)")
    ap.add_argument("--quiet", action="store_true")
    save_depth(ap.parse_args(argv))


if __name__ == "__main__":
    main()
