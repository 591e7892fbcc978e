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

from ..datasets import find_dataset_def, save_pfm
from ..models import MVSNet, load_reference_checkpoint


def save_depth(args):
    dataset = find_dataset_def(args.dataset)(args.testpath, args.testlist, "test", args.nviews, args.numdepth,
                                             args.interval_scale)
    loader = torch.utils.data.DataLoader(dataset, args.batch_size, shuffle=False, num_workers=args.num_workers,
                                         drop_last=False)
    model = MVSNet(refine=False)
    if args.loadckpt:
        load_reference_checkpoint(model, torch.load(args.loadckpt, map_location="cpu"))
    model = model.cuda().eval()
    with torch.no_grad():
        for it, sample in enumerate(loader):
            out = model(sample["imgs"].cuda(non_blocking=True), sample["proj_matrices"].cuda(non_blocking=True),
                        sample["depth_values"].cuda(non_blocking=True))
            depth = out["depth"].cpu().numpy().astype(np.float32)
            conf = out["photometric_confidence"].cpu().numpy().astype(np.float32)
            print(f"Iter {it}/{len(loader)}")
            for name, d, c in zip(sample["filename"], depth, conf):
                for kind, arr in (("depth_est", d), ("confidence", c)):
                    path = os.path.join(args.outdir, name.format(kind, ".pfm"))
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    save_pfm(path, arr)


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
    save_depth(ap.parse_args(argv))


if __name__ == "__main__":
    main()
