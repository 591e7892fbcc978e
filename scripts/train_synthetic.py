#!/usr/bin/env python3
"""Build-owned counterpart of the reference's training loop (MVSNet/train.py:204-248
`train_sample`: zero_grad -> forward(train) -> mvsnet_loss -> backward -> Adam step) on
synthetic DTU-shaped data, data-parallel over one process per GPU:

    python scripts/train_synthetic.py --steps 5
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 \
        scripts/train_synthetic.py --steps 5          # BASELINE config 5: 1 ref view per GPU

Each rank draws its own samples (shard = rank), gradients are averaged with ONE flat
fp32 all-reduce (1.35 MB) over RCCL (mvs_amd.parallel.FlatGradAllReduce), BatchNorm
statistics stay per rank as in the reference's nn.DataParallel (train.py:95).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import parallel, synth  # noqa: E402
from mvs_amd.models import MVSNet, mvsnet_loss  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--ndepth", type=int, default=192)
    ap.add_argument("--lr", type=float, default=1e-3)     # train.py:98
    args = ap.parse_args()
    # (train.py:25 sets cudnn.benchmark; on this stack it costs a 280 s MIOpen search in the
    # first step and changes nothing afterwards: 186 ms either way -- left off)
    rank, world, dev = parallel.init_distributed()
    torch.manual_seed(1)                                  # train.py:53,65-66
    model = MVSNet(refine=False).to(dev)
    parallel.broadcast_parameters(model, 0)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, betas=(0.9, 0.999), weight_decay=0.0)
    reduce_grads = parallel.FlatGradAllReduce(model.parameters())
    H, W, V, D = args.height, args.width, args.views, args.ndepth
    h, w = H // 4, W // 4
    rng = np.random.default_rng(100 + rank)
    proj = torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev)
    dvals = torch.from_numpy(synth.depth_values(D)).to(dev)
    log = []
    model.train()
    for step in range(args.steps):
        imgs = torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev)
        gt = torch.from_numpy((synth.DTU_TARGET_Z + 20 * rng.standard_normal((1, h, w)))
                              .astype(np.float32)).to(dev)
        mask = torch.ones(1, h, w, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        out = model(imgs, proj, dvals)
        loss = mvsnet_loss(out["depth"], gt, mask)
        loss.backward()
        reduce_grads()
        opt.step()
        torch.cuda.synchronize()
        sc = parallel.reduce_scalars({"loss": float(loss.item()), "ms": (time.perf_counter() - t0) * 1e3})
        if rank == 0:
            log.append({k: round(v, 4) for k, v in sc.items()})
    if rank == 0:
        print(json.dumps({"world": world, "config": f"{W}x{H} V={V} D={D}, 1 ref view per GPU",
                          "grad_floats": reduce_grads.numel, "steps": log,
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
