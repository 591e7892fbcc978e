"""Worker of tests/test_gpu_multi.py::test_rccl_gradient_average_equals_mean_of_shard_gradients
(launched with torch.distributed.run, one process per GPU)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import parallel, synth  # noqa: E402
from mvs_amd.models import MVSNet, mvsnet_loss  # noqa: E402


def shard_grads(model, dev, shard, H=128, W=160, V=3, D=16):
    h, w = H // 4, W // 4
    rng = np.random.default_rng(500 + shard)
    imgs = torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev)
    proj = torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D, interval=synth.sweep_interval(D))).to(dev)
    gt = torch.from_numpy((synth.DTU_TARGET_Z + 20 * rng.standard_normal((1, h, w))).astype(np.float32)).to(dev)
    model.zero_grad()
    out = model(imgs, proj, dv)
    mvsnet_loss(out["depth"], gt, torch.ones(1, h, w, device=dev)).backward()
    return [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in model.parameters()]


def main():
    rank, world, dev = parallel.init_distributed()
    torch.manual_seed(3)
    model = MVSNet(refine=False).to(dev)
    parallel.broadcast_parameters(model, 0)
    model.train()
    for m in model.modules():          # frozen statistics: the two shards must not perturb each other's pass
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.momentum = 0.0
    own = shard_grads(model, dev, rank)
    for p, g in zip(model.parameters(), own):
        p.grad = g.clone()
    parallel.FlatGradAllReduce(model.parameters())()
    avg = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(avg) for _ in range(world)]
    dist.all_gather(gathered, avg)
    if rank == 0:
        shards = [torch.cat([g.reshape(-1) for g in shard_grads(model, dev, r)]) for r in range(world)]
        mean = sum(shards) / world
        scale = float(mean.abs().max())
        print(json.dumps({"ranks_agree": all(bool(torch.equal(gathered[0], x)) for x in gathered[1:]),
                          "max_rel_err_vs_mean_of_shards": float((avg - mean).abs().max()) / scale,
                          "grad_absmax": scale}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
