"""Worker of tests/test_gpu_rehearsal.py / test_gpu_multi.py: the averaged gradient of the two-graph training step
(mvs_amd/parallel.py::GraphedTrainStep, split form: graph A | all-reduce | graph B) against the mean of the per-shard gradients
recomputed eagerly on rank 0.  Launched with torch.distributed.run; MVS_BENCH_ONE_DEVICE=1 = every rank on cuda:0 over gloo."""
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

H, W, V, D = 128, 160, 3, 16


def sample(dev, shard):
    h, w = H // 4, W // 4
    rng = np.random.default_rng(500 + shard)
    imgs = torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev)
    gt = torch.from_numpy((synth.DTU_TARGET_Z + 20 * rng.standard_normal((1, h, w))).astype(np.float32)).to(dev)
    return imgs, gt


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    one_device = os.environ.get("MVS_BENCH_ONE_DEVICE") == "1"
    local = 0 if one_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo" if one_device else "nccl")
    torch.manual_seed(3)
    model = MVSNet(refine=False).to(dev)
    parallel.broadcast_parameters(model, 0)
    model.train()
    for m in model.modules():          # frozen running statistics: extra passes must not change what a later pass sees
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.momentum = 0.0
    h, w = H // 4, W // 4
    proj = torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev)
    dv = torch.from_numpy(synth.depth_values(D, interval=synth.sweep_interval(D))).to(dev)
    mask = torch.ones(1, h, w, device=dev)
    imgs, gt = sample(dev, rank)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)          # the step must not move the weights: rank 0 recomputes both shards

    def forward_loss():
        return mvsnet_loss(model(imgs, proj, dv)["depth"], gt, mask)

    g = parallel.GraphedTrainStep(model.parameters(), opt, forward_loss, split=True, warmup=1)
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    avg = g.flat.clone()               # after graph B: the all-reduced sum scaled by 1 / world
    gathered = [torch.empty_like(avg) for _ in range(world)]
    dist.all_gather(gathered, avg)
    if rank == 0:
        shards = []
        for r in range(world):
            si, sg = sample(dev, r)
            model.zero_grad(set_to_none=True)
            mvsnet_loss(model(si, proj, dv)["depth"], sg, mask).backward()
            shards.append(torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.requires_grad]))
        mean = sum(shards) / world
        scale = float(mean.abs().max())
        print(json.dumps({"ranks_agree": all(bool(torch.equal(gathered[0], x)) for x in gathered[1:]),
                          "max_rel_err_vs_mean_of_shards": float((avg - mean).abs().max()) / scale,
                          "shards_differ": float((shards[0] - shards[-1]).abs().max()) / scale,
                          "grad_absmax": scale, "graphs": len(g.graphs)}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
