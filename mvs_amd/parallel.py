"""One process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on
ROCm) -- or gloo on CPU for the tests.

Inference (BASELINE configs 2-4): reference views are independent units
(MVSNet/datasets/dtu_yao_eval.py:73-108), so ranks take them round-robin and
the data path has NO collective.

Training (config 5): data parallel, one exchange per iteration: all-reduce of
the flat fp32 gradient (338,129 elements = 1.35 MB for MVSNet).  The message is
latency-bound on xGMI, so it goes out as ONE bucket -- the counterpart of the
reference's nn.DataParallel reduce-add (MVSNet/train.py:95) and CasMVSNet's DDP
(CasMVSNet/train.py:365-372).  BatchNorm statistics stay per-rank, as in both.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from the torchrun environment.  Returns (rank, world, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank,
                                world_size=world)
    return rank, world, device


def shard_ref_views(n_items, rank, world):
    """Indices of the reference views rank `rank` processes (stride = world, the
    DistributedSampler pattern of CasMVSNet/train.py:386-387, without padding)."""
    return list(range(rank, n_items, world))


class FlatGradAllReduce:
    """Average gradients across ranks with a single all-reduce of one flat
    fp32 buffer (allocated once; parameters without grad contribute zeros)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def __call__(self):
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        # pack with ONE concatenation kernel, unpack with ONE multi-tensor copy (~60 parameters:
        # per-tensor copies were 120 launches a step)
        pieces = [p.grad.reshape(-1) if p.grad is not None else
                  torch.zeros(p.numel(), device=dev, dtype=torch.float32) for p in self.params]
        torch.cat(pieces, out=self.flat)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        views = [v.view(p.shape) for v, p in zip(self.flat.split([p.numel() for p in self.params]), self.params)]
        for p, v in zip(self.params, views):
            if p.grad is None:
                p.grad = torch.empty_like(p)
        torch._foreach_copy_([p.grad for p in self.params], views)


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s weights and buffers."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src)


def reduce_scalars(scalars, dst=0):
    """Mean of a dict of python/0-d scalars on rank dst (CasMVSNet/utils.py:183-201)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(scalars)
    keys = sorted(scalars)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" \
        else torch.device("cpu")
    t = torch.tensor([float(scalars[k]) for k in keys], device=dev, dtype=torch.float64)
    dist.reduce(t, dst=dst)
    if dist.get_rank() == dst:
        t /= dist.get_world_size()
    return {k: float(v) for k, v in zip(keys, t.tolist())}
