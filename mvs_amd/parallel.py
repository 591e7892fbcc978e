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


class GraphedTrainStep:
    """One data-parallel training step -- zero_grad -> forward -> loss -> backward -> gradient average -> optimizer step
    (MVSNet/train.py:204-248 under CasMVSNet/train.py:365-393's one process per GPU) -- as HIP-graph replays, so that N ranks do
    not each push ~300 kernel launches per step through their interpreters (the eager step is host-bound: DESIGN.md section 4).

    split=False (one rank): the whole step is ONE graph.
    split=True (any number of ranks; the default when the process group has more than one): graph A = zero_grad -> forward ->
    loss -> backward -> pack every gradient into one flat fp32 buffer (one concatenation kernel); then ONE RCCL all-reduce of
    that buffer, launched eagerly on the same stream between the two replays (1.35 MB for MVSNet: latency-bound on xGMI, one
    bucket); graph B = scale by 1/world, one multi-tensor copy back into the gradient tensors, the optimizer step.  The
    optimizer must be `capturable` (torch.optim.Adam(..., capturable=True)).

    forward_loss() runs the model on STATIC input tensors the caller refills before each replay and returns the scalar loss."""

    def __init__(self, params, opt, forward_loss, split=None, allreduce=True, warmup=2):
        self.params = [p for p in params if p.requires_grad]
        self.opt = opt
        self.forward_loss = forward_loss
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.split = (self.world > 1) if split is None else bool(split)
        if self.world > 1 and not self.split:
            raise ValueError("more than one rank needs the split form (the all-reduce sits between the two graphs)")
        self.allreduce = bool(allreduce) and self.world > 1
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32) if self.split else None
        self.views = [v.view(p.shape) for v, p in zip(self.flat.split([p.numel() for p in self.params]), self.params)] \
            if self.split else None
        self.graphs = []
        self.loss = None
        self._capture(warmup)

    # the two halves of a split step; eager and captured runs go through the same code
    def _backward_and_pack(self):
        self.opt.zero_grad(set_to_none=True)      # backward then WRITES each gradient (graph-pool memory under capture): no fill, no add
        loss = self.forward_loss()
        loss.backward()
        if self.split:
            dev = self.flat.device
            torch.cat([p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), device=dev, dtype=torch.float32)
                       for p in self.params], out=self.flat)
        return loss

    def _average_and_step(self):
        if self.split:
            if self.world > 1 and self.allreduce:      # (allreduce=False is a timing aid: nothing was summed, nothing to average)
                self.flat.mul_(1.0 / self.world)
            # back into the tensors backward wrote (one multi-tensor copy): the optimizer then runs the very kernels of the
            # one-graph step on the very same operands -- with the gradients as VIEWS of the flat buffer torch's multi-tensor
            # Adam took another code path (unaligned views) and the parameters drifted from the one-graph step's in the last bit
            pairs = [(p.grad, v) for p, v in zip(self.params, self.views) if p.grad is not None]     # (a parameter the loss does not reach has none)
            if pairs:
                torch._foreach_copy_([g for g, _ in pairs], [v for _, v in pairs])
        self.opt.step()

    def _reduce(self):
        if self.allreduce:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    def _capture(self, warmup):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):      # optimizer state, lazy packs, the RCCL communicator: all before the capture
                self._backward_and_pack()
                self._reduce()
                self._average_and_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ga = torch.cuda.CUDAGraph()
        if not self.split:
            with torch.cuda.graph(ga, capture_error_mode="relaxed"):
                self.loss = self._backward_and_pack()
                self._average_and_step()
            self.graphs = [ga]
            return
        with torch.cuda.graph(ga, capture_error_mode="relaxed"):
            self.loss = self._backward_and_pack()
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="relaxed"):
            self._average_and_step()
        self.graphs = [ga, gb]

    def replay(self, events=None):
        """One step.  events: a list that receives (start, stop) HIP events around the all-reduce."""
        self.graphs[0].replay()
        if self.split:
            if events is not None and self.allreduce:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); self._reduce(); b.record()
                events.append((a, b))
            else:
                self._reduce()
            self.graphs[1].replay()
        return self.loss

    @property
    def launch_note(self):
        if not self.split:
            return "one HIP graph replay per step"
        return ("two HIP graph replays per step (backward + flat pack | average + optimizer) around "
                + ("one eager RCCL all-reduce" if self.allreduce else "a no-op all-reduce"))


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
