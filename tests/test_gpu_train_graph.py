"""The data-parallel training step as HIP-graph replays (mvs_amd/parallel.py::GraphedTrainStep):
on one GPU the two-graph form that N ranks run (graph A: zero_grad -> forward -> loss -> backward -> flat pack | all-reduce |
graph B: average -> copy back -> optimizer) must reproduce the single-graph step and the eager step when its all-reduce is a
no-op.  Reference step: MVSNet/train.py:204-248.

"Reproduce" is to the training path's own run-to-run noise: the weight-gradient kernels finish with floating-point atomic adds
(conv3d_wgrad.hip) and train-mode BatchNorm reduces its statistics the same way, so two identical one-graph runs already
differ in the last bits of a loss or a gradient (scripts/diag_graph_determinism.py: third loss
227.07487 / 227.07492 for the same form twice).  Under Adam a last-bit change of a near-zero gradient moves a parameter by
2 x lr, so the parameter comparison runs under plain SGD, where it is proportional; the Adam step is compared through the
bench line's loss."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def _run(split, steps=3, graphed=True, adam=False):
    from mvs_amd import parallel, synth
    from mvs_amd.models import MVSNet, mvsnet_loss
    dev = torch.device("cuda:0")
    H, W, V, D = 128, 160, 3, 16
    h, w = H // 4, W // 4
    torch.manual_seed(3)
    model = MVSNet(refine=False).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True) if adam else torch.optim.SGD(model.parameters(), lr=1e-4)
    rng = np.random.default_rng(5)
    proj = torch.from_numpy(synth.proj_matrices(V, h, w)).to(dev)
    dvals = torch.from_numpy(synth.depth_values(D)).to(dev)
    pool = [(torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev),
             torch.from_numpy((synth.DTU_TARGET_Z + 20 * rng.standard_normal((1, h, w))).astype(np.float32)).to(dev)) for _ in range(2)]
    mask = torch.ones(1, h, w, device=dev)
    slot_i, slot_g = pool[0][0].clone(), pool[0][1].clone()

    def forward_loss():
        return mvsnet_loss(model(slot_i, proj, dvals)["depth"], slot_g, mask)

    losses = []
    if graphed:
        g = parallel.GraphedTrainStep(model.parameters(), opt, forward_loss, split=split, warmup=2)
        assert len(g.graphs) == (2 if split else 1)
        for i in range(steps):
            slot_i.copy_(pool[i % 2][0]); slot_g.copy_(pool[i % 2][1])
            losses.append(g.replay().detach().clone())
    else:
        for i in range(2 + steps):          # the same 2 warm-up steps GraphedTrainStep takes on pool[0], then the timed ones
            j = 0 if i < 2 else (i - 2) % 2
            slot_i.copy_(pool[j][0]); slot_g.copy_(pool[j][1])
            opt.zero_grad(set_to_none=True)
            ls = forward_loss()
            ls.backward()
            opt.step()
            if i >= 2:
                losses.append(ls.detach().clone())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def test_split_graph_step_equals_single_graph_and_eager_steps():
    l1, p1 = _run(split=False)
    l1b, p1b = _run(split=False)            # the same form again: the yardstick (run-to-run noise of the float atomics)
    l2, p2 = _run(split=True)
    l0, p0 = _run(split=False, graphed=False)
    assert torch.isfinite(l1).all() and float(l1[0]) != float(l1[-1])          # the step does train
    # (even the first loss moves in its last bit from run to run: train-mode BatchNorm reduces its batch statistics with atomics)
    def dist(a, b, k):
        return float((a[k].double() - b[k].double()).abs().max()) / (float(a[k].double().abs().max()) + 1e-3)
    # a FIXED bound: two runs of the SAME form differ by up to ~6e-4 here (a BatchNorm bias starts at 0 -- after five steps it IS
    # lr x the sum of its gradients, noise and all); a wrong step -- a gradient lost in the flat pack, the average applied twice,
    # a stale pack in a replay -- moves parameters by O(1) of their change
    for k in p1:
        assert dist(p1, p1b, k) <= 5e-3, (k, dist(p1, p1b, k))
        assert dist(p1, p2, k) <= 5e-3, (k, dist(p1, p2, k))
        assert dist(p1, p0, k) <= 5e-3, (k, dist(p1, p0, k))
    # ... and the parameters did move
    assert max(float((p1[k].double() - p0[k].double()).abs().max()) for k in p1) < 1.0
    assert torch.allclose(l1, l2, rtol=1e-5, atol=0) and torch.allclose(l1, l0, rtol=1e-5, atol=0), (l1, l2, l0)


def test_split_graph_adam_step_tracks_single_graph():
    l1, _ = _run(split=False, adam=True)
    l2, _ = _run(split=True, adam=True)
    assert torch.allclose(l1, l2, rtol=2e-5, atol=0), (l1, l2)


def test_bench_train_line_split_graph_matches_single_graph():
    """`bench.py --mode train --graph` (one graph) and `--graph --graph-split` (the N-rank form on one rank): same loss after
    the same steps, and the line says which launch form ran."""
    def run(extra):
        r = subprocess.run([sys.executable, "bench.py", "--mode", "train", "--graph", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                            "--height", "256", "--width", "320", "--views", "3", "--ndepth", "32"] + extra,
                           cwd=REPO, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    a, b = run([]), run(["--graph-split"])
    assert a["launch"].startswith("one HIP graph replay") and b["launch"].startswith("two HIP graph replays"), (a["launch"], b["launch"])
    assert abs(a["loss"] - b["loss"]) <= 1e-4 * abs(a["loss"]) and a["n_gpus"] == b["n_gpus"] == 1
    assert b["allreduce_us"] is None and b["guard_fallbacks"] == 0
