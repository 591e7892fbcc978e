"""Multi-GPU paths on real hardware; every test skips on a box with fewer than two GPUs (the
build pool hands out single-GPU boxes -- the driver's multi-GPU node runs these)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")]


def _torchrun(args, port, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_bench_two_gpus_shards_reference_views_without_a_collective():
    out = _torchrun(["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], 29611)
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert "no collective" in line["config"]["sharding"]


def test_plain_python_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher starts its own two ranks and forwards rank 0's line;
    the training child of the line runs on two ranks as graph replays around the RCCL all-reduce."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2"], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    tr = line["train"]
    assert "error" not in tr, tr
    assert tr["n_gpus"] == 2 and tr["launch"].startswith("two HIP graph replays") and tr["allreduce_us"] > 0


def test_bench_train_graph_two_gpus():
    out = _torchrun(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "2", "--mode", "train", "--graph", "--no-cpu-baseline"], 29614)
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["launch"].startswith("two HIP graph replays")
    assert line["allreduce_us"] is not None and line["allreduce_us"] > 0


def test_bench_train_mode_two_gpus_allreduces_the_flat_gradient():
    out = _torchrun(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "2", "--mode", "train"], 29612)
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["grad_floats"] == 338129
    assert line["allreduce_us"] is not None and line["allreduce_us"] > 0


def test_rccl_gradient_average_equals_mean_of_shard_gradients():
    """Two ranks, two different samples, the HIP training path: after FlatGradAllReduce every rank holds
    the mean of the two per-shard gradients (checked against both shards recomputed on rank 0)."""
    out = _torchrun([os.path.join("tests", "rccl_grad_worker.py")], 29613)
    res = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert res["ranks_agree"] and res["max_rel_err_vs_mean_of_shards"] < 1e-5, res


def test_two_graph_step_averages_the_gradient_over_two_gpus():
    """The same check as tests/test_gpu_rehearsal.py::test_two_graph_step_averages_the_gradient_over_the_ranks over RCCL on two devices."""
    out = _torchrun([os.path.join("tests", "graph_grad_worker.py")], 29618)
    res = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert res["graphs"] == 2 and res["ranks_agree"] and res["max_rel_err_vs_mean_of_shards"] < 1e-4, res
