"""The N-rank control flow of bench.py on a ONE-GPU box (the build pool has nothing else): MVS_BENCH_ONE_DEVICE=1 puts every rank
on cuda:0 and swaps RCCL for gloo, so that what the driver's multi-GPU node would run -- plain `python bench.py --gpus N` starting
its own ranks, the barrier / max-over-ranks timing, rank 0's line, and the N-rank training child (two HIP graphs around the
all-reduce of the flat gradient) -- runs end to end before it ever meets eight GPUs.  The numbers mean nothing and the line says
so; tests/test_gpu_multi.py holds the same checks for real devices (skipped below two GPUs)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["MVS_BENCH_ONE_DEVICE"] = "1"
    return env


def test_plain_bench_gpus_2_runs_two_ranks_and_the_two_rank_training_child():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--height", "256", "--width", "320", "--views", "3", "--ndepth", "32"], cwd=REPO, env=_env(),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines          # ONE JSON line, rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert "REHEARSAL" in line["config"]["sharding"]
    tr = line["train"]
    assert "error" not in tr, tr
    assert tr["n_gpus"] == 2 and tr["launch"].startswith("two HIP graph replays") and tr["allreduce_us"] > 0
    assert tr["guard_fallbacks"] == 0 and tr["loss"] == tr["loss"]


def test_train_mode_two_ranks_eager_and_graph_lines():
    """`--mode train` on two ranks, eager (FlatGradAllReduce) and as graph replays: both lines carry the all-reduce's time and a
    finite loss (the two take different numbers of warm-up steps: their losses are not comparable)."""
    def run(extra):
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--mode", "train", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                            "--height", "256", "--width", "320", "--views", "3", "--ndepth", "32"] + extra, cwd=REPO, env=_env(),
                           capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    a, b = run([]), run(["--graph"])
    assert a["n_gpus"] == b["n_gpus"] == 2 and a["launch"].startswith("eager") and b["launch"].startswith("two HIP graph replays")
    assert a["allreduce_us"] > 0 and b["allreduce_us"] > 0
    assert 0 < a["loss"] < 1e4 and 0 < b["loss"] < 1e4


def test_two_graph_step_averages_the_gradient_over_the_ranks():
    """GraphedTrainStep's split form on two ranks with different samples: after a replay every rank holds the same flat gradient,
    and it is the MEAN of the two per-shard gradients (recomputed eagerly on rank 0) to the training path's run-to-run noise."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join("tests", "graph_grad_worker.py")]
    r = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["graphs"] == 2 and res["ranks_agree"], res
    assert res["shards_differ"] > 1e-2, res                 # the two ranks did see different data
    assert res["max_rel_err_vs_mean_of_shards"] < 1e-4, res


def test_plain_bench_gpus_8_runs_eight_ranks():
    """ The N the driver's node will use: plain `python bench.py --gpus 8 --no-extras` starts EIGHT ranks (all on
    this one device here), scrubs the rendezvous environment, picks its port, and rank 0 prints ONE line with n_gpus = 8 whose
    value is the MAX-over-ranks time turned into an aggregate -- eight times what one rank did."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--no-extras", "--steps", "2", "--warmup", "1",
                        "--height", "128", "--width", "160", "--views", "3", "--ndepth", "16"], cwd=REPO, env=_env(),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["steps"] == 2
    assert "REHEARSAL" in line["config"]["sharding"] and "x8" in line["config"]["sharding"]
    assert abs(line["value"] - 8 * 1e3 / line["ms_per_step"]) <= 1e-3 * line["value"]       # aggregate = ranks x steps / max time
    assert "train" not in line and line["summary"]["n_gpus"] == 8
