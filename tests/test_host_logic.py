"""CPU-side checks: the C-ABI library loads and exports every symbol
include/mvs_hip.h declares, the Python mirror keeps the reference's surface and
state_dict layout, host logic (sharding, flat gradient all-reduce over gloo with
world_size 2), and bench.py's algorithmic work equals BASELINE.md section 3.
No compute kernel is launched here."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden


def test_library_exports_every_declared_symbol():
    from mvs_amd import _lib
    hdr = open(os.path.join(REPO, "include", "mvs_hip.h")).read()
    declared = set(re.findall(r"\b(mvs_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in mvs_hip.h but not exported"
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert lib.mvs_version() >= 100
    assert lib.mvs_arch() == b"gfx950"


def test_only_hip_runtime_of_the_process_is_torchs():
    from mvs_amd import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    hips = {ln.split()[-1] for ln in maps.splitlines() if "libamdhip64" in ln}
    assert len(hips) == 1, hips   # one HIP runtime: pointers/streams are shared with torch


def test_ops_fail_loudly_without_device_tensors():
    from mvs_amd import ops
    from mvs_amd._lib import MvsHipError
    with pytest.raises(MvsHipError):
        ops.softmax_regress_conf(torch.zeros(1, 4, 2, 2), torch.zeros(1, 4))
    with pytest.raises(MvsHipError):
        ops.costvol_variance_cl(torch.zeros(1, 4, 4, 8), torch.zeros(1, 1, 4, 4, 8),
                                torch.zeros(1, 1, 12), torch.zeros(1, 2))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from mvs_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MvsHipError):
        _lib.load()


def test_mfma_config_table():
    from mvs_amd import _lib
    lib = _lib.load()
    for cin, cout, s in ((32, 8, 1), (8, 16, 2), (16, 16, 1), (16, 32, 2), (32, 32, 1),
                         (32, 64, 2), (64, 64, 1)):
        assert lib.mvs_conv3d_mfma_supported(0, cin, cout, s) == 1
        assert lib.mvs_conv3d_packed_weight_floats(0, cin, cout, s) >= 27 * cin * cout
    assert lib.mvs_conv3d_mfma_supported(0, 5, 7, 1) == 0
    assert lib.mvs_conv3d_packed_weight_floats(0, 5, 7, 1) == 0


def test_state_dict_layout_matches_reference(weights):
    from mvs_amd.models import MVSNet, load_reference_checkpoint
    model = MVSNet(refine=False)
    assert set(model.state_dict().keys()) == set(weights.keys())
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(weights[k].shape), k
    assert sum(p.numel() for p in model.parameters()) == 338129
    assert sum(p.numel() for p in model.cost_regularization.parameters()) == 298009
    ckpt = {"epoch": 3, "model": {"module." + k: torch.from_numpy(v) for k, v in weights.items()}}
    load_reference_checkpoint(model, ckpt)          # DataParallel-prefixed (train.py:159-164)
    np.testing.assert_array_equal(model.feature.feature.bias.detach().numpy(),
                                  weights["feature.feature.bias"])
    assert set(MVSNet(refine=True).state_dict()) > set(weights)


def test_python_surface():
    import inspect
    from mvs_amd import models
    for name in ("MVSNet", "mvsnet_loss", "homo_warping", "depth_regression"):
        assert hasattr(models, name)
    sig = inspect.signature(models.homo_warping)
    assert list(sig.parameters)[:4] == ["src_fea", "src_proj", "ref_proj", "depth_values"]
    p = torch.softmax(torch.randn(2, 5, 3, 4), 1)
    dv = torch.linspace(400, 900, 5).repeat(2, 1)
    np.testing.assert_allclose(models.depth_regression(p, dv).numpy(),
                               (p * dv.view(2, 5, 1, 1)).sum(1).numpy())
    est, gt = torch.tensor([1.0, 5.0, 2.0]), torch.tensor([1.5, 2.0, 2.0])
    loss = models.mvsnet_loss(est, gt, torch.tensor([1.0, 1.0, 0.0]))
    assert loss.item() == pytest.approx((0.5 * 0.25 + 2.5) / 2)


def test_synthetic_geometry():
    from mvs_amd import synth
    P = synth.proj_matrices(5, 296, 400)[0].astype(np.float64)
    tgt = np.array([0, 0, synth.DTU_TARGET_Z, 1.0])
    for v in range(5):                       # every camera looks at the target point
        uvw = P[v] @ tgt
        u, vv = uvw[0] / uvw[2], uvw[1] / uvw[2]
        assert abs(u - synth.DTU_CX / 4) < 1e-3 and abs(vv - synth.DTU_CY / 4) < 1e-3
    dv = synth.depth_values(192)
    assert dv.shape == (1, 192) and dv[0, 0] == 425.0
    assert abs(dv[0, -1] - (425 + 2.65 * 191)) < 1e-3


def test_bench_algorithmic_work_matches_baseline_md():
    sys.path.insert(0, REPO)
    import bench
    work = bench.algorithmic_work(5, 32, 192, 296, 400)
    assert abs(work["costvol_variance"][1] / 1e9 - 2.986) < 2e-3
    assert abs(work["softmax_regress_conf"][1] / 1e6 - 91.9) < 0.1
    flops = sum(v for k, (kind, v) in work.items() if kind == "mfma" and k.startswith("costreg."))
    assert abs(flops / 1e9 - 461.6) < 0.5
    fnet = sum(v for k, (kind, v) in work.items() if k.startswith("feature."))
    assert abs(fnet / 1e9 - 89.0) < 1.0          # ~17.8 GFLOP per view
    assert abs(work["costreg.conv0"][1] / 1e9 - 314.3) < 0.1


def test_ref_view_sharding():
    from mvs_amd.parallel import shard_ref_views
    for world in (1, 2, 4, 8):
        got = sorted(i for r in range(world) for i in shard_ref_views(49, r, world))
        assert got == list(range(49))
    assert shard_ref_views(49, 3, 8) == [3, 11, 19, 27, 35, 43]


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from mvs_amd.parallel import (FlatGradAllReduce, broadcast_parameters, init_distributed,
                                  reduce_scalars, shard_ref_views)
    from mvs_amd.models import FeatureNet
    init_distributed("gloo")
    torch.manual_seed(100 + rank)            # different init per rank ...
    net = FeatureNet()
    broadcast_parameters(net, 0)             # ... made identical here
    net.to(memory_format=torch.channels_last)   # strided (NHWC) weight gradients, as in the training path
    torch.manual_seed(0)
    data = torch.rand(4, 3, 16, 16)          # the global batch, same on every rank
    mine = shard_ref_views(4, rank, world)
    net.train()
    loss = net(data[mine]).square().mean()
    loss.backward()
    if rank == 1:
        net.feature.bias.grad = None         # a parameter without a gradient counts as zeros
    FlatGradAllReduce(net.parameters())()
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    sc = reduce_scalars({"loss": loss.item()})
    q.put((rank, flat.numpy(), torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy(),
           sc))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_gloo_world2():
    """world_size-2 data-parallel step over gloo: averaged gradients are
    identical on both ranks and equal the mean of the per-shard gradients."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, g0, w0, s0), (_, g1, w1, s1) = res
    np.testing.assert_array_equal(w0, w1)
    np.testing.assert_array_equal(g0, g1)
    assert np.abs(g0).max() > 0
    assert "loss" in s0
    # single-process check of the same average
    from mvs_amd.models import FeatureNet
    net = FeatureNet()
    off = 0
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.from_numpy(w0[off:off + p.numel()]).reshape(p.shape))
            off += p.numel()
    torch.manual_seed(0)
    data = torch.rand(4, 3, 16, 16)
    net.train()
    grads = []
    for shard in ([0, 2], [1, 3]):
        net.zero_grad()
        net(data[shard]).square().mean().backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy())
    nb = net.feature.bias.numel()            # the last parameter: rank 1 contributed zeros for it
    grads[1][-nb:] = 0
    np.testing.assert_allclose(g0, (grads[0] + grads[1]) / 2, atol=1e-6)


def test_derived_weight_cache_is_tied_to_the_tensor_object():
    """mvs_amd.train_ops._cached: an entry is hit only by the SAME tensor object at the same version -- not by
    another tensor that shares its storage address and shape (what the caching allocator hands a second model), not after
    an in-place update; it disappears with the weight."""
    import gc
    import torch
    from mvs_amd import train_ops
    train_ops.invalidate_derived()
    w = torch.nn.Parameter(torch.randn(4, 3))
    calls = []
    make = lambda: calls.append(1) or torch.zeros(1)
    a = train_ops._cached(w, "k", make)
    assert train_ops._cached(w, "k", make) is a and len(calls) == 1
    alias = w.detach()                       # same data_ptr, same shape, same _version counter -- a different object
    assert alias.data_ptr() == w.data_ptr()
    train_ops._cached(alias, "k", make)
    assert len(calls) == 2
    with torch.no_grad():
        w.add_(1.0)                          # what an optimizer step does: bumps _version
    train_ops._cached(w, "k", make)
    assert len(calls) == 3
    n = len(train_ops._derived)
    del w, alias, a
    gc.collect()
    assert len(train_ops._derived) < n       # the weakref callbacks evicted the entries


def test_derived_packs_die_with_the_packed_tensor():
    """the split-operand companions and the lazy fp32 fill of a packed weight must not be owned by any
    module-level table -- an id-keyed registry whose value closed over `packed` pinned every eager training step's packs
    (~5 MB per step).  They are attributes of the tensor object: N pack / drop cycles leave nothing behind."""
    import gc
    import weakref
    from mvs_amd import ops
    assert not hasattr(ops, "_lazy_fill") and not hasattr(ops, "_split_registry")
    refs, filled = [], []
    for i in range(50):
        packed = torch.empty(16)
        comp, comp16 = torch.empty(8), torch.empty(4)
        weight = torch.randn(4)
        ops._register_split(packed, comp, comp16)
        ops._register_lazy(packed, lambda dst, w=weight: filled.append(dst.data_ptr()))   # closes over the weight, not the pack
        assert ops.split_companion(packed) is comp and ops.f16_companion(packed) is comp16
        if i % 2:
            assert ops.materialize_packed(packed) is packed and filled[-1] == packed.data_ptr()
            n = len(filled)
            ops.materialize_packed(packed)                 # a second call does not fill again
            assert len(filled) == n
        refs += [weakref.ref(packed), weakref.ref(comp), weakref.ref(comp16)]
        del packed, comp, comp16, weight
    gc.collect()
    assert all(r() is None for r in refs)
    assert ops.split_companion(None) is None and ops.f16_companion(torch.empty(1)) is None


def test_bench_self_launch_command_and_environment(monkeypatch):
    """plain `python bench.py --gpus N` must start N ranks (one process per GPU, CasMVSNet/train.py:365-393)
    -- never relabel a one-rank run.  The command construction, the scrubbed environment and the refusal on a box with fewer
    GPUs are host logic."""
    import json
    sys.path.insert(0, REPO)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29555)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29555"
    i = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        monkeypatch.setenv(k, "7")
    env = bench.clean_env({"MVS_X": "1"})
    assert not any(k in env for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"))
    assert env["MVS_X"] == "1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert 1024 < bench.free_port() < 65536
    assert bench.last_json_line('noise\n{"n_gpus": 2}\ntrailing') == {"n_gpus": 2}
    assert bench.last_json_line("no json here") is None
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(2, ["--gpus", "2"])
    assert "refusing" in str(e.value)
    # a launcher that started another number of ranks than --gpus says: an error, not a relabelled line
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=1" in str(e.value)


def test_depth_metrics_are_per_image_then_batch_means():
    """mvs_amd.tools.metrics against hand-computed cases of MVSNet/utils.py:129-158 (which cannot be imported: torchvision):
    per-image masked means averaged over the batch -- NOT pooled --, the threshold strict, an empty mask NaN."""
    import torch
    from mvs_amd.tools import metrics
    est = torch.tensor([[[1.0, 2.0], [3.0, 4.0]], [[10.0, 10.0], [10.0, 10.0]]])
    gt = torch.tensor([[[1.5, 2.0], [7.0, 0.0]], [[12.0, 10.0], [2.0, 10.0]]])
    mask = torch.tensor([[[True, True], [True, False]], [[True, False], [False, False]]])
    # image 0: |err| over 3 pixels = 0.5, 0, 4 -> 1.5; image 1: one pixel, 2 -> 2; batch mean 1.75 (pooled would be 6.5 / 4 = 1.625)
    assert metrics.abs_depth_error(est, gt, mask).item() == 1.75
    # > 2 strictly: image 0: one of three (4 > 2); image 1: 2 > 2 is false -> 0; mean 1/6
    assert abs(metrics.thres_error(est, gt, mask, 2).item() - (1.0 / 3.0) / 2) < 1e-7
    assert abs(metrics.thres_error(est, gt, mask, 0.4).item() - (2.0 / 3.0 + 1.0) / 2) < 1e-7
    assert metrics.thres_error(est, gt, mask, 8).item() == 0.0
    # against the reference's own formulation (boolean indexing per image)
    g = torch.Generator().manual_seed(0)
    e, t = torch.rand(3, 16, 20, generator=g) * 50, torch.rand(3, 16, 20, generator=g) * 50
    m = torch.rand(3, 16, 20, generator=g) > 0.3
    want_abs = torch.stack([(e[i][m[i]] - t[i][m[i]]).abs().mean() for i in range(3)]).mean()
    want_thr = torch.stack([((e[i][m[i]] - t[i][m[i]]).abs() > 8).float().mean() for i in range(3)]).mean()
    assert abs(metrics.abs_depth_error(e, t, m).item() - want_abs.item()) < 1e-5
    assert abs(metrics.thres_error(e, t, m, 8).item() - want_thr.item()) < 1e-6
    out = metrics.scalar_outputs(e, t, m.float(), loss=torch.tensor(1.0))
    assert sorted(out) == ["abs_depth_error", "loss", "thres2mm_error", "thres4mm_error", "thres8mm_error"]
    empty = torch.zeros(1, 4, 4, dtype=torch.bool)
    assert torch.isnan(metrics.abs_depth_error(e[:1, :4, :4], t[:1, :4, :4], empty))
    import pytest
    with pytest.raises(AssertionError):
        metrics.thres_error(e, t, m, torch.tensor(2.0))
