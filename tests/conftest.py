import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "poisoned_inputs: the test feeds non-finite / outlier data on purpose (the range guard is expected to speak)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests skip (not fail) on a box without a GPU or without the built library."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have = False
    have = have and os.path.exists(os.path.join(REPO, "mvs_amd", "csrc", "libmvs_hip.so"))
    if have:
        return
    skip = pytest.mark.skip(reason="needs an MI355X and the built libmvs_hip.so (python -m mvs_amd.build)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _range_guard_stays_silent(request):
    """Every -m gpu test outside tests/test_gpu_range_guard.py (and those marked `poisoned_inputs`) runs on ordinary data: no launch of a two-piece fp16 layer may take
    the range guard's fp32 path (mvs_amd/csrc/conv_guard.h) -- a false positive would be correct but ~35x slower, silently."""
    if "gpu" not in request.keywords or "poisoned_inputs" in request.keywords or request.module.__name__.endswith("test_gpu_range_guard"):
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from mvs_amd import ops
    before = ops.guard_fallback_count()
    yield
    taken = ops.guard_fallback_count() - before
    assert taken == 0, f"{taken} two-piece launches fell back to fp32 on this test's data (range guard false positive?)"


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def weights():
    return load_golden("weights_seed0")


def rot_trans_torch(proj, v, ref=0):
    """rows of (src_proj @ inverse(ref_proj))[:3,:4] evaluated with torch in
    fp32 exactly as the reference does (MVSNet/models/module.py:63-65)."""
    import torch
    P = torch.from_numpy(np.ascontiguousarray(proj))
    M = P[:, v] @ torch.inverse(P[:, ref])
    return M[:, :3, :4].reshape(-1, 12).contiguous().numpy()
