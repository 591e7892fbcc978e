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
