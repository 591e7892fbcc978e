"""Static properties of the compiled hot kernel that its speed rests on (no GPU: hipcc cross-compiles gfx950 here).

The persistent plane-sweep kernel is bound by vector-instruction issue; a scalar the register allocator spills comes back as a
`v_readlane`, a VECTOR instruction.  Round 6 found ~350 such reloads per wave and depth plane (142 spilled scalars: the kernel's ~70
argument words held in registers for its whole life) and removed them by reading the arguments from the kernel-argument segment where
they are used (mvs_amd/csrc/sweep_persist.hip, `ka->`).  This test keeps a later edit from quietly putting them back."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _isa_count():
    spec = importlib.util.spec_from_file_location("isa_count", os.path.join(ROOT, "scripts", "isa_count.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_persistent_sweep_kernel_does_not_reload_scalars_through_the_vector_pipe():
    stats, meta = _isa_count().analyze("sweep_persist", "variance_fwd_persist_kernelILi4ELi16ELi2ELb1")
    assert len(stats) == 1, list(stats)
    (name, c), = stats.items()
    m = meta[name]
    # configs[1]'s kernel (4 source views, 16-plane tiles, FAST): 25 spilled scalars / 29 v_readlane in the whole binary at the time
    # of writing (142 / 459 with the arguments in registers); no vector register may spill, and LDS must leave room for the tables
    assert m["sgpr_spill_count"] <= 60, m
    assert c["v_readlane"] <= 80, dict(c)
    assert m["vgpr_spill_count"] == 0 and m["vgpr_count"] <= 128, m         # 16 waves per CU = 4 per SIMD
    assert m["group_segment_fixed_size"] <= 160 * 1024, m
    assert c["barrier"] >= 1 and c["lds"] > 0 and c["vmem"] > 0, dict(c)     # (the parse saw the kernel, not an empty stub)
