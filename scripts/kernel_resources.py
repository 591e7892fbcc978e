#!/usr/bin/env python3
"""Registers / spills / occupancy of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage):
    python scripts/kernel_resources.py conv_f16x3 [-DMVS_TUNING]
A before/after check for changes that must not move a hot kernel's register allocation."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mvs_amd import build as B  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    src = os.path.join(B.CSRC, name + ".hip")
    cmd = [B.hipcc()] + B.FLAGS + extra + ["--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", line) or re.search(r":\d+:\d+: +(Function Name|Name): (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur)
            rows[cur] = {}
            continue
        m = re.search(r"(VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    for k, v in rows.items():
        print(f"{k:90s} vgpr {v.get('VGPRs', 0):3d} agpr {v.get('AGPRs', 0):3d} sspill {v.get('SGPRs Spill', 0):3d} vspill {v.get('VGPRs Spill', 0):3d} "
              f"scratch {v.get('ScratchSize', 0):4d} occ {v.get('Occupancy', 0)} lds {v.get('LDS Size', 0)}")


if __name__ == "__main__":
    main()
