#!/usr/bin/env python3
"""HIP path vs the REFERENCE's own CPU outputs at the sizes BASELINE.json names (fixtures
tests/golden/g12..g14, made by tests/golden/make_golden_fullsize.py in the build container).
Prints the numbers the assertions of tests/test_gpu_fullsize_reference.py are set from."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from fullsize_cases import run_cas, run_cvp, run_mvsnet  # noqa: E402


def main():
    res = {}
    which = sys.argv[1:] or ["mvsnet", "mvsnet_fast", "cas", "cvp"]
    with torch.no_grad():
        for w in which:
            res[w] = {"mvsnet": lambda: run_mvsnet(False), "mvsnet_fast": lambda: run_mvsnet(True),
                      "cas": run_cas, "cvp": run_cvp}[w]()
            print(w, json.dumps(res[w]), flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "fullsize_reference_parity.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
