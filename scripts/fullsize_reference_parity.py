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
from fullsize_cases import run_cas, run_cvp, run_eval_small, run_mvsnet, run_train_step  # noqa: E402
from trained_cases import run_trained  # noqa: E402


def main():
    res = {}
    which = sys.argv[1:] or ["mvsnet", "mvsnet_fast", "mvsnet_s1", "mvsnet_fast_s1", "eval_small", "eval_small_fast",
                             "cas", "cas_s1", "cvp", "cvp_s1", "cvp_s2", "train", "trained_small", "trained_small_exact", "trained_full",
                             "trained_full_exact"]
    table = {"mvsnet": lambda: run_mvsnet(False), "mvsnet_fast": lambda: run_mvsnet(True),
             "mvsnet_s1": lambda: run_mvsnet(False, 1), "mvsnet_fast_s1": lambda: run_mvsnet(True, 1),
             "eval_small": lambda: run_eval_small(False), "eval_small_fast": lambda: run_eval_small(True),
             "trained_small": lambda: run_trained("small", True), "trained_small_exact": lambda: run_trained("small", False),
             "trained_full": lambda: run_trained("full", True), "trained_full_exact": lambda: run_trained("full", False),
             "cas": run_cas, "cvp": run_cvp, "cas_s1": lambda: run_cas(1), "cvp_s1": lambda: run_cvp(1), "cvp_s2": lambda: run_cvp(2)}
    for w in which:
        if w == "train":
            res[w] = run_train_step()
        else:
            with torch.no_grad():
                res[w] = table[w]()
        print(w, json.dumps(res[w]), flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "fullsize_reference_parity.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
