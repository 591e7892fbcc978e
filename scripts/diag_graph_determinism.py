#!/usr/bin/env python3
"""Diagnostic: is the small training step of tests/test_gpu_train_graph.py deterministic run to run, and do the one-graph,
two-graph and eager forms agree bit for bit?"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from test_gpu_train_graph import _run

runs = {"single_a": _run(False), "single_b": _run(False), "split": _run(True), "eager_a": _run(False, graphed=False),
        "eager_b": _run(False, graphed=False)}
base = runs["single_a"]
for k, (l, p) in runs.items():
    d = max(float((p[n].float() - base[1][n].float()).abs().max()) for n in p)
    print(k, [float(x) for x in l], "max param diff vs single_a", d, "loss equal", torch.equal(l, base[0]))
