#!/usr/bin/env python3
"""Does a producer -> consumer pair that REUSES one slab-sized buffer keep the intermediate out of HBM (Infinity Cache, 256 MB)?
K rounds of (write S bytes, read S bytes) on one buffer vs on K different buffers, and one pass over K*S bytes.
python scripts/exp_mall_reuse.py [slab_MB] [K]"""
import sys
import torch

S = int(sys.argv[1]) if len(sys.argv) > 1 else 91
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
n = S * (1 << 20) // 4
src = torch.randn(n, device=dev)
bufs = [torch.empty(n, device=dev) for _ in range(K)]
big = torch.empty(K * n, device=dev)
acc = torch.zeros(n, device=dev)


def producer(dst):
    torch.mul(src, 1.0001, out=dst)          # read S (src stays hot), write S


def consumer(x):
    acc.add_(x)                               # read S (+ acc read/write)


def run(kind):
    if kind == "same":
        for _ in range(K):
            producer(bufs[0]); consumer(bufs[0])
    elif kind == "different":
        for k in range(K):
            producer(bufs[k]); consumer(bufs[k])
    else:   # all producers first, then all consumers (the unfused order at full size)
        for k in range(K):
            producer(bufs[k])
        for k in range(K):
            consumer(bufs[k])


for kind in ("same", "different", "batched", "same", "different", "batched"):
    for _ in range(2):
        run(kind)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run(kind)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    moved = K * S * 5 / 1024          # GB touched by the kernels per run: src read, buf write, buf read, acc read + write
    print(f"{kind:10s} slab {S} MB x {K}: {ms:.3f} ms per run  ({moved / ms * 1e3:.0f} GB/s of kernel-side traffic)")
