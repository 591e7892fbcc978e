#!/usr/bin/env python3
"""BASELINE configs[1] forward (1600x1184, N=5, D=192) issued four ways: eager on one stream (bench.py's headline protocol without
its HIP events), eager alternating over two streams, one HIP graph replayed on one stream, two graphs replayed on two streams.
K forwards between two synchronize() calls each; depth-maps/s = K / elapsed."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import synth
from mvs_amd.models import MVSNet
dev = torch.device("cuda:0")
H, W, V, D = 1184, 1600, 5, 192
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(0)
imgs = torch.from_numpy(synth.images(rng, 1, V, H, W)).to(dev)
proj = torch.from_numpy(synth.proj_matrices(V, H // 4, W // 4)).to(dev)
dv = torch.from_numpy(synth.depth_values(D)).to(dev)
model = MVSNet(refine=False)
model.load_state_dict(synth.random_state_dict(seed=0))
model = model.to(dev).eval()
model.proj_where = "device"
def step():
    with torch.no_grad():
        return model(imgs, proj, dv)
for _ in range(10): ref = step()
torch.cuda.synchronize()
def timed(fn, n=K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return {"depth_maps_per_s": round(n / el, 2), "ms_per_step": round(el / n * 1e3, 4)}
out = {}
out["eager_one_stream"] = timed(lambda i: step())
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def two(i):
    with torch.cuda.stream(streams[i % 2]): step()
for i in range(4): two(i)
out["eager_two_streams"] = timed(two)
graphs, outs = [], []
for s in streams:
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step(); step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        o = step()
    graphs.append(g); outs.append(o)
torch.cuda.synchronize()
graphs[0].replay(); torch.cuda.synchronize()
out["graph_bit_equal_to_eager"] = bool(torch.equal(outs[0]["depth"], ref["depth"]))
out["graph_one_stream"] = timed(lambda i: graphs[0].replay())
def two_g(i):
    with torch.cuda.stream(streams[i % 2]): graphs[i % 2].replay()
out["graph_two_streams"] = timed(two_g)
out["eager_one_stream_again"] = timed(lambda i: step())
# bench.py's timed region: the same loop with HIP events around the dominant stage of every step
from mvs_amd import ops
for rep in range(2):
    live = ops.StageTimer(only={"costvol_variance"})
    ops.set_timer(live)
    out["eager_one_stream_bench_events_%d" % rep] = timed(lambda i: step())
    torch.cuda.synchronize()
    out["eager_one_stream_bench_events_%d" % rep]["costvol_variance_ms"] = round(live.summary_ms()["costvol_variance"][1], 4)
    ops.set_timer(None)
    out["eager_one_stream_no_events_%d" % rep] = timed(lambda i: step())
print(json.dumps(out, indent=1))
