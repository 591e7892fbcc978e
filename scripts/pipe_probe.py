"""Where the eval driver's end-to-end rate goes (device pipeline): marginal seconds per scan with the PFM writer
on / off and with few / many decode threads.  python scripts/pipe_probe.py"""
import os, sys, time, tempfile, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts")); sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import bench_pipeline as bp
from mvs_amd import synth
from mvs_amd.tools import eval_depth
root = tempfile.mkdtemp(prefix="dtu_synth_")
for i in range(6):
    lf = bp.build(root, 49, scan=f"scan{i + 1}")
ckpt = os.path.join(root, "m.ckpt"); torch.save({"model": synth.random_state_dict(0)}, ckpt)
def lst(n):
    p = os.path.join(root, f"l{n}.txt"); open(p, "w").write("".join(f"scan{i+1}\n" for i in range(n))); return p
def run(n, extra=()):
    t0 = time.perf_counter()
    eval_depth.main(["--testpath", root, "--testlist", lst(n), "--loadckpt", ckpt, "--nviews", "5", "--numdepth", "192", "--quiet",
                     "--outdir", os.path.join(root, f"o{n}"), "--device_pipeline"] + list(extra))
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run(1)
for name, extra, env in (("default", (), {}), ("no writer", (), {"MVS_EVAL_NO_WRITE": "1"}), ("4 decode threads", ("--decode_workers", "4"), {}),
                         ("no writer, 4 decode threads", ("--decode_workers", "4"), {"MVS_EVAL_NO_WRITE": "1"})):
    os.environ.update(env)
    a, b = run(2, extra), run(6, extra)
    for k in env: os.environ.pop(k)
    print(f"{name}: 2 scans {a:.3f} s, 6 scans {b:.3f} s, marginal {(b - a) / 4 / 49 * 1e3:.2f} ms per depth map", flush=True)
