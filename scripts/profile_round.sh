#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel trace + stats of the bench command,
# then HBM-traffic counters in their own passes (no tracing mixed in).
# Usage: bash scripts/profile_round.sh r01     (writes gpurun_out/prof_<tag>/...)
TAG=${1:-r06}
cd "$(dirname "$0")/.." ; mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
CMD="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG/trace -o trace --output-format csv -- $CMD > gpurun_out/prof_$TAG/bench_under_trace.json 2> gpurun_out/prof_$TAG/trace.log
# HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots)
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_$TAG/pmc_fetch -o pmc --output-format csv -- $CMD > /dev/null 2> gpurun_out/prof_$TAG/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_$TAG/pmc_write -o pmc --output-format csv -- $CMD > /dev/null 2> gpurun_out/prof_$TAG/pmc_write.log
python - "$TAG" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
root = f"gpurun_out/prof_{tag}"
out = {}
for f in glob.glob(f"{root}/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = rows[:25]
for name in ("fetch", "write"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{root}/pmc_{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:110]
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
    out[name.upper() + "_SIZE_per_launch_KB"] = {k: round(v[1] / v[0], 1) for k, v in
                                                 sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]}
json.dump(out, open(f"{root}/summary.json", "w"), indent=1)
# per-stage HBM-side traffic for bench.py's roofline.traffic (2*FETCH + WRITE, bytes)
names = {"costvol_variance": "variance_fwd_persist_kernel", "costreg.conv0": "conv3d_c8p_f16x3_kernel<32",
         "costreg.conv1": "conv_s2_march_kernel", "feature.conv3+conv4": "conv2d_pair_kernel", "costreg.conv2": "SplitCfg<16, 16, 3",
         "costreg.conv4": "SplitCfg<32, 32, 3", "costreg.conv11": "DeconvSplitCfg<16, true",
         "costreg.prob": "conv3d_cout1_march_kernel", "costreg.tail": "costreg_tail_kernel", "softmax_regress_conf": "softmax_regress_conf_kernel",
         "feature.head": "feature_head_kernel", "feature.conv2": "SplitCfg<8, 16, 1, 2, 5"}
F, W = out["FETCH_SIZE_per_launch_KB"], out["WRITE_SIZE_per_launch_KB"]
def find(d, sub):
    return next((v for k, v in d.items() if sub in k), None)
traffic = {"_note": "HBM-side bytes per launch from rocprofv3 PMC passes (scripts/profile_round.sh): "
           "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  FETCH_SIZE is doubled per "
           "MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced streaming reads; "
           "Infinity-Cache hits are included; other access widths are uncalibrated); WRITE_SIZE is "
           "used as reported (it equals the algorithmic write bytes of these kernels).",
           "round": tag, "kernels": {}}
for st, sub in names.items():
    f, w = find(F, sub), find(W, sub)
    traffic["kernels"][st] = {"fetch_size_kb": f, "write_size_kb": w,
                              "traffic_bytes": None if f is None and w is None
                              else int((2 * (f or 0) + (w or 0)) * 1024)}
json.dump(traffic, open(f"{root}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
