#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel trace + stats of the bench command,
# then HBM-traffic counters in their own passes (no tracing mixed in).
# Usage: bash scripts/profile_round.sh r01     (writes gpurun_out/prof_<tag>/...)
TAG=${1:-r01}
cd "$(dirname "$0")/.." ; mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
CMD="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
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
            k = row["Kernel_Name"].split("(")[0][:90]
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
    out[name.upper() + "_SIZE_per_launch_KB"] = {k: round(v[1] / v[0], 1) for k, v in
                                                 sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]}
json.dump(out, open(f"{root}/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
