#!/bin/bash
# effective shader clock per kernel of an arbitrary command: GRBM_GUI_ACTIVE cycles / kernel duration
#   scripts/prof_clock_cmd.sh <tag> <command...>
TAG=$1; shift
cd "$(dirname "$0")/.." ; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_clk_$TAG
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_clk_$TAG -o clk --output-format csv -- "$@" > gpurun_out/pmc_clk_$TAG.log 2>&1
python - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
cnt = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmc_clk_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[(r["Kernel_Name"][:70], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(f"gpurun_out/pmc_clk_{tag}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for (k, d), v in cnt.items():
    if d in dur and dur[d] > 50000:
        agg[k].append((sum(v), dur[d]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(b for _, b in kv[1])):
    cyc = sum(a for a, _ in v) / len(v); ns = sum(b for _, b in v) / len(v)
    print(f"{tag} {k:70s} n={len(v)} cycles={cyc:.3e} ns={ns:.0f} clock_GHz={cyc/ns:.3f}")
PY
rm -rf gpurun_out/pmc_clk_$TAG
