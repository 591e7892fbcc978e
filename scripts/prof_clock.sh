#!/bin/bash
# effective shader clock per kernel: GRBM_GUI_ACTIVE cycles / kernel duration
cd "$(dirname "$0")/.." ; mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_clk -o clk --output-format csv -- python scripts/bench_kernels.py 3 ${1:-conv0} > gpurun_out/pmc_clk.log 2>&1
python - <<'PY'
import csv, glob, collections
cnt = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[(r["Kernel_Name"][:60], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob("gpurun_out/pmc_clk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for (k, d), v in cnt.items():
    if d in dur and dur[d] > 50000:
        agg[k].append((sum(v), dur[d]))
for k, v in agg.items():
    cyc = sum(a for a, _ in v) / len(v); ns = sum(b for _, b in v) / len(v)
    print(f"{k:60s} n={len(v)} cycles={cyc:.3e} ns={ns:.0f} clock_GHz={cyc/ns:.3f}")
PY
