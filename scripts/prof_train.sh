#!/bin/bash
cd "$(dirname "$0")/.." ; mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o t --output-format csv -- python scripts/train_synthetic.py --steps 3 > gpurun_out/prof_train.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_train/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.1f} pct {r['Percentage']}")
PY
