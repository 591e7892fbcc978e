#!/bin/bash
# conv0 tuning: rocprofv3 SQ counters (own passes, no tracing mixed in)
cd "$(dirname "$0")/.." ; mkdir -p gpurun_out; export TMPDIR=/tmp
run() {  # tag, env...
  tag=$1; shift
  env "$@" rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU -d gpurun_out/pmc_${tag}a -o p --output-format csv -- python scripts/bench_kernels.py 2 conv0 > /dev/null 2>&1
  env "$@" rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_${tag}b -o p --output-format csv -- python scripts/bench_kernels.py 2 conv0 > /dev/null 2>&1
  python - $tag <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"gpurun_out/pmc_{tag}[ab]/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "conv3d_mfma" not in row.get("Kernel_Name", ""):
            continue
        agg[row["Counter_Name"]][0] += 1
        agg[row["Counter_Name"]][1] += float(row["Counter_Value"])
print(tag, {k: f"{v[1] / v[0]:.4g}" for k, v in sorted(agg.items())})
PY
}
run base MVS_X=0
run abl7 MVS_CONV_ABLATE=7
run db MVS_CONV0_VARIANT=9
