#!/bin/bash
# variance (LDS sweep) kernel tuning: rocprofv3 SQ counters, own passes, no tracing mixed in
cd "$(dirname "$0")/.." ; mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU -d gpurun_out/pmc_var_a -o p --output-format csv -- python scripts/bench_kernels.py 2 variance_lds > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_var_b -o p --output-format csv -- python scripts/bench_kernels.py 2 variance_lds > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY TCP_TCC_READ_REQ_sum TCC_HIT_sum -d gpurun_out/pmc_var_c -o p --output-format csv -- python scripts/bench_kernels.py 2 variance_lds > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmc_var_[abc]/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "variance_fwd_lds" not in row.get("Kernel_Name", ""):
            continue
        agg[row["Counter_Name"]][0] += 1
        agg[row["Counter_Name"]][1] += float(row["Counter_Value"])
print({k: f"{v[1] / v[0]:.4g}" for k, v in sorted(agg.items())})
PY
