#!/bin/bash
# conv0 tuning run on the GPU box: ablations + rocprofv3 counters (own pass, no tracing mixed in)
cd "$(dirname "$0")/.." ; mkdir -p gpurun_out
export TMPDIR=/tmp
for a in 0 1 2 4 8 3 7; do
  echo "ABLATE=$a $(MVS_CONV_ABLATE=$a python scripts/bench_kernels.py 5 conv0 2>/dev/null | tr -d '\n ')"
done > gpurun_out/conv0_ablate.log 2>&1
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU -d gpurun_out/pmc_a -o pmc_a --output-format csv -- python scripts/bench_kernels.py 2 conv0 > gpurun_out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d gpurun_out/pmc_b -o pmc_b --output-format csv -- python scripts/bench_kernels.py 2 conv0 > gpurun_out/pmc_b.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d gpurun_out/pmc_c -o pmc_c --output-format csv -- python scripts/bench_kernels.py 2 conv0 > gpurun_out/pmc_c.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in "abc":
    for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if "conv3d_mfma" not in row.get("Kernel_Name", ""):
                continue
            k = row["Counter_Name"]
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
        print(tag, {k: (v[0], v[1] / v[0]) for k, v in agg.items()})
PY
cat gpurun_out/conv0_ablate.log
