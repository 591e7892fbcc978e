#!/bin/bash
# HBM-side traffic of the plane-sweep kernel at configs[1]: FETCH_SIZE and WRITE_SIZE in their own passes
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$TAG/$c
  rocprofv3 --pmc $c -d gpurun_out/pmc_$TAG/$c -o p --output-format csv -- python scripts/bench_kernels.py 2 variance_lds > /dev/null 2>&1
done
python scripts/pmc_summary.py gpurun_out/pmc_$TAG variance > gpurun_out/pmc_traffic_var_$TAG.json
rm -rf gpurun_out/pmc_$TAG
