#!/bin/bash
# L2 behaviour of the plane-sweep kernel at configs[1] (hits / misses, not FETCH_SIZE alone): each counter group in its
# own pass, no tracing mixed in.   MVS_BENCH_FAST=1 bash scripts/prof_variance_l2.sh <tag>
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/pmc_$TAG/$d
  rocprofv3 --pmc $c -d gpurun_out/pmc_$TAG/$d -o p --output-format csv -- python scripts/bench_kernels.py 2 variance_lds > /dev/null 2> gpurun_out/pmc_l2_$TAG.$d.log
done
python scripts/pmc_summary.py gpurun_out/pmc_$TAG variance > gpurun_out/pmc_l2_var_$TAG.json
rm -rf gpurun_out/pmc_$TAG
