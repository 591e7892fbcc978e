#!/bin/bash
# memory-side counters of conv0 (fp32 persistent kernel and the split-operand bf16 kernel) at the headline shape,
# one counter group per pass
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  d=gpurun_out/pmc_conv_$TAG/$(echo $c | tr ' ' '_')
  rm -rf $d
  rocprofv3 --pmc $c -d $d -o p --output-format csv -- python scripts/exp_conv_split_time.py > /dev/null 2>&1
done
python scripts/pmc_summary.py gpurun_out/pmc_conv_$TAG conv3d_c8 > gpurun_out/pmc_conv_$TAG.json
rm -rf gpurun_out/pmc_conv_$TAG
