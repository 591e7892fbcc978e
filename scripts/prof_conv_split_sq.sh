#!/bin/bash
# SQ-side counters of conv0 (fp32 persistent kernel and the split-operand bf16 kernel) at the headline shape
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"; do
  d=gpurun_out/pmc_convsq_$TAG/$(echo $c | tr ' ' '_' | cut -c1-60)
  rm -rf $d
  rocprofv3 --pmc $c -d $d -o p --output-format csv -- python scripts/exp_conv_split_time.py > /dev/null 2>&1
done
python scripts/pmc_summary.py gpurun_out/pmc_convsq_$TAG conv3d_c8 > gpurun_out/pmc_convsq_$TAG.json
rm -rf gpurun_out/pmc_convsq_$TAG
