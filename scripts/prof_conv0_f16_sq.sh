#!/bin/bash
# SQ-side counters of the two-piece fp16 conv0 kernel at the headline shape (one counter group per --pmc pass, no tracing mixed in)
TAG=${1:-r03}
cd "$(dirname "$0")/.." ; export TMPDIR=/tmp
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  d=gpurun_out/pmc_conv0f16_$TAG/$(echo $c | tr ' ' '_' | cut -c1-60)
  rm -rf $d
  rocprofv3 --pmc $c -d $d -o p --output-format csv -- python scripts/exp_conv0_f16_run.py > /dev/null 2>&1
done
python scripts/pmc_summary.py gpurun_out/pmc_conv0f16_$TAG conv3d_c8_f16x3 > gpurun_out/pmc_conv0f16_$TAG.json
rm -rf gpurun_out/pmc_conv0f16_$TAG
