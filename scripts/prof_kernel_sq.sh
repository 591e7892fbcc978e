#!/bin/bash
# SQ counters of one kernel, each counter group in its own rocprofv3 --pmc pass (no tracing mixed in).
#   scripts/prof_kernel_sq.sh <tag> <kernel-name substring> <command...>   -> gpurun_out/pmc_sq_<tag>.json
TAG=$1; SUB=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() {  # name counters...
  local name=$1; shift
  rm -rf gpurun_out/pmc_$TAG/$name
  rocprofv3 --pmc "$@" -d gpurun_out/pmc_$TAG/$name -o p --output-format csv -- "${CMD[@]}" > /dev/null 2>&1
}
CMD=("$@")
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
python scripts/pmc_summary.py gpurun_out/pmc_$TAG "$SUB" > gpurun_out/pmc_sq_$TAG.json
rm -rf gpurun_out/pmc_$TAG
