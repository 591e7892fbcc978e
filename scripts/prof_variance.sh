#!/bin/bash
# variance kernel tuning: rocprofv3 SQ counters in their own passes (no tracing mixed in), summarised
# per kernel into gpurun_out/pmc_var_<tag>.json.   scripts/prof_variance.sh <tag>   (env selects the variant)
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() {  # name counters...
  local name=$1; shift
  rm -rf gpurun_out/pmc_$TAG/$name
  rocprofv3 --pmc "$@" -d gpurun_out/pmc_$TAG/$name -o p --output-format csv -- python scripts/bench_kernels.py 2 variance_lds > /dev/null 2>&1
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
python scripts/pmc_summary.py gpurun_out/pmc_$TAG variance > gpurun_out/pmc_var_$TAG.json
rm -rf gpurun_out/pmc_$TAG
