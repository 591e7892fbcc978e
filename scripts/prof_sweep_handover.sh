#!/bin/bash
# SQ counters of the hand-over sweep's persistent kernel (own --pmc passes, no tracing): gpurun_out/pmc_sweep_<tag>.json
TAG=${1:-r06}; MODE=${2:-hand}
cd "$(dirname "$0")/.." ; export TMPDIR=/tmp
run() { local name=$1; shift; rm -rf gpurun_out/pmc_sw_$TAG/$name
  rocprofv3 --pmc "$@" -d gpurun_out/pmc_sw_$TAG/$name -o p --output-format csv -- python scripts/run_sweep_once.py $MODE > /dev/null 2>&1; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
python scripts/pmc_summary.py gpurun_out/pmc_sw_$TAG variance > gpurun_out/pmc_sweep_${TAG}_$MODE.json
rm -rf gpurun_out/pmc_sw_$TAG
