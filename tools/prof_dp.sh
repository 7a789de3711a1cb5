#!/bin/bash
# usage: tools/prof_dp.sh <nq>   (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NQ=${1:-128}
mkdir -p gpurun_out/prof
rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/prof/kt -o kt -- python3 tools/perf_dp.py $NQ > gpurun_out/prof/kt.log 2>&1
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INSTS_WAVE32_LDS SQ_LDS_UNALIGNED_STALL"; do
  n=$(echo $c | awk '{print $1}')
  rocprofv3 --output-format csv --pmc $c -d gpurun_out/prof/pmc_$n -o pmc -- python3 tools/perf_dp.py $NQ > gpurun_out/prof/pmc_$n.log 2>&1
done
find gpurun_out/prof -name "*.csv" | head -30
