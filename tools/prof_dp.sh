#!/bin/bash
# usage: tools/prof_dp.sh <nq> [tag]   (run on the GPU box through gpurun; PMC passes are separate runs)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NQ=${1:-256}
TAG=${2:-dp}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python3 tools/perf_dp.py $NQ > $OUT/kt.log 2>&1
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INSTS_WAVE32_LDS SQ_LDS_UNALIGNED_STALL"; do
  n=$(echo $c | awk '{print $1}')
  rocprofv3 --output-format csv --pmc $c -d $OUT/pmc_$n -o pmc -- python3 tools/perf_dp.py $NQ > $OUT/pmc_$n.log 2>&1
done
python3 tools/prof_summary.py $OUT
