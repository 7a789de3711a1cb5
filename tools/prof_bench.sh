#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args]   -- rocprofv3 summaries of the bench command itself.
# Kernel trace and the two HBM-traffic PMC passes are SEPARATE runs (guide: FETCH_SIZE needs 3 TCC
# slots, WRITE_SIZE 2; never combined with trace domains other than --kernel-trace/--stats).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py "$@" > $OUT/bench_kt.log 2>&1
timeout 400 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 bench.py "$@" > $OUT/bench_fetch.log 2>&1
timeout 400 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 bench.py "$@" > $OUT/bench_write.log 2>&1
python3 tools/prof_traffic.py $OUT
