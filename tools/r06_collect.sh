#!/bin/bash
# usage: tools/r06_collect.sh  -- copies what a tools/prof_round.sh r06 session left under gpurun_out/ (scratch) into
# profiles/r06_* (tracked): bench lines, kernel statistics, HBM traffic, the DP kernel's counters and phase profile.
cd "$(dirname "$0")/.."
G=gpurun_out
P=profiles
cp $G/prof_r06/kt/kt_kernel_stats.csv $P/r06_bench_kernel_stats.csv
cp $G/prof_r06/summary.txt $P/r06_bench_summary.txt
cp $G/prof_r06/traffic.json $P/r06_traffic.json
cp $G/pmc_r06/dp_valu.json $P/r06_dp_valu.json
cp $G/pmc_r06/summary.txt $P/r06_dp_sq_counters.txt
for f in bench_line_default bench_line_plain bench_line_noscout bench_line_mix bench_line_mix_noscout bench_line_dup50 bench_line_chain0 bench_line_hostprof v4_bench 23s_bench 500k_bench v4_bench_plain 23s_bench_plain 500k_bench_plain bench_line_exact30 v4_bench_exact30; do cp $G/r06/$f.json $P/r06_$f.json; done
for f in dp_phase_profile heavy_gaps kt_union host_profile perf_dp perf_dp_9216 perf_fasta perf_graph perf_scout; do cp $G/r06/$f.txt $P/r06_$f.txt; done
for s in v4 23s 500k; do cp $G/r06/kt_$s/kt_kernel_stats.csv $P/r06_${s}_kernel_stats.csv; done
bash tools/r06_summary.sh $G/r06 > $P/r06_bench_lines_summary.txt
