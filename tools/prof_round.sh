#!/bin/bash
# usage: tools/prof_round.sh <tag>  -- everything profiles/<tag>_* is made of, in one GPU session:
# the bench command under rocprofv3 (kernel trace + the two HBM-traffic PMC passes), the DP kernel's SQ
# counters and phase timers, the other BASELINE shapes with their kernel statistics, the FASTA driver, and
# plain bench lines (with and without chained launches, with repeats).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
O=gpurun_out/$TAG
mkdir -p $O
bash tools/prof_bench.sh $TAG --no-cpu-baseline --confined-cpus 0 > $O/prof_bench.log 2>&1
python3 tools/kt_union.py gpurun_out/prof_$TAG/kt/kt_kernel_trace.csv > $O/kt_union.txt 2>&1
python3 tools/heavy_gaps.py gpurun_out/prof_$TAG/kt/kt_kernel_trace.csv > $O/heavy_gaps.txt 2>&1
rm -f gpurun_out/prof_$TAG/kt/kt_kernel_trace.csv   # (tens of MB; its summaries are what is kept)
python3 tools/perf_dp.py 3072 > /dev/null 2>&1
bash tools/prof_dp_pmc.sh $TAG 3072 > $O/dp_pmc.log 2>&1
SINA_HIP_LIB=sina_amd/libsina_hip_prof1.so python3 tools/perf_dp.py 3072 > $O/dp_phase_profile.txt 2>&1
python3 tools/perf_dp.py 3072 > $O/perf_dp.txt 2>&1
python3 tools/perf_dp.py 9216 > $O/perf_dp_9216.txt 2>&1
python3 tools/perf_graph.py 3072 > $O/perf_graph.txt 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_v4 -o kt -- python3 bench.py --no-cpu-baseline --confined-cpus 0 --window 250 --batch 16384 --sub-batch 5120 > $O/v4_bench.json 2> $O/v4.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_23s -o kt -- python3 bench.py --no-cpu-baseline --confined-cpus 0 --length 3000 --width 150000 --batch 6144 --sub-batch 3072 --inflight 2 --steps 6 --warmup 1 > $O/23s_bench.json 2> $O/23s.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_500k -o kt -- python3 bench.py --no-cpu-baseline --confined-cpus 0 --refs 500000 > $O/500k_bench.json 2> $O/500k.err
rm -f $O/kt_*/kt_kernel_trace.csv
# (the same three without the profiler: its interception costs a core of kernel-mode time -- host_cores_busy of the lines above is not the pipeline's)
python3 bench.py --no-cpu-baseline --confined-cpus 0 --window 250 --batch 16384 --sub-batch 5120 > $O/v4_bench_plain.json 2>/dev/null
python3 bench.py --no-cpu-baseline --confined-cpus 0 --length 3000 --width 150000 --batch 6144 --sub-batch 3072 --inflight 2 --steps 6 --warmup 1 > $O/23s_bench_plain.json 2>/dev/null
python3 bench.py --no-cpu-baseline --confined-cpus 0 --refs 500000 > $O/500k_bench_plain.json 2>/dev/null
python3 tools/perf_fasta.py align 40000 100000 > $O/perf_fasta.txt 2>&1
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dup-rate 0.5 > $O/bench_line_dup50.json 2> $O/bench_dup50.err
SINA_HIP_CHAIN=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --confined-cpus 0 > $O/bench_line_chain0.json 2> $O/bench_chain0.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_line_plain.json 2> $O/bench_plain.err
python3 bench.py > $O/bench_line_default.json 2> $O/bench_default.err
# round 6: the scout pass alone and against the store's guess, a launch that mixes divergences (with and without the scout)
python3 tools/perf_scout.py 9216 > $O/perf_scout.txt 2>&1
SINA_HIP_TEST="scout=0" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --confined-cpus 0 > $O/bench_line_noscout.json 2> $O/bench_noscout.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --confined-cpus 0 --divergence-mix > $O/bench_line_mix.json 2> $O/bench_mix.err
SINA_HIP_TEST="scout=0" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --confined-cpus 0 --divergence-mix > $O/bench_line_mix_noscout.json 2> $O/bench_mix_noscout.err
SINA_HOST_PROFILE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --confined-cpus 0 > $O/bench_line_hostprof.json 2> $O/host_profile.txt
# exact relatives (the aligner's copy short-cut: no DP for them, a string search per family member on the host)
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --confined-cpus 0 --exact-rate 0.3 > $O/bench_line_exact30.json 2> $O/bench_exact30.err
python3 bench.py --no-cpu-baseline --confined-cpus 0 --window 250 --batch 16384 --sub-batch 5120 --exact-rate 0.3 > $O/v4_bench_exact30.json 2> $O/v4_exact30.err
ls -la $O
