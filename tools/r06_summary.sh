#!/bin/bash
# usage: tools/r06_summary.sh <dir>  -- one line per bench JSON of a tools/prof_round.sh session
cd ${1:-gpurun_out/r06}
for f in bench_line_default bench_line_plain bench_line_noscout bench_line_mix bench_line_mix_noscout bench_line_dup50 bench_line_chain0 v4_bench 23s_bench 500k_bench v4_bench_plain 23s_bench_plain 500k_bench_plain bench_line_exact30 v4_bench_exact30 bench_line_hostprof; do python3 -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]+'.json'))
except Exception as e:
    print(sys.argv[1], 'ERR', e); sys.exit()
r=d['roofline']; rs=r['row_skip']
print('%-24s %7.0f seq/s %6.2f ms/step cores %.2f (kernel %.2f) conf %s faults/s %.0f | dp %.2f ms frac %.3f rows %.3f cells %.3f 2nd %d full %d rho %.3f scout %.1f ms | %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['host_cores_busy'], d['host_cores_busy_kernel_mode'], d.get('confined_rate_frac') and round(d['confined_rate_frac'],3), d['host_minor_faults_per_s'], r['ms_per_launch'], r['frac'], r['wave_rows_computed_frac'], r['cells_computed_frac'], rs['second_attempts'], rs['full_sweeps'], rs['guess_rho'], rs.get('scout_ms_per_launch',0), {k:round(v,1) for k,v in d['kernels_ms_per_step_isolated'].items()}))
" $f; done
