mkdir -p gpurun_out/r06c
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
r=d['roofline']
print(sys.argv[1], round(d['value']), round(d['ms_per_step'],2), 'cores', round(d['host_cores_busy'],2), 'conf', d.get('confined_rate_frac'), 'dp', round(r['ms_per_launch'],2), 'rows', round(r['wave_rows_computed_frac'],3), '2nd', r['row_skip']['second_attempts'], {k:round(v,1) for k,v in d['stages_ms_per_step'].items()})
" $1; }
for inf in 4 6; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --confined-cpus 0 --inflight $inf > gpurun_out/r06c/b_inf$inf.json 2> gpurun_out/r06c/b_inf$inf.err; show gpurun_out/r06c/b_inf$inf.json
done
SINA_HIP_TEST="scout=0" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --confined-cpus 0 > gpurun_out/r06c/b_noscout.json 2> gpurun_out/r06c/b_noscout.err; show gpurun_out/r06c/b_noscout.json
