#!/bin/bash
# usage: tools/r06_ab.sh [n]  -- n bench lines of the headline command (A/B runs of a library variant in one GPU session)
for i in $(seq 1 ${1:-3}); do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --confined-cpus 0 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],2), 'cores', round(d['host_cores_busy'],2), 'dp', round(r['ms_per_launch'],2), 'graph', round(d['stages_ms_per_step']['graph_kernel'],1), 'kmer', round(d['stages_ms_per_step']['kmer_count_kernel'],1), 'bt', round(d['stages_ms_per_step']['backtrack_kernel'],1))"
done
