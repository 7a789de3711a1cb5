# bench.py under runtime environment variants: rate, host cores and the busiest threads of the timed region
#   bash tools/ubench/bench_env_matrix.sh "ROC_SIGNAL_POOL_SIZE=64" "ROC_SIGNAL_POOL_SIZE=1024" ...
for v in "$@"; do
  echo "=== env: $v"
  env $v SINA_HOST_PROFILE=1 python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/envm.json 2> gpurun_out/envm.err
  python - <<'PY'
import json
t=open("gpurun_out/envm.json").read().strip().splitlines()
j=json.loads([l for l in t if l.startswith("{")][-1])
print("seq/s %d  host cores %.2f  kernel-mode %.2f" % (j["value"], j["host_cores_busy"], j["host_cores_busy_kernel_mode"]))
PY
  grep "timed region" gpurun_out/envm.err | head -2
  grep "timed region: whole" gpurun_out/envm.err
done
