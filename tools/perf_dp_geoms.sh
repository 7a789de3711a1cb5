#!/bin/bash
# usage: tools/perf_dp_geoms.sh "T,B:nq" ...  -- DP kernel alone on one launch of nq queries, per geometry override
export SINA_HIP_TB_GB=${SINA_HIP_TB_GB:-72}
for x in "$@"; do
  g=${x%%:*}; nq=${x##*:}
  echo "== SINA_HIP_TEST=geom=$g nq=$nq"
  SINA_HIP_TEST=geom=$g python3 tools/perf_dp.py $nq 2>&1 | tail -2
done
