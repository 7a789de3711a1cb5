cd tools/ubench
for v in "" "ROC_ACTIVE_WAIT_TIMEOUT=0" "HSA_ENABLE_INTERRUPT=0" "ROC_CPU_WAIT_FOR_SIGNAL=0" "DEBUG_HIP_BLOCK_SYNC=1" "AMD_DIRECT_DISPATCH=0" "HSA_ENABLE_MWAITX=1"; do
  echo "=== env: ${v:-default}"
  env $v timeout 120 ./wait_cpu
done
