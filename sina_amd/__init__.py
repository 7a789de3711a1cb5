"""MI355X-native hot path of SINA (k-mer search + mesh DP aligner) behind its stage surface."""
import os

# The pipeline keeps half a dozen HIP streams busy (aligner and k-mer search contexts, the store's heavy
# stream).  With the runtime's default of 4 hardware queues, streams share a queue and a DP launch
# waits for an unrelated backtrack / k-mer kernel queued before it (rocprofv3 kernel trace: 10 % of
# the time no DP kernel resident).  The runtime reads this once, when it starts: set here, at package
# import, and again by libsina_hip.so's load-time constructor for non-Python hosts (csrc/api.hip).
# SINA_HIP_NO_RUNTIME_DEFAULTS (anything but "0"): leave the process environment alone, here and in the library.
_NO_DEFAULTS = os.environ.get("SINA_HIP_NO_RUNTIME_DEFAULTS", "0") not in ("", "0")
if not _NO_DEFAULTS:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# the runtime's pool of completion signals (default 64) runs dry with four batches in flight: its helper thread then
# spends 0.8 of a core creating and waiting for signals (csrc/api.hip, sina_hip_runtime_defaults)
if not _NO_DEFAULTS:
    os.environ.setdefault("ROC_SIGNAL_POOL_SIZE", "1024")
