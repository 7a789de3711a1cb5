"""DP phase profile on the bench workload (needs `make -C sina_amd/csrc PROFILE=1`).
usage: tools/perf_bench_prof.py [n_queries] [n_refs]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sina_amd import capi, pipeline, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nrefs = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
refs = synth.make_refs(nrefs, length=1500, width=50000, seed=2)
qs = synth.make_queries(refs, 2 * nq, seed=3)
store = pipeline.Store(":mem:prof", refs, device=0)
store.build_index(10, False)
pl = pipeline.Pipeline(store, aligner={"device-graph": True})
lib = capi.load()

def run(first):
    lo, hi = qs.off[first], qs.off[first + nq]
    off = (qs.off[first:first + nq + 1] - lo).astype(np.uint64)
    s0 = store.stats()
    pl.run(qs.mask[lo:hi], off, batch=2048, inflight=1)
    s1 = store.stats()
    d = {k: s1[k] - s0[k] for k in s1}
    print("dp %.2f ms  %.3g cells  %.1f Gcell/s  launches %d" % (
        d["dp_ms"], d["dp_cells"], d["dp_cells"] / d["dp_ms"] / 1e6, d["dp_launches"]))

run(0)
a = (ctypes.c_ulonglong * 32)()
if hasattr(lib, "sina_hip_debug_dp_profile"):
    lib.sina_hip_debug_dp_profile(a, 1)
for rep in range(int(os.environ.get("REPS", "1"))):
    run(nq)
if hasattr(lib, "sina_hip_debug_dp_profile"):
    lib.sina_hip_debug_dp_profile(a, 1)
    names = ["setup", "handshake", "far preds", "near preds", "chain+verify", "rerun", "publish", "tb+end"]
    tot = float(sum(a[:8]))
    rows = a[8]
    for i, n in enumerate(names):
        print("%-14s %5.1f%%  %8.0f ticks/row" % (n, 100 * a[i] / tot, a[i] / rows))
    print("wave-rows %d  far preds/row %.3f  rerun iters/row %.3f  spill rows/row %.3f" % (
        rows, a[9] / rows, a[10] / rows, a[11] / rows))
    print("rerun iterations/row histogram [0,1,2,3,4,5-8,9-16,17-32,33+]:",
          " ".join("%.3f" % (a[16 + i] / rows) for i in range(9)))
