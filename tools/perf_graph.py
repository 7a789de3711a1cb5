"""Device DAG build alone: time per launch and (profiling build) per-phase share."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sina_amd import synth, capi
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
refs = synth.make_refs(20000, length=1500, width=50000, seed=2)
qs = synth.make_queries(refs, nq, seed=3)
ctx = capi.Context(0)
ctx.upload_refs(refs.ab, refs.off, refs.width)
ctx.build_index(10, False)
ids, sc, n = ctx.kmer_topk(qs.mask, qs.off, 40)
fam = [np.asarray(ids[q, :n[q]], np.uint32) for q in range(nq)]
foff = np.zeros(nq + 1, np.uint64); foff[1:] = np.cumsum([len(f) for f in fam])
masks = (qs.mask & 0x0f).astype(np.uint8)
lib = capi.load()
for rep in range(3):
    s0 = ctx.stats()
    ctx.align_families(np.concatenate(fam), foff, masks, qs.off, ctx.params())
    s1 = ctx.stats()
    print("graph %.3f ms  dp %.2f ms  bt %.2f ms" % (s1["graph_ms"] - s0["graph_ms"], s1["dp_ms"] - s0["dp_ms"], s1["backtrack_ms"] - s0["backtrack_ms"]))
if hasattr(lib, "sina_hip_debug_graph_profile"):
    a = (ctypes.c_ulonglong * 16)()
    lib.sina_hip_debug_graph_profile(a, 1)
    names = ["bitmap", "rank+init", "tile clear", "tile fill", "tile nodes", "tile scans", "tile prev", "tile emit",
             "(after tiles)", "sinks/fence", "slot alloc", "pred encode", "skip bound"]
    tot = float(sum(a[:13]))
    for i, nme in enumerate(names):
        print("%-14s %5.1f%%  %9.0f ticks per family" % (nme, 100 * a[i] / tot, a[i] / (3.0 * nq)))
    print("total %.0f ticks per family (thread 0 of each workgroup, 3 launches)" % (tot / (3.0 * nq)))
