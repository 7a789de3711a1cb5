"""k-mer count at a large reference count against the dense-list threshold (SINA_HIP_TEST=dense_div=N: lists longer
than n_refs / N are kept as bitmaps too).  usage: tools/perf_kmer_dense.py [n_queries] [n_refs] [div,div,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sina_amd import capi, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
nrefs = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
divs = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "32,16,64,128,256").split(",")]
refs = synth.make_refs(nrefs, length=1500, width=50000, seed=2)
qs = synth.make_queries(refs, nq, seed=3)
want = None
for div in divs:
    os.environ["SINA_HIP_TEST"] = "dense_div=%d" % div
    ctx = capi.Context(0)
    ctx.upload_refs(refs.ab, refs.off, refs.width)
    ctx.build_index(10, False)
    best = None
    for rep in range(4):
        s0 = ctx.stats()
        ids, sc, n = ctx.kmer_topk(qs.mask, qs.off.astype(np.uint64), 41)
        s1 = ctx.stats()
        cms = s1["kmer_count_ms"] - s0["kmer_count_ms"]
        best = cms if best is None else min(best, cms)
    chk = (int(ids.astype(np.uint64).sum()), float(sc.sum()))
    if want is None:
        want = chk
    print("dense_div %4d: count %.2f ms (best of 4) select %.2f ms  %s" % (
        div, best, s1["kmer_select_ms"] - s0["kmer_select_ms"], "same results" if chk == want else "RESULTS DIFFER"), flush=True)
    del ctx
