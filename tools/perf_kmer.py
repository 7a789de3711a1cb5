"""k-mer search kernels alone on the bench workload.  usage: tools/perf_kmer.py [n_queries] [n_refs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sina_amd import capi, synth

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nrefs = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
refs = synth.make_refs(nrefs, length=1500, width=50000, seed=2)
qs = synth.make_queries(refs, nq, seed=3)
ctx = capi.Context(0)
ctx.upload_refs(refs.ab, refs.off, refs.width)
ctx.build_index(10, False)
for rep in range(4):
    s0 = ctx.stats()
    t = time.time()
    ids, sc, n = ctx.kmer_topk(qs.mask, qs.off.astype(np.uint64), 40)
    dt = time.time() - t
    s1 = ctx.stats()
    post = s1["postings"] - s0["postings"]
    cms = s1["kmer_count_ms"] - s0["kmer_count_ms"]
    print("wall %.3fs count %.2f ms select %.2f ms postings %.3g -> %.1f Gposting/s, %.2f TB/s list reads" % (
        dt, cms, s1["kmer_select_ms"] - s0["kmer_select_ms"], post, post / cms / 1e6, 4 * post / cms / 1e9))
print("checksum", int(ids.astype(np.uint64).sum()), float(sc.sum()))
