import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from sina_amd import synth, capi
from tests import util
refs = synth.make_refs(400, length=300, width=3000, seed=11, amb_rate=0.01, lower_rate=0.02)
qs = synth.make_queries(refs, 12, seed=12, amb_rate=0.01)
cs = util.cseqs_from_refs(refs)
idx = po.Index(cs, k=10)
ctx = capi.Context(0)
ctx.upload_refs(refs.ab, refs.off, refs.width)
off, ids = idx.csr()
ctx.upload_index(10, False, off, ids)
print("uploaded", flush=True)
for qi in range(3):
    t = time.time()
    s = ctx.kmer_scores(qs.seq(qi))
    print("scores", qi, time.time() - t, (s == idx.scores(util.query_cseq(qs, qi))).all(), flush=True)
gi, gs, gn = ctx.kmer_topk(qs.mask, qs.off, 41)
print("topk ok", gn, flush=True)
