import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd())
import numpy as np
from sina_amd import synth, capi
nq = 9216
refs = synth.make_refs(20000, length=1500, width=50000, seed=2)
qs = synth.make_queries(refs, nq, seed=3)
ctx = capi.Context(0)
ctx.upload_refs(refs.ab, refs.off, refs.width)
ctx.build_index(10, False)
ids, sc, n = ctx.kmer_topk(qs.mask, qs.off, 40)
fam = [np.asarray(ids[q, :n[q]], np.uint32) for q in range(nq)]
foff = np.zeros(nq + 1, np.uint64); foff[1:] = np.cumsum([len(f) for f in fam])
masks = (qs.mask & 0x0f).astype(np.uint8)
fid = np.concatenate(fam)
for add in (0, -20, -40, -50, -60, -80):
    os.environ["SINA_HIP_TEST"] = "scout_add=%d" % add
    for rep in range(2):
        s0 = ctx.stats()
        ctx.align_families(fid, foff, masks, qs.off, ctx.params())
        s1 = ctx.stats()
    print("scout_add %4d: dp %.2f ms  rows swept %.3f  second attempts %d" % (add, s1["dp_ms"] - s0["dp_ms"],
        (s1["dp_rows_swept"] - s0["dp_rows_swept"]) / max(1, s1["dp_rows"] - s0["dp_rows"]), s1["dp_second_attempts"] - s0["dp_second_attempts"]))
