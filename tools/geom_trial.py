"""The DP launch under another strip geometry (SINA_HIP_TEST=geom=T,B: T/64 strips of 64 x B columns), the pipeline's own
call on bench-shaped queries: time, rows swept, second attempts.   python tools/geom_trial.py [geom ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sina_amd import synth, capi
nq = 9216
refs = synth.make_refs(20000, length=1500, width=50000, seed=2)
qs = synth.make_queries(refs, nq, seed=3)
print("longest query", int(np.diff(qs.off).max()))
ctx = capi.Context(0)
ctx.upload_refs(refs.ab, refs.off, refs.width)
ctx.build_index(10, False)
ids, sc, n = ctx.kmer_topk(qs.mask, qs.off, 40)
fam = [np.asarray(ids[q, :n[q]], np.uint32) for q in range(nq)]
foff = np.zeros(nq + 1, np.uint64); foff[1:] = np.cumsum([len(f) for f in fam])
masks = (qs.mask & 0x0f).astype(np.uint8)
fid = np.concatenate(fam)
ref_out = None
for g in (sys.argv[1:] or ["", "geom=384,4", "geom=448,4", "geom=192,8"]):
    os.environ["SINA_HIP_TEST"] = g
    for rep in range(3):
        s0 = ctx.stats()
        out, pos = ctx.align_families(fid, foff, masks, qs.off, ctx.params())
        s1 = ctx.stats()
    if ref_out is None:
        ref_out = (out.copy(), pos.copy())
    same = bool((out["raw"] == ref_out[0]["raw"]).all() and (pos == ref_out[1]).all())
    print("%-14s dp %.2f ms  bt %.2f ms  rows swept %.3f  cells swept %.3f  second attempts %d  same results %s" % (g or "(default)", s1["dp_ms"] - s0["dp_ms"], s1["backtrack_ms"] - s0["backtrack_ms"],
        (s1["dp_rows_swept"] - s0["dp_rows_swept"]) / max(1, s1["dp_rows"] - s0["dp_rows"]),
        (s1["dp_cells_swept"] - s0["dp_cells_swept"]) / max(1, s1["dp_cells"] - s0["dp_cells"]),
        s1["dp_second_attempts"] - s0["dp_second_attempts"], same))
