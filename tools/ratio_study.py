"""How well does the k-mer score of the best reference predict a query's optimum / bound ratio (the row skip's guess)?
usage: tools/ratio_study.py [n_queries] [n_refs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sina_amd import synth, capi
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
nrefs = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
refs = synth.make_refs(nrefs, length=1500, width=50000, seed=2)
qs = synth.make_queries(refs, nq, seed=3)
ctx = capi.Context(0)
ctx.upload_refs(refs.ab, refs.off, refs.width)
ctx.build_index(10, False)
ids, sc, n = ctx.kmer_topk(qs.mask, qs.off, 40)
fam = [np.asarray(ids[q, :n[q]], np.uint32) for q in range(nq)]
foff = np.zeros(nq + 1, np.uint64); foff[1:] = np.cumsum([len(f) for f in fam])
masks = (qs.mask & 0x0f).astype(np.uint8)
ctx.align_families(np.concatenate(fam), foff, masks, qs.off, ctx.params())
L = np.diff(qs.off).astype(np.float64)
rows = []
for q in range(nq):
    d = ctx.dp_info(q)
    if d["gain0"] > 0 and d["raw"] < 0:
        rows.append((sc[q, 0] / max(1.0, (L[q] - 10) / 4.0), sc[q, :n[q]].mean() / max(1.0, (L[q] - 10) / 4.0), -d["raw"] / d["gain0"]))
a = np.array(rows)
print("queries", len(a), "ratio min %.4f p1 %.4f p10 %.4f median %.4f p90 %.4f max %.4f" % (a[:, 2].min(), *np.percentile(a[:, 2], [1, 10, 50, 90]), a[:, 2].max()))
for col, name in ((0, "top-1 k-mer score / (L-10)/4"), (1, "mean score of the family / (L-10)/4")):
    x, y = a[:, col], a[:, 2]
    print(name, "corr %.3f" % np.corrcoef(x, y)[0, 1])
    A = np.vstack([x, np.ones_like(x)]).T
    coef, res, *_ = np.linalg.lstsq(A, y, rcond=None)
    r = y - A @ coef
    print("  fit ratio = %.3f * x + %.3f   residual sd %.4f  min %.4f p0.1 %.4f" % (coef[0], coef[1], r.std(), r.min(), np.percentile(r, 0.1)))
    # binned minima: what a per-bin guess could use
    qs_ = np.quantile(x, np.linspace(0, 1, 9))
    for b in range(8):
        m = (x >= qs_[b]) & (x <= qs_[b + 1])
        print("  bin %d  x %.3f..%.3f  n %4d  ratio min %.4f  p50 %.4f" % (b, qs_[b], qs_[b + 1], m.sum(), y[m].min(), np.median(y[m])))
