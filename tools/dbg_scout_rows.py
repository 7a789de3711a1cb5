"""Per query: rows the skipping kernel swept with the scout's bound against the store's guess (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from sina_amd import synth, capi
from tests import util
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
refs = synth.make_refs(2000, length=1500, width=50000, seed=2)
qs = synth.make_queries(refs, nq, seed=3)
cs = util.cseqs_from_refs(refs)
idx = po.Index(cs, k=10)
graphs, qms = [], []
for qi in range(qs.n):
    q = util.query_cseq(qs, qi)
    ids, sc, _ = idx.famfinder(q)
    graphs.append(util.graph_dict([cs[i] for i in ids]))
    qms.append((q.packed() >> 24).astype(np.uint8))
qoff = np.zeros(nq + 1, np.uint64)
qoff[1:] = np.cumsum([len(m) for m in qms])
ctx = capi.Context(0)
gb = ctx.graph_batch(graphs, refs.width)
qm = np.concatenate(qms)
res = {}
for mode in ("scout=0", "scout=1"):
    os.environ["SINA_HIP_TEST"] = mode
    for rep in range(2):
        ctx.align_graphs(gb, qm, qoff)
    res[mode] = [ctx.dp_info(q) for q in range(nq)]
a, b = res["scout=0"], res["scout=1"]
d = np.array([b[q]["rows_swept"] - a[q]["rows_swept"] for q in range(nq)])
print("rows swept: guess %d, scout %d; queries with MORE rows under the scout: %d" % (sum(x["rows_swept"] for x in a), sum(x["rows_swept"] for x in b), int((d > 0).sum())))
for q in np.argsort(-d)[:8]:
    print("q%d N %d L %d: guess rows %d att %d U %.1f | scout rows %d att %d U %.1f scout %.2f raw %.2f gain0 %.1f" % (
        q, graphs[q]["n"], len(qms[q]), a[q]["rows_swept"], a[q]["attempts"], a[q]["ubound"], b[q]["rows_swept"], b[q]["attempts"],
        b[q]["ubound"], b[q]["scout"], b[q]["raw"], b[q]["gain0"]))
big = np.argsort([-x["rows_swept"] for x in b])[:8]
print("most rows under the scout:", [(int(q), len(qms[q]), b[q]["rows_swept"], b[q]["attempts"], round(b[q]["ubound"] - b[q]["raw"], 2)) for q in big])
