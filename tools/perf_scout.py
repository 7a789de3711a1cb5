"""The scout pass alone (scout.hip): its time per launch and, per query, its value against the optimum the DP found.
  python tools/perf_scout.py [queries]     (one GPU; the 2000-reference workload of tools/perf_dp.py)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from sina_amd import synth, capi  # noqa: E402
from tests import util  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sub = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
refs = synth.make_refs(2000, length=1500, width=50000, seed=2)
qs = synth.make_queries(refs, min(nq, 256), seed=3, sub=sub)
cs = util.cseqs_from_refs(refs)
idx = po.Index(cs, k=10)
graphs, qms = [], []
for qi in range(qs.n):
    q = util.query_cseq(qs, qi)
    ids, sc, _ = idx.famfinder(q)
    graphs.append(util.graph_dict([cs[i] for i in ids]))
    qms.append((q.packed() >> 24).astype(np.uint8))
while len(graphs) < nq:  # (more queries than prepared: repeats -- the kernels do not know)
    graphs.append(graphs[len(graphs) % qs.n])
    qms.append(qms[len(qms) % qs.n])
qoff = np.zeros(nq + 1, np.uint64)
qoff[1:] = np.cumsum([len(m) for m in qms])
ctx = capi.Context(0)
gb = ctx.graph_batch(graphs, refs.width)
qm = np.concatenate(qms)
for rep in range(3):
    s0 = ctx.stats()
    t = time.time()
    out, pos = ctx.align_graphs(gb, qm, qoff)
    dt = time.time() - t
    st = ctx.stats()
    rows = st["dp_rows"] - s0["dp_rows"]
    print("wall %.3f s  scout %.2f ms (%d launches)  dp %.2f ms  rows swept %.3f  second attempts %d  full %d" % (
        dt, st["scout_ms"] - s0["scout_ms"], st["scout_launches"] - s0["scout_launches"], st["dp_ms"] - s0["dp_ms"],
        (st["dp_rows_swept"] - s0["dp_rows_swept"]) / max(rows, 1), st["dp_second_attempts"] - s0["dp_second_attempts"],
        st["dp_full_sweeps"] - s0["dp_full_sweeps"]))
n = min(nq, 64)
diff = []
for q in range(n):
    i = ctx.dp_info(q)
    diff.append(i["scout"] - i["raw"])
diff = np.array(diff)
print("scout - optimum over %d queries: exact %d, max %.2f, the values: %s" % (n, int((diff == 0).sum()), diff.max(), np.round(diff[:32], 2).tolist()))
