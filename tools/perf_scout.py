"""The scout pass in the pipeline's own call (sina_hip_align_families): its time per launch, the rows the skipping kernel
sweeps with and without it, and per query its value against the optimum the DP found.
  python tools/perf_scout.py [queries] [substitution rate]     (one GPU; 2000 references)"""
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
fams, qms = [], []
for qi in range(qs.n):
    q = util.query_cseq(qs, qi)
    ids, sc, _ = idx.famfinder(q)
    fams.append(np.asarray(ids, np.uint32))
    qms.append((q.packed() >> 24).astype(np.uint8))
while len(fams) < nq:  # (more queries than prepared: repeats -- the kernels do not know)
    fams.append(fams[len(fams) % qs.n])
    qms.append(qms[len(qms) % qs.n])
qoff = np.zeros(nq + 1, np.uint64)
qoff[1:] = np.cumsum([len(m) for m in qms])
foff = np.zeros(nq + 1, np.uint64)
foff[1:] = np.cumsum([len(f) for f in fams])
ctx = capi.Context(0)
ctx.upload_refs(refs.ab, refs.off, refs.width)
qm = np.concatenate(qms)
fid = np.concatenate(fams)
info = {}
for mode in ("scout=0", "scout=1"):
    os.environ["SINA_HIP_TEST"] = mode
    for rep in range(3):
        s0 = ctx.stats()
        t = time.time()
        out, pos = ctx.align_families(fid, foff, qm, qoff)
        dt = time.time() - t
        st = ctx.stats()
        rows = st["dp_rows"] - s0["dp_rows"]
        print("%s: wall %.3f s  scout %.2f ms (%d launches)  dp %.2f ms  rows swept %.3f  second attempts %d  full %d" % (
            mode, dt, st["scout_ms"] - s0["scout_ms"], st["scout_launches"] - s0["scout_launches"], st["dp_ms"] - s0["dp_ms"],
            (st["dp_rows_swept"] - s0["dp_rows_swept"]) / max(rows, 1), st["dp_second_attempts"] - s0["dp_second_attempts"],
            st["dp_full_sweeps"] - s0["dp_full_sweeps"]))
    info[mode] = [ctx.dp_info(q) for q in range(min(nq, 256))]
b = info["scout=1"]
diff = np.array([x["scout"] - x["raw"] for x in b])
print("scout - optimum over %d queries: median %.1f, 90 %% %.1f, max %.1f, below the optimum %d; bound used - optimum: median %.1f max %.1f" % (
    len(diff), np.median(diff), np.percentile(diff, 90), diff.max(), int((diff < 0).sum()),
    np.median([x["ubound"] - x["raw"] for x in b]), max(x["ubound"] - x["raw"] for x in b)))
rs = np.array([x["rows_swept"] for x in b], dtype=np.float64)
print("rows swept per query under the scout: median %.0f, max %.0f (%.2f x the median)" % (np.median(rs), rs.max(), rs.max() / np.median(rs)))
