"""Which waves of a DP launch end last, and why: per wave its time (profiling build's spans) beside the query's
rows swept, attempts, DAG rows and length (sina_hip_debug_dp_info).  Run with the profiling build:
SINA_HIP_LIB=sina_amd/libsina_hip_prof1.so python tools/dp_tail.py [queries] [refs]"""
import sys, os, ctypes
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
lib = capi.load()
for rep in range(3):
    s0 = ctx.stats()
    ctx.align_families(np.concatenate(fam), foff, masks, qs.off, ctx.params())
    s1 = ctx.stats()
    print("dp %.2f ms  scout %.2f ms  rows swept %.3f  second attempts %d" % (
        s1["dp_ms"] - s0["dp_ms"], s1["scout_ms"] - s0["scout_ms"],
        (s1["dp_rows_swept"] - s0["dp_rows_swept"]) / max(1, s1["dp_rows"] - s0["dp_rows"]),
        s1["dp_second_attempts"] - s0["dp_second_attempts"]))
info = [ctx.dp_info(q) for q in range(nq)]
# DAG rows of a family = its distinct (column, character) words; the launch order is by rows x length, largest first
N = np.array([len(np.unique(np.concatenate([refs.ab[int(refs.off[r]):int(refs.off[r + 1])] for r in f]))) for f in fam], np.float64)
L = np.diff(qs.off).astype(np.float64)
order = np.array(sorted(range(nq), key=lambda q: -(N[q] * L[q])), dtype=np.int64)  # (stable, like api.hip's)
sp = (ctypes.c_ulonglong * (2 * nq))()
if not hasattr(lib, "sina_hip_debug_dp_spans") or lib.sina_hip_debug_dp_spans(sp, nq) != 0:
    raise SystemExit("no spans: not the profiling build")
t = np.array(sp, dtype=np.float64).reshape(nq, 2) / 100e3
start, end = t[:, 0] - t[:, 0].min(), t[:, 1] - t[:, 0].min()
dur = end - start
rows = np.array([info[q]["rows_swept"] for q in order], np.float64)
att = np.array([info[q]["attempts"] for q in order])
print("wave time: median %.2f ms, 99 %% %.2f, max %.2f; ms per 1000 rows swept: median %.3f, 99 %% %.3f, max %.3f" % (
    np.median(dur), np.percentile(dur, 99), dur.max(), np.median(dur / rows * 1e3), np.percentile(dur / rows * 1e3, 99), (dur / rows * 1e3).max()))
print("corr(duration, rows swept) = %.3f" % np.corrcoef(dur, rows)[0, 1])
for w in np.argsort(end)[-16:]:
    q = order[w]
    i = info[q]
    print("  wave %5d query %5d: start %.2f end %.2f (%.2f ms)  N %d L %d rows swept %d (%.2f of N x strips) attempts %d  U %.1f scout %.1f raw %.1f" % (
        w, q, start[w], end[w], dur[w], N[q], L[q], i["rows_swept"], i["rows_swept"] / (N[q] * np.ceil(L[q] / 512)), i["attempts"],
        i["ubound"], i["scout"], i["raw"]))
