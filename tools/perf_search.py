"""Search-stage comparison kernel alone at bench scale: 1000 k-mer candidates per aligned query against a
100 k-reference store.  Prints one JSON line: pairs/s, candidate bases/s, achieved HBM GB/s
(algorithmic bytes = 4 B per candidate base) and the oracle's rate on a few queries."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sina_amd import synth, capi

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_refs = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
ncand = 1000
refs = synth.make_refs(n_refs, length=1500, width=50000, seed=2)
rng = np.random.default_rng(5)
src = rng.integers(0, refs.n, size=nq)
qs_ab, masks = [], []
for i in src:                                  # aligned queries: a reference with 3 % substitutions
    ab = refs.seq(int(i)).copy()
    sub = rng.random(len(ab)) < 0.03
    ab[sub] = (ab[sub] & 0xFFFFFF) | (rng.choice([1, 2, 4, 8], size=int(sub.sum())).astype(np.uint32) << 24)
    qs_ab.append(ab)
    masks.append(((ab >> 24) & 0x0f).astype(np.uint8))
q_off = np.zeros(nq + 1, np.uint64); q_off[1:] = np.cumsum([len(x) for x in qs_ab])
ctx = capi.Context(0)
ctx.upload_refs(refs.ab, refs.off, refs.width)
ctx.build_index(10, False)
ids, sc, n = ctx.kmer_topk(np.concatenate(masks), q_off, ncand)
cand = np.concatenate([ids[q, :n[q]] for q in range(nq)]).astype(np.uint32)
c_off = np.zeros(nq + 1, np.uint64); c_off[1:] = np.cumsum(n)
flat = np.concatenate(qs_ab)
ctx.compare(flat, q_off, cand, c_off, 0, False)          # warm-up (buffers)
s0 = ctx.stats()
reps = 5
t = time.time()
for _ in range(reps):
    got = ctx.compare(flat, q_off, cand, c_off, 0, False)
wall = (time.time() - t) / reps
s1 = ctx.stats()
ms = (s1["compare_ms"] - s0["compare_ms"]) / reps
bases = (s1["compare_bases"] - s0["compare_bases"]) / reps
out = {"kernel": "compare_kernel", "queries_per_launch": nq, "candidates_per_query": ncand, "n_refs": n_refs,
       "ms_per_launch": ms, "wall_ms_per_call": 1e3 * wall, "pairs_per_s": len(cand) / (ms * 1e-3),
       "roofline": {"bound": "hbm", "achieved": 4 * bases / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": 4 * bases / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": 4 * bases}}
try:  # CPU baseline: the oracle's literal traverse on a bounded sample (test infrastructure, not the product)
    from oracle import pyoracle as po
    from tests import util
    take = 8
    need = sorted({int(r) for q in range(take) for r in ids[q, :n[q]]})
    cs = {r: po.Cseq.from_packed("ref%d" % r, refs.seq(r), refs.width) for r in need}
    t = time.time(); pairs = 0
    for q in range(take):
        qc = po.Cseq.from_packed("q", qs_ab[q], refs.width)
        for x, r in enumerate(ids[q, :n[q]]):
            want = po.compare_counts(qc, cs[int(r)])
            assert tuple(got[int(c_off[q]) + x]) == want
            pairs += 1
    dt = time.time() - t
    out["cpu_baseline"] = {"value": pairs / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
                           "sample": "%d queries x %d candidates through the oracle's traverse() via ctypes "
                                     "(includes the Python call overhead), all equal to the GPU counters" % (take, ncand)}
except ImportError:
    pass
print(json.dumps(out))
