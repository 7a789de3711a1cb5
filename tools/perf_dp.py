import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from sina_amd import synth, capi
from tests import util
import pickle
nq=int(sys.argv[1]) if len(sys.argv)>1 else 256
refs=synth.make_refs(2000, length=1500, width=50000, seed=2)
t=time.time()
# the oracle-side preparation, reused by later runs of the same session: kept in a directory of this user's own
# (mode 0700; not a fixed name in a world-writable /tmp) and keyed by everything it is computed from -- the
# generator, the oracle and the helpers -- so that a stale file is never taken for the current workload
import hashlib
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h=hashlib.sha1(("refs=2000,1500,50000,2;queries=%d,3" % nq).encode())
for _f in ("sina_amd/synth.py","tests/util.py","oracle/pyoracle.py","oracle/sina_oracle.c","oracle/sina_oracle.h"):
    _h.update(open(os.path.join(ROOT,_f),"rb").read())
import tempfile
_dir=os.path.join(tempfile.gettempdir(),"sina_amd_cache_%d" % os.getuid())
os.makedirs(_dir,mode=0o700,exist_ok=True)
if os.stat(_dir).st_uid!=os.getuid() or (os.stat(_dir).st_mode&0o077): raise SystemExit("%s is not a private directory of this user" % _dir)
cache=os.path.join(_dir,"perf_dp_prep_%d_%s.pkl" % (nq,_h.hexdigest()[:12]))
if os.path.exists(cache):
    graphs,qms=pickle.load(open(cache,"rb"))
else:
    qs=synth.make_queries(refs, nq, seed=3)
    cs=util.cseqs_from_refs(refs)
    idx=po.Index(cs,k=10)
    graphs=[];qms=[]
    for qi in range(nq):
        q=util.query_cseq(qs,qi)
        ids,sc,_=idx.famfinder(q)
        graphs.append(util.graph_dict([cs[i] for i in ids]))
        qms.append((q.packed()>>24).astype(np.uint8))
    pickle.dump((graphs,qms),open(cache,"wb"))
print("prep",time.time()-t)
qoff=np.zeros(nq+1,np.uint64); qoff[1:]=np.cumsum([len(m) for m in qms])
ctx=capi.Context(0)
gb=ctx.graph_batch(graphs, refs.width)
qm=np.concatenate(qms)
import ctypes
lib = capi.load()
prof = hasattr(lib, "sina_hip_debug_dp_profile")
masks = [int(x) for x in os.environ.get("SINA_DP_ABL", "0").split(",")] if prof else [0]
for mask in masks:
    if prof:
        lib.sina_hip_debug_dp_ablate(mask)
    for rep in range(3):
        s0 = ctx.stats()
        t = time.time(); out, pos = ctx.align_graphs(gb, qm, qoff); dt = time.time() - t
        st = ctx.stats()
        dp = st['dp_ms'] - s0['dp_ms']; cells = st['dp_cells'] - s0['dp_cells']
        print("abl %d wall %.3fs dp %.2f ms bt %.2f ms cells %.0f  -> %.1f Gcell/s  q/s(dp) %.0f" % (
            mask, dt, dp, st['backtrack_ms'] - s0['backtrack_ms'], cells, cells / dp / 1e6, nq / dp * 1e3))
        rows = st['dp_rows'] - s0['dp_rows']
        if rows:  # the certified row skip (SINA_HIP_DP_PRUNE=0: off; SINA_HIP_TEST=rho=X: the guess)
            print("      rows swept %.3f of %d  cells swept %.3f  second attempts %d  full sweeps %d  of %d queries  next guess %.3f" % (
                (st['dp_rows_swept'] - s0['dp_rows_swept']) / rows, rows, (st['dp_cells_swept'] - s0['dp_cells_swept']) / cells,
                st['dp_second_attempts'] - s0['dp_second_attempts'], st['dp_full_sweeps'] - s0['dp_full_sweeps'],
                st['dp_queries_pruned'] - s0['dp_queries_pruned'], st['dp_prune_rho']))
# profiling build (make -C sina_amd/csrc PROFILE=1): per-phase share of wave time
if prof:
    a = (ctypes.c_ulonglong * 32)()
    lib.sina_hip_debug_dp_profile(a, 1)
    names = ["scalar prefetch issue", "row setup + match scores", "-", "preds", "chain first pass", "propagate+scan+overwrite", "wait scalars + publish + prefetch", "tb+end"]
    tot = float(sum(a[:8]))
    rows = a[8]
    for i, n in enumerate(names):
        print("%-14s %5.1f%%  %8.0f ticks/row" % (n, 100 * a[i] / tot, a[i] / rows))
    print("wave-rows %d  far preds/row %.3f  rerun iters/row %.3f  spill rows/row %.3f" % (
        rows, a[9] / rows, a[10] / rows, a[11] / rows))
    print("simple kernel: rows with an entering gap %.3f, chain iterations/row %.3f, log-step scans/row %.3f, rows with >= 2 predecessors %.3f, predecessors/row %.3f, rows kept in a slot %.3f, sink rows %.4f, skipped rows visited per swept row %.3f" % (
        a[12] / rows, a[13] / rows, a[14] / rows, a[15] / rows, a[25] / rows, a[26] / rows, a[27] / rows, a[28] / rows))
    print("(simple kernel: rows that reached chain iteration 1, 2, ... 8, 9+, per row / general kernel:) rerun iterations/row histogram [0,1,2,3,4,5-8,9-16,17-32,33+]:",
          " ".join("%.3f" % (a[16 + i] / rows) for i in range(9)))
# ... and the launch's drain: when the waves of the last launch started and ended (100 MHz clock)
if prof and hasattr(lib, "sina_hip_debug_dp_spans") and nq <= 16384:
    sp = (ctypes.c_ulonglong * (2 * nq))()
    if lib.sina_hip_debug_dp_spans(sp, nq) == 0:
        t = np.array(sp, dtype=np.float64).reshape(nq, 2) / 100e3  # ms
        t0 = t[:, 0].min()
        start, end = t[:, 0] - t0, t[:, 1] - t0
        span = end.max()
        cells = np.array(sorted((g["n"] * len(m) for g, m in zip(graphs, qms)), reverse=True), dtype=np.float64)  # launch order: largest first
        dur = end - start
        print("launch %.2f ms; waves started: first %.2f .. last %.2f ms; ended: 1%% %.2f  50%% %.2f  90%% %.2f  99%% %.2f  last %.2f ms" % (
            span, start.min(), start.max(), *np.percentile(end, [1, 50, 90, 99]), end.max()))
        for back in (8, 6, 4, 3, 2, 1, 0.5):
            print("  resident waves %.1f ms before the end: %d" % (back, int(((start <= span - back) & (end > span - back)).sum())))
        rate = cells / dur  # cells per ms of a wave
        print("  cells per wave-ms: median %.0f, 5%% %.0f, 95%% %.0f; of the last 3072 waves to start: median %.0f" % (
            np.median(rate), *np.percentile(rate, [5, 95]), np.median(rate[np.argsort(start)[-3072:]])))
        lastq = np.argsort(end)[-10:]
        print("  the ten waves that end last: launch index", lastq.tolist(), "durations", np.round(dur[lastq], 2).tolist(), "started", np.round(start[lastq], 2).tolist())
