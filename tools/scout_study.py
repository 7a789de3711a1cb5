"""CPU study for the scout pass.  Two candidates for the bound U of the certified row skip,
both costs of real paths: (a) scout(): a banded copy of the recurrence over the WHOLE DAG -- K columns per row,
re-centred on the best predecessor's minimum (a_credit, rate_frac: how rows that have consumed different numbers of
query bases are compared) -- which finds the optimum itself (built on the GPU, measured, dropped: scout.hip's
header); (b) chain(): the query against the chain of ONE family member (--chain: members 0, 1, 2), what scout.hip
does now.  For a handful of bench-shaped queries: the oracle's optimum V*, the scout's
value at several K, and how wide the run of cells at or below T(m, s) is per row at U = scout (what a single
band-following sweep would have to hold).  Test infrastructure: uses the oracle.

  python tools/scout_study.py [--refs 20000] [--queries 6] [--sub 0.03]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from sina_amd import synth  # noqa: E402
from tests import util  # noqa: E402

DEAD = np.float32(1e6)


def scout(g, qmask, K, ms=-2.0, mms=1.0, gp=5.0, gpe=2.0, a_credit=0.0, rate_frac=0.0):
    """Banded recurrence (float32, the reference's operators); returns (value, centre column per row)."""
    f = np.float32
    ms, mms, gp, gpe = f(ms), f(mms), f(gp), f(gpe)
    N, L = g["n"], len(qmask)
    po_, pr = g["pred_off"], g["pred"]
    c_of = np.zeros(N, np.int64)
    V = np.full((N, K), DEAD, np.float32)
    G = np.full((N, K), DEAD, np.float32)
    amin = np.zeros(N, np.int64)
    vmin = np.full(N, DEAD, np.float32)
    vraw = np.full(N, DEAD, np.float32)
    snk = np.zeros(N, bool)
    snk[g["snk"]] = True
    best = np.float32(np.inf)
    kk = np.arange(K)
    c_prev = 0
    for m in range(N):
        preds = pr[po_[m]:po_[m + 1]]
        if len(preds) == 0:
            c = c_prev
        else:
            # (which predecessor to follow: rows that have consumed different numbers of query bases compare by value
            # plus what the bases one of them is behind would have gained at a fraction of the path's own rate so far)
            rate = rate_frac * max(0.0, -float(vraw[m - 1])) / max(1.0, float(amin[m - 1])) if m > 0 else 0.0
            pb = preds[np.argmin(vraw[preds] + np.float32(rate) * amin[preds])]
            c = int(amin[pb]) + 1 - K // 2
        c = max(0, min(c, max(0, L - K)))
        cols = c + kk
        inside = cols < L
        q = np.where(inside, qmask[np.minimum(cols, L - 1)] & 0xf, 0)
        wgt = g["weight"][m]
        csel = np.where((q & g["mask"][m] & 0xf) != 0, f(ms * wgt), f(mms * wgt)).astype(np.float32)
        if len(preds) == 0:
            loc = np.full(K, 1.0, np.float32)
            gm = loc.copy()
        else:
            dv = mt = None
            for i, p in enumerate(preds):
                d = c - c_of[p]
                idx = kk + d
                ok = (idx >= 0) & (idx < K)
                sv = np.where(ok, V[p][np.clip(idx, 0, K - 1)], DEAD)
                sg = np.where(ok, G[p][np.clip(idx, 0, K - 1)], DEAD)
                idl = idx - 1
                okl = (idl >= 0) & (idl < K)
                svl = np.where(okl, V[p][np.clip(idl, 0, K - 1)], DEAD)
                v = (sv + gp).astype(np.float32)
                gg = (sg + gpe).astype(np.float32)
                cand = np.where(v < gg, v, gg)
                gm = cand
                mv = (svl + csel).astype(np.float32)
                mv = np.where(cols == 0, np.float32(np.inf), mv)
                if i == 0:
                    dv = np.where(cols == 0, np.where(cand < 1.0, cand, f(1.0)), cand)
                    mt = mv
                else:
                    dv = np.where(cand < dv, cand, dv)
                    mt = np.where(mv < mt, mv, mt)
            loc = np.where(mt < dv, mt, dv).astype(np.float32)
        fv = loc.copy()
        e_prev = (cols[0] == 0) and fv[0] == 1.0
        for k in range(1, K):
            gsx = f(fv[k - 1] + (gpe if e_prev else gp))
            ins = gsx <= loc[k]
            if ins:
                fv[k] = gsx
            e_prev = bool(ins)
        fv = np.where(inside, fv, DEAD)
        V[m], G[m], c_of[m] = fv, np.where(inside, gm, DEAD), c
        # (the row's "minimum": by value plus a credit per query base not yet consumed -- cells of one row, and rows
        # of one set of predecessors, have consumed different numbers of bases; a_credit = 0: the plain minimum)
        fcred = fv + np.float32(a_credit) * (c + kk).astype(np.float32)
        a = int(np.argmin(fcred))
        amin[m], vmin[m], vraw[m] = c + a, fcred[a], fv[a]
        c_prev = c
        if c <= L - 1 < c + K:
            best = min(best, fv[L - 1 - c])
        if snk[m]:
            best = min(best, fv.min())
    return best, c_of


def chain(g, fam_member, qmask, ms=-2.0, mms=1.0, gp=5.0, gpe=2.0):
    """Full (unbanded) pairwise recurrence of the query against the chain of one family member's nodes."""
    f = np.float32
    ab = fam_member.packed()
    key = {(int(p), int(m)): r for r, (p, m) in enumerate(zip(g["pos"], g["mask"]))}
    nodes = [key[(int(c), int(m) & 31)] for c, m in zip(ab & 0xFFFFFF, (ab >> 24) & 0xff)]
    L = len(qmask)
    snk = set(g["snk"].tolist())
    V = G = None
    best = np.inf
    for r, m in enumerate(nodes):
        w = g["weight"][m]
        csel = np.where((qmask & g["mask"][m] & 0xf) != 0, f(ms * w), f(mms * w)).astype(np.float32)
        if r == 0:
            src = g["pred_off"][m] == g["pred_off"][m + 1]
            loc = np.ones(L, np.float32) if src else np.full(L, DEAD, np.float32)
            loc[0] = 1.0
            gm = loc.copy()
        else:
            v, gg = V + f(gp), G + f(gpe)
            gm = np.where(v < gg, v, gg)
            mv = np.concatenate([[np.inf], V[:-1] + csel[1:]]).astype(np.float32)
            dv = gm.copy()
            dv[0] = min(gm[0], 1.0)
            loc = np.where(mv < dv, mv, dv)
        fv = loc.copy()
        e_prev = fv[0] == 1.0
        for k in range(1, L):
            gsx = f(fv[k - 1] + (gpe if e_prev else gp))
            ins = gsx <= loc[k]
            if ins:
                fv[k] = gsx
            e_prev = bool(ins)
        V, G = fv, gm
        best = min(best, fv[L - 1])
        if m in snk:
            best = min(best, fv.min())
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chain", action="store_true", help="the chain of family members 0, 1, 2 instead of the DAG band")
    ap.add_argument("--refs", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=6)
    ap.add_argument("--length", type=int, default=1500)
    ap.add_argument("--width", type=int, default=50000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--sub", type=float, default=0.03)
    ap.add_argument("--dele", type=float, default=0.005)
    ap.add_argument("--ins", type=float, default=0.003)
    a = ap.parse_args()
    refs = synth.make_refs(a.refs, length=a.length, width=a.width, seed=a.seed)
    qs = synth.make_queries(refs, a.queries, seed=a.seed + 100, sub=a.sub, dele=a.dele, ins=a.ins)
    cs = util.cseqs_from_refs(refs)
    idx = po.Index(cs)
    for qi in range(qs.n):
        q = util.query_cseq(qs, qi)
        ids, sc, _ = idx.famfinder(q, po.ff_opts())
        fam = [cs[i] for i in ids]
        g = po.mseq_build(fam, 1.0)
        cells = po.mesh_compute(fam, q)
        val = cells["value"]
        N, L = val.shape
        vstar = min(val[:, L - 1].min(), val[g["snk"]].min())
        qmask = qs.seq(qi)
        # the bound T(m, s) as the kernel has it, in floats (units ignored: a study)
        pos, w = g["pos"].astype(np.int64), g["weight"].astype(np.float64)
        cols, inv = np.unique(pos, return_inverse=True)
        cmax = np.zeros(len(cols))
        np.maximum.at(cmax, inv, 2.0 * w)
        R = np.concatenate([np.cumsum(cmax[::-1])[::-1][1:], [0.0]])[inv]
        C = (len(cols) - 1 - inv).astype(np.float64)
        gmin, amax = cmax.min(), cmax.max()
        r = (L - 1 - np.arange(L)).astype(np.float64)
        Tm = np.minimum(amax * r[None, :], R[:, None] - gmin * np.maximum(0.0, C[:, None] - r[None, :]))
        line = "query %d: N %d L %d  V* %.2f |" % (qi, N, L, vstar)
        if a.chain:
            print(line + "  chain of member 0 / 1 / 2: U - V* = " + " / ".join("%.1f" % (chain(g, fam[j], qmask) - vstar) for j in (0, 1, 2)), flush=True)
            continue
        for K in (8, 16, 32):
            u, c_of = scout(g, qmask, K)
            alive = val <= u + Tm
            any_row = alive.any(axis=1)
            lo = np.where(any_row, alive.argmax(axis=1), 0)
            hi = np.where(any_row, L - 1 - alive[:, ::-1].argmax(axis=1), 0)
            wd = (hi - lo + 1)[any_row]
            line += "  K=%d: U-V* %.2f, run med %d max %d, cells %.3f |" % (K, u - vstar, np.median(wd), wd.max(), alive.mean())
        print(line, flush=True)


if __name__ == "__main__":
    main()
