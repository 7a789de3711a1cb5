"""CPU study for the certified row skip of the DP kernels.

For a handful of bench-shaped queries: the oracle's full value plane, the bound
T(m, s) = U + |match| * min(wmax * (L-1-s), R(m)) for several choices of U, and what survives:
cells with value <= T, and (row, 512-column strip) pairs holding at least one such cell -- the
unit the strip kernel can skip.  Test infrastructure: uses the oracle.

  python tools/prune_study.py [--refs 20000] [--queries 6] [--length 1500] [--width 50000]
"""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from sina_amd import synth  # noqa: E402
from tests import util  # noqa: E402


def col_suffix_gain(g):
    """R(m) = sum over occupied columns right of pos(m) of the column's largest node weight (double)."""
    pos, w = g["pos"].astype(np.int64), g["weight"].astype(np.float64)
    cols, inv = np.unique(pos, return_inverse=True)
    cmax = np.zeros(len(cols))
    np.maximum.at(cmax, inv, w)
    suffix = np.concatenate([np.cumsum(cmax[::-1])[::-1][1:], [0.0]])  # strictly right of the column
    return suffix[inv], cmax, inv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--refs", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=6)
    ap.add_argument("--length", type=int, default=1500)
    ap.add_argument("--width", type=int, default=50000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--strip", type=int, default=512)
    ap.add_argument("--window", type=int, default=0)
    a = ap.parse_args()
    refs = synth.make_refs(a.refs, length=a.length, width=a.width, seed=a.seed)
    qs = synth.make_queries(refs, a.queries, seed=a.seed + 100,
                            window=(1.0 / 3, a.window) if a.window else None)
    cs = util.cseqs_from_refs(refs)
    idx = po.Index(cs)
    match = 2.0
    for qi in range(qs.n):
        q = util.query_cseq(qs, qi)
        ids, sc, _ = idx.famfinder(q, po.ff_opts())
        fam = [cs[i] for i in ids]
        g = po.mseq_build(fam, 1.0)
        cells = po.mesh_compute(fam, q)
        val = cells["value"].astype(np.float64)
        N, L = val.shape
        R, cmax, inv = col_suffix_gain(g)
        wmax = float(g["weight"].max())
        # the optimum as backtrack() finds it (approximately: min over last column and sink rows)
        vstar = min(val[:, L - 1].min(), val[g["snk"]].min())
        rem_q = wmax * (L - 1 - np.arange(L))
        G = np.minimum(rem_q[None, :], R[:, None])
        gmax0 = match * min(wmax * (L - 1), R.max())
        print("query %d: N %d L %d family %d  V* %.1f  bound at start %.1f  ratio %.3f  wmax %.3f  cols %d"
              % (qi, N, L, len(ids), vstar, -gmax0, vstar / -gmax0, wmax, len(cmax)))
        for slack in (0.0, 50.0, 150.0, 300.0, 600.0):
            U = vstar + slack
            alive = val <= U + match * G
            cells_alive = alive.mean()
            ns = (L + a.strip - 1) // a.strip
            rows = 0
            per_strip = []
            for k in range(ns):
                r = alive[:, k * a.strip:(k + 1) * a.strip].any(axis=1)
                rows += int(r.sum())
                per_strip.append(float(r.mean()))
            print("   U = V* + %5.0f: cells alive %.3f   wave-rows alive %.3f  per strip %s"
                  % (slack, cells_alive, rows / (N * ns), " ".join("%.2f" % x for x in per_strip)))


if __name__ == "__main__":
    main()
