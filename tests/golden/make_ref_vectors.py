"""Generates tests/golden/ref_vectors.npz from the REAL reference parts compiled in
oracle/_ref (kmer.h, idset.h, aligned_base.cpp, graph.h, scoring_schemes.h under
/root/reference/src).  Run in the build container only:

    make -C oracle && python tests/golden/make_ref_vectors.py

The .npz holds inputs and the reference's outputs; tests compare the oracle (and,
on the GPU, the HIP path) against them without needing /root/reference.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from sina_amd import synth  # noqa: E402

R = po.ref()
out = {}

# ---- k-mers: iterable semantics incl. dropped final k-mer, prefix and unique filters
refs = synth.make_refs(24, length=160, width=1600, seed=101, amb_rate=0.02, lower_rate=0.05, long_del_prob=0.3)
modes = [(10, 1, 0, 0), (10, 1, 0, 1), (10, 0, 0, 0), (10, 0, 0, 1), (6, 0, 0, 0), (4, 2, 5, 0), (8, 1, 0, 1)]
out["kmer_modes"] = np.array(modes, np.int64)
out["kmer_ab"] = refs.ab
out["kmer_off"] = refs.off
res, roff = [], [0]
for i in range(refs.n):
    ab = np.ascontiguousarray(refs.seq(i))
    for (k, pl, pv, u) in modes:
        buf = np.zeros(len(ab) + 1, np.uint32)
        n = R.ref_kmers(ab.ctypes.data_as(po.u32p), len(ab), k, pl, pv, u, buf.ctypes.data_as(po.u32p))
        res.append(buf[:n].copy())
        roff.append(roff[-1] + n)
out["kmer_out"] = np.concatenate(res)
out["kmer_out_off"] = np.array(roff, np.int64)

# ---- vlimap: push_back / invert / serialised bytes (idset.h:310-398)
rng = np.random.default_rng(7)
vl_sets, vl_bytes, vl_inv_bytes, vl_sizes = [], [], [], []
for size, fill in [(0, 0), (255, 10), (256, 50), (257, 100), (10000, 10), (10000, 50), (70000, 3)]:
    n = size * fill // 100
    data = np.sort(rng.choice(size, n, replace=False)).astype(np.uint32) if n else np.zeros(0, np.uint32)
    v = R.ref_vlimap_new(size)
    for x in data:
        R.ref_vlimap_push_back(v, int(x))
    buf = np.zeros(4 * size + 64, np.uint8)
    nb = R.ref_vlimap_write(v, buf.ctypes.data_as(po.u8p), len(buf))
    b1 = buf[:nb].copy()
    R.ref_vlimap_invert(v)
    nb = R.ref_vlimap_write(v, buf.ctypes.data_as(po.u8p), len(buf))
    b2 = buf[:nb].copy()
    R.ref_vlimap_free(v)
    vl_sets.append(data)
    vl_bytes.append(b1)
    vl_inv_bytes.append(b2)
    vl_sizes.append(size)
out["vl_sizes"] = np.array(vl_sizes, np.int64)
for i, (d, b1, b2) in enumerate(zip(vl_sets, vl_bytes, vl_inv_bytes)):
    out["vl_set_%d" % i] = d
    out["vl_ser_%d" % i] = b1
    out["vl_inv_%d" % i] = b2

# ---- family DAG on the real dag<T> + cell planes with the real scoring_scheme_simple
fam_refs = synth.make_refs(14, length=110, width=700, seed=202, n_clades=3, amb_rate=0.03, lower_rate=0.05,
                           long_del_prob=0.4, del_rate=0.03, ins_rate=0.02)
qs = synth.make_queries(fam_refs, 3, seed=203, amb_rate=0.02)
out["dag_ab"] = fam_refs.ab
out["dag_off"] = fam_refs.off
out["dag_width"] = np.int64(fam_refs.width)
arrs = [np.ascontiguousarray(fam_refs.seq(i)) for i in range(fam_refs.n)]
ptrs = (po.u32p * len(arrs))(*[a.ctypes.data_as(po.u32p) for a in arrs])
ns = np.array([len(a) for a in arrs], np.uint32)
for wi, fsw in enumerate([1.0, 0.0, 2.5]):
    rd = R.ref_dag_build(ptrs, ns.ctypes.data_as(po.u32p), len(arrs), fam_refs.width, fsw)
    N = R.ref_dag_size(rd)
    ids = np.zeros(N, np.uint32); pos = np.zeros(N, np.uint32); mask = np.zeros(N, np.uint8)
    w = np.zeros(N, np.float32); poff = np.zeros(N + 1, np.uint32); pred = np.zeros(16 * N + 16, np.uint32)
    src = np.zeros(N, np.uint32); snk = np.zeros(N, np.uint32)
    nsrc, nsnk = C.c_uint32(), C.c_uint32()
    e = R.ref_dag_dump(rd, ids.ctypes.data_as(po.u32p), pos.ctypes.data_as(po.u32p), mask.ctypes.data_as(po.u8p),
                       w.ctypes.data_as(po.f32p), poff.ctypes.data_as(po.u32p), pred.ctypes.data_as(po.u32p),
                       C.byref(nsrc), src.ctypes.data_as(po.u32p), C.byref(nsnk), snk.ctypes.data_as(po.u32p))
    p = "dag%d_" % wi
    out[p + "fs_weight"] = np.float32(fsw)
    out[p + "ids"] = ids; out[p + "pos"] = pos; out[p + "mask"] = mask; out[p + "weight"] = w
    out[p + "pred_off"] = poff; out[p + "pred"] = pred[:e].copy()
    out[p + "src"] = src[:nsrc.value].copy(); out[p + "snk"] = snk[:nsnk.value].copy()
    if wi == 0:
        for qi in range(qs.n):
            qm = qs.seq(qi) & 0x0f   # aligner upper-cases the working copy (align.cpp:324-326)
            qa = (np.arange(len(qm), dtype=np.uint32) | (qm.astype(np.uint32) << 24))
            cells = np.zeros((N, len(qa)), po.CELL_DTYPE)
            R.ref_mesh_compute_simple(rd, qa.ctypes.data_as(po.u32p), len(qa), -2.0, 1.0, 5.0, 2.0,
                                      cells.ctypes.data_as(C.c_void_p))
            out["mesh_q%d" % qi] = qa
            for f in ("value_midx", "value_sidx", "gapm_idx", "gaps_idx"):
                out["mesh%d_%s" % (qi, f)] = cells[f].astype(np.uint16)
            for f in ("value", "gapm_val", "gaps_val"):
                out["mesh%d_%s" % (qi, f)] = cells[f].view(np.uint32)
    R.ref_dag_free(rd)

# ---- cell planes of many families, all scheme / transition combinations, as plane hashes
# (the cell loop of oracle/ref_parts.cpp on the real dag<T> with the real scoring_scheme_simple /
# scoring_scheme_weighted; --insertion=forbid through the aspace-aware transition).  Inputs are
# regenerated by the tests from the same synth seeds: tests/util.py mesh_case_inputs().
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests import util  # noqa: E402
R.ref_mesh_compute.argtypes = [C.c_void_p, po.u32p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float,
                               po.f32p, C.c_uint32, C.c_int, C.c_void_p]
case_rows, case_hashes = [], []
for case in util.MESH_CASES:
    fam, qa, width, w, sch = util.mesh_case_inputs(case)
    arrs = [np.ascontiguousarray(a) for a in fam]
    ptrs = (po.u32p * len(arrs))(*[a.ctypes.data_as(po.u32p) for a in arrs])
    ns = np.array([len(a) for a in arrs], np.uint32)
    rd = R.ref_dag_build(ptrs, ns.ctypes.data_as(po.u32p), len(arrs), width, sch["fs_weight"])
    N = R.ref_dag_size(rd)
    cells = np.zeros((N, len(qa)), po.CELL_DTYPE)
    R.ref_mesh_compute(rd, qa.ctypes.data_as(po.u32p), len(qa), -sch["match"], -sch["mismatch"], sch["gap"],
                       sch["gapext"], w.ctypes.data_as(po.f32p) if w is not None else None,
                       len(w) if w is not None else 0, int(sch["forbid"]), cells.ctypes.data_as(C.c_void_p))
    R.ref_dag_free(rd)
    case_rows.append((N, len(qa)))
    case_hashes.append([util.plane_hash(cells[f]) for f in util.MESH_PLANES])
out["mesh_case_shape"] = np.array(case_rows, np.int64)
out["mesh_case_hash"] = np.array(case_hashes, np.uint64)

# ---- scoring scheme single-op probes (real scoring_schemes.h arithmetic)
rs = np.random.default_rng(9)
wts = rs.uniform(0.1, 2.0, 64).astype(np.float32)
probes, vals = [], []
chars = "AGCUNRYagcu"
for i in range(400):
    op = int(rs.integers(0, 5)); prev = np.float32(rs.uniform(-500, 500)); mpos = int(rs.integers(0, 50))
    mc = chars[int(rs.integers(0, len(chars)))]; sc = chars[int(rs.integers(0, 5))]
    mw = np.float32(1.0 / 2 + rs.integers(1, 41) / np.float32(40)); offs = int(rs.integers(0, 8))
    weighted = int(rs.integers(0, 2))
    v = R.ref_score_op(op, prev, mpos, ord(mc), mw, ord(sc), offs, -2.0, 1.0, 5.0, 2.0,
                       wts.ctypes.data_as(po.f32p) if weighted else None, len(wts))
    probes.append((op, np.float32(prev).view(np.uint32), mpos, ord(mc), np.float32(mw).view(np.uint32), ord(sc), offs, weighted))
    vals.append(np.float32(v).view(np.uint32))
out["score_weights"] = wts
out["score_probes"] = np.array(probes, np.int64)
out["score_vals"] = np.array(vals, np.uint32)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vectors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
