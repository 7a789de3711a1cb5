"""Shared helpers for the parity tests (test infrastructure)."""
import ctypes as C

import numpy as np

from oracle import pyoracle as po
from sina_amd import synth


def cseqs_from_refs(refs, idxs=None):
    idxs = range(refs.n) if idxs is None else idxs
    return [po.Cseq.from_packed("ref%d" % i, refs.seq(i), refs.width) for i in idxs]


def query_cseq(qs, i, upper=True):
    m = qs.seq(i)
    if upper:
        m = m & 0x0f
    ab = np.arange(len(m), dtype=np.uint32) | (m.astype(np.uint32) << 24)
    return po.Cseq.from_packed("q%d" % i, ab, len(m))


def graph_dict(fam, weight=1.0):
    """Oracle-built family DAG in the layout sina_hip_graph_batch wants."""
    g = po.mseq_build(fam, weight)
    n = g["n"]
    succ_min = np.full(n, 1000000, np.uint32)
    for m in range(n):
        s = g["succ"][g["succ_off"][m]:g["succ_off"][m + 1]]
        if len(s):
            succ_min[m] = g["pos"][s].min()
    g["succ_minpos"] = succ_min
    return g


def finish_alignment(q_masks, out, pos, width, lowercase_unaligned=False, insertion_remove=False):
    """Applies the cseq container steps of backtrack() (mesh.h:603-726) to the device's
    per-append columns, using the oracle's cseq ops.  Returns (aligned string, log)."""
    L = po.lib()
    c = po.Cseq("out")
    n = int(out["n_out"])
    qlen = len(q_masks)
    tail, head = int(out["cutoff_tail"]), int(out["cutoff_head"])
    n_al = int(out["aligned_bases"])
    # append order: tail overhang (unless overhang=remove), aligned bases from end_s
    # downwards, head overhang
    if n == n_al:
        kept_tail = kept_head = 0
    else:
        kept_tail, kept_head = tail, head
    assert n == n_al + kept_tail + kept_head, (n, n_al, tail, head)
    idx = []
    if kept_tail:
        idx += [qlen - 1 - i for i in range(tail)]
    s_end = int(out["end_s"])
    idx += [s_end - i for i in range(n_al)]
    if kept_head:
        idx += [head - 1 - i for i in range(head)]
    assert len(idx) == n
    for i, qi in enumerate(idx):
        m = int(q_masks[qi])
        unaligned = (i < kept_tail) or (i >= kept_tail + n_al)
        if lowercase_unaligned and unaligned:
            m |= 16
        L.so_cseq_append_base(c.h, int(pos[i]) & 0xFFFFFF | (m << 24), None)
    lg = po.new_log()
    assert L.so_cseq_set_width(c.h, width) == 0
    L.so_cseq_reverse(c.h)
    rc = L.so_cseq_fix_duplicate_positions(c.h, C.byref(lg), int(lowercase_unaligned), int(insertion_remove))
    txt = po.log_text(lg)
    L.so_log_free(C.byref(lg))
    return (c.aligned() if rc == 0 else None), txt


def f32_bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def row_store_model(pred_off, pred, n_slots, far_lds=192):
    """Where mesh_dp_kernel keeps each finished DP row for its successors (sina_amd/csrc/common.h):
    0xFFFFFFFF nowhere (no successors, or only the next row: the kernel hands that over in registers), an LDS slot number (first slot whose occupant has seen its
    last successor), or 0x80000000 | spill row index when no slot is free, some successor is more
    than far_lds rows away, or the last successor lies in a later allocation segment (the rows are
    allocated in independent segments of max(256, ceil(n / 16)) rows, every segment starting with all
    slots free: graph_build.hip step 7)."""
    n = len(pred_off) - 1
    last = np.zeros(n, np.int64)
    node = np.repeat(np.arange(n), np.diff(pred_off))
    np.maximum.at(last, np.asarray(pred, np.int64), node)
    out = np.full(n, 0xFFFFFFFF, np.uint32)
    seg_len = max(256, (n + 15) // 16)
    free_at = [0] * n_slots
    nsp = 0
    for m in range(n):
        if m % seg_len == 0:
            free_at = [0] * n_slots
        if last[m] == 0 or last[m] == m + 1:   # no successor / only the next row: handed over in registers
            continue
        slot = -1
        seg_end = min(n, (m // seg_len + 1) * seg_len)
        if last[m] - m <= far_lds and last[m] < seg_end:
            for x in range(n_slots):
                if free_at[x] <= m:
                    slot = x
                    break
        if slot >= 0:
            free_at[slot] = last[m]
            out[m] = slot
        else:
            out[m] = 0x80000000 | nsp
            nsp += 1
    return out


# ---------------------------------------------------------------- reference-parts mesh cases
# (tests/golden/make_ref_vectors.py computes the planes of these cases with the reference's own
# scoring schemes and DAG and stores one hash per plane; the oracle and the GPU are checked against
# the hashes.  A case: synth seed, references generated, their length / width, family size, scheme.)
MESH_PLANES = ("value", "gapm_val", "gaps_val", "value_midx", "value_sidx", "gapm_idx", "gaps_idx")


def _mesh_cases():
    schemes = [
        dict(match=2.0, mismatch=-1.0, gap=5.0, gapext=2.0, weighted=False, forbid=False),
        dict(match=3.0, mismatch=-2.0, gap=4.0, gapext=1.5, weighted=False, forbid=False),
        dict(match=2.0, mismatch=-1.0, gap=5.0, gapext=2.0, weighted=True, forbid=False),
        dict(match=2.0, mismatch=-1.0, gap=5.0, gapext=2.0, weighted=False, forbid=True),
        dict(match=2.0, mismatch=-1.0, gap=5.0, gapext=2.0, weighted=True, forbid=True),
        dict(match=1.0, mismatch=-1.0, gap=2.0, gapext=2.0, weighted=False, forbid=False),
    ]
    cases = []
    fam_sizes = [1, 2, 3, 5, 7, 12, 15, 23, 40, 41, 40, 40]
    for i in range(24):
        sch = dict(schemes[i % len(schemes)])
        sch["fs_weight"] = (1.0, 0.0, 2.5)[i % 3]
        cases.append(dict(seed=300 + i, n_refs=96, length=260 + 20 * (i % 5), width=1700 + 100 * (i % 4),
                          F=fam_sizes[i % len(fam_sizes)], scheme=sch))
    # one full-length 16S family of 40 (2800 rows x 1500 columns), default scheme
    cases.append(dict(seed=777, n_refs=48, length=1500, width=50000, F=40,
                      scheme=dict(schemes[0], fs_weight=1.0)))
    return cases


MESH_CASES = _mesh_cases()


def mesh_case_inputs(case):
    """(family as packed aligned sequences, packed upper-cased query, width, column weights or None, scheme)."""
    small = case["length"] < 1000
    refs = synth.make_refs(case["n_refs"], length=case["length"], width=case["width"], seed=case["seed"], n_clades=3,
                           amb_rate=0.03 if small else 0.0, lower_rate=0.05 if small else 0.0,
                           long_del_prob=0.3 if small else 0.1, del_rate=0.03 if small else 0.01,
                           ins_rate=0.02 if small else 0.005)
    qs = synth.make_queries(refs, 16, seed=case["seed"] + 5000, amb_rate=0.02 if small else 0.0)
    rng = np.random.default_rng(case["seed"] + 9000)
    sizes = np.diff(refs.off)
    usable = np.flatnonzero(sizes >= case["length"] // 3)   # (a long deletion can take most of a short reference)
    ids = rng.choice(usable, size=case["F"], replace=False)
    fam = [refs.seq(int(i)).copy() for i in ids]
    qi = next(i for i in range(qs.n) if len(qs.seq(i)) >= case["length"] // 3)
    qm = qs.seq(qi) & 0x0f   # the aligner upper-cases its working copy (align.cpp:324-326)
    qa = np.arange(len(qm), dtype=np.uint32) | (qm.astype(np.uint32) << 24)
    w = None
    if case["scheme"]["weighted"]:   # long enough that pos + 1 + offset never runs past the end
        w = rng.uniform(0.2, 2.0, size=refs.width + len(qm) + 8).astype(np.float32)
    return fam, qa, refs.width, w, case["scheme"]


def plane_hash(a):
    """64-bit hash of a plane's 32-bit patterns (floats by their bits)."""
    import hashlib
    b = np.ascontiguousarray(a)
    b = b.view(np.uint32) if b.dtype == np.float32 else b.astype(np.uint32)
    return np.frombuffer(hashlib.sha1(b.tobytes()).digest()[:8], np.uint64)[0]


def set_knobs(monkeypatch, **kw):
    """Test hooks of the library, all in ONE environment variable (csrc/common.h test_knob):
    SINA_HIP_TEST="geom=T,B;generic=1;dense_div=N;lds_kb=N;rho=X".  A value of None removes the key; the other keys
    of the variable stay as they are."""
    import os
    cur = dict(x.split("=", 1) for x in os.environ.get("SINA_HIP_TEST", "").split(";") if "=" in x)
    for k, v in kw.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    if cur:
        monkeypatch.setenv("SINA_HIP_TEST", ";".join("%s=%s" % kv for kv in cur.items()))
    else:
        monkeypatch.delenv("SINA_HIP_TEST", raising=False)
