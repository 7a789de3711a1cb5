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
    0xFFFFFFFF nowhere (no successors), an LDS slot number (first slot whose occupant has seen its
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
        if last[m] == 0:
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
