"""Deterministic synthetic rRNA-like workloads (SURVEY.md section 8d clade model).

A random ancestor of length LEN; `n_clades` clade consensus sequences at
`clade_div` divergence; each reference = its clade consensus with a per-reference
substitution rate U(sub_lo, sub_hi), `del_rate` single-base deletions,
`ins_rate` single-base insertions (placed in the column right after the base),
occasional long deletions (`long_del_prob` per reference, 50-250 bases) that
create long-range DAG edges.  Ancestor position i lives in alignment column
i * (WIDTH // LEN).  Queries = the bases of a random reference with 3 %
substitutions, 0.5 % deletions, 0.3 % insertions (so the substring shortcut of
align.cpp:336-388 never fires), optionally cut to a window (V4 amplicons).

Sequences are handed around in the packed form the C-ABI uses:
  aligned base  = uint32  (column & 0xFFFFFF) | iupac_mask << 24
  iupac mask    = A 1, G 2, C 4, T/U 8, lower-case bit 16
"""
from dataclasses import dataclass

import numpy as np

CODE_TO_MASK = np.array([1, 2, 4, 8], dtype=np.uint8)  # BASE_A, BASE_G, BASE_C, BASE_TU


@dataclass
class RefSet:
    ab: np.ndarray      # uint32, concatenated packed aligned bases
    off: np.ndarray     # int64 [n+1]
    width: int

    @property
    def n(self):
        return len(self.off) - 1

    def seq(self, i):
        return self.ab[self.off[i]:self.off[i + 1]]


@dataclass
class QuerySet:
    mask: np.ndarray    # uint8, concatenated iupac masks
    off: np.ndarray     # int64 [n+1]
    src: np.ndarray     # int64 [n] reference each query was derived from

    @property
    def n(self):
        return len(self.off) - 1

    def seq(self, i):
        return self.mask[self.off[i]:self.off[i + 1]]

    def packed(self, i):
        m = self.seq(i)
        return (np.arange(len(m), dtype=np.uint32) | (m.astype(np.uint32) << 24))


def make_refs(n_refs, length=1500, width=50000, seed=1, n_clades=20, clade_div=0.12, sub_lo=0.01,
              sub_hi=0.08, del_rate=0.01, ins_rate=0.005, long_del_prob=0.1, amb_rate=0.0,
              lower_rate=0.0, chunk=8192):
    rng = np.random.default_rng(seed)
    stride = max(width // length, 2)
    anc = rng.integers(0, 4, size=length, dtype=np.uint8)
    clades = np.tile(anc, (n_clades, 1))
    mut = rng.random((n_clades, length)) < clade_div
    clades[mut] = (clades[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3

    parts, counts = [], []
    for lo in range(0, n_refs, chunk):
        n = min(chunk, n_refs - lo)
        cl = rng.integers(0, n_clades, size=n)
        base = clades[cl].copy()                                    # [n, LEN]
        rate = rng.uniform(sub_lo, sub_hi, size=(n, 1))
        sub = rng.random((n, length)) < rate
        base[sub] = (base[sub] + rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)) & 3
        keep = rng.random((n, length)) >= del_rate
        # long deletions
        has_ld = rng.random(n) < long_del_prob
        ld_len = rng.integers(50, 251, size=n)
        ld_start = rng.integers(0, max(length - 250, 1), size=n)
        idx = np.arange(length)[None, :]
        ld = has_ld[:, None] & (idx >= ld_start[:, None]) & (idx < (ld_start + ld_len)[:, None])
        keep &= ~ld
        ins = (rng.random((n, length)) < ins_rate) & keep
        insb = rng.integers(0, 4, size=(n, length), dtype=np.uint8)

        m_main = CODE_TO_MASK[base]
        m_ins = CODE_TO_MASK[insb]
        if amb_rate > 0:
            amb = rng.random((n, length)) < amb_rate
            m_main = np.where(amb, rng.integers(1, 16, size=(n, length), dtype=np.uint8), m_main)
        if lower_rate > 0:
            low = rng.random((n, length)) < lower_rate
            m_main = np.where(low, m_main | 16, m_main).astype(np.uint8)
        col = (np.arange(length, dtype=np.uint32) * stride)[None, :]
        packed = np.empty((n, length, 2), dtype=np.uint32)
        packed[:, :, 0] = col | (m_main.astype(np.uint32) << 24)
        packed[:, :, 1] = (col + 1) | (m_ins.astype(np.uint32) << 24)
        valid = np.stack([keep, ins], axis=2)
        parts.append(packed[valid])
        counts.append(valid.reshape(n, -1).sum(axis=1))
    ab = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
    cnt = np.concatenate(counts) if counts else np.zeros(0, np.int64)
    off = np.zeros(n_refs + 1, dtype=np.int64)
    np.cumsum(cnt, out=off[1:])
    return RefSet(ab=ab, off=off, width=int(width))


def make_queries(refs, n_queries, seed=2, sub=0.03, dele=0.005, ins=0.003, window=None,
                 amb_rate=0.0, lower_rate=0.0):
    """window=(start_frac, length) cuts each derived query to a sub-window.  sub / dele / ins may be sequences:
    query i then gets rate[i % len(rate)] (a launch that mixes near-identical and distant queries)."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, refs.n, size=n_queries)
    lens = (refs.off[src + 1] - refs.off[src]).astype(np.int64)
    qoff = np.zeros(n_queries + 1, dtype=np.int64)
    np.cumsum(lens, out=qoff[1:])
    total = int(qoff[-1])
    # flat gather of the chosen references' base masks
    flat_idx = np.arange(total, dtype=np.int64) - np.repeat(qoff[:-1], lens) + np.repeat(refs.off[src], lens)
    m = (refs.ab[flat_idx] >> 24).astype(np.uint8)
    owner = np.repeat(np.arange(n_queries, dtype=np.int64), lens)
    code = np.zeros(total, dtype=np.uint8)           # base code of unambiguous bases
    lowbit = m & 16
    mm = m & 15
    for c in range(4):
        code[mm == (1 << c)] = c
    def per_base(rate):  # a rate per query, spread over the query's bases
        r = np.atleast_1d(np.asarray(rate, dtype=np.float64))
        return r[0] if len(r) == 1 else np.repeat(r[np.arange(n_queries) % len(r)], lens)
    sub, dele, ins = per_base(sub), per_base(dele), per_base(ins)
    s = rng.random(total) < sub
    code_s = (code + rng.integers(1, 4, size=total, dtype=np.uint8)) & 3
    newm = np.where(s, CODE_TO_MASK[code_s] | lowbit, m).astype(np.uint8)
    keep = rng.random(total) >= dele
    addins = (rng.random(total) < ins) & keep
    reps = keep.astype(np.int64) + addins.astype(np.int64)
    out_m = np.repeat(newm, reps)
    out_owner = np.repeat(owner, reps)
    # second copy of a repeated element is the inserted base
    first = np.ones(len(out_m), dtype=bool)
    pos = np.cumsum(reps) - reps
    ins_pos = pos[addins] + 1
    first[ins_pos] = False
    out_m[~first] = CODE_TO_MASK[rng.integers(0, 4, size=int((~first).sum()))]
    if amb_rate > 0:
        amb = rng.random(len(out_m)) < amb_rate
        out_m = np.where(amb, rng.integers(1, 16, size=len(out_m), dtype=np.uint8), out_m).astype(np.uint8)
    if lower_rate > 0:
        low = rng.random(len(out_m)) < lower_rate
        out_m = np.where(low, out_m | 16, out_m).astype(np.uint8)
    cnt = np.bincount(out_owner, minlength=n_queries).astype(np.int64)
    off = np.zeros(n_queries + 1, dtype=np.int64)
    np.cumsum(cnt, out=off[1:])
    if window is not None:
        frac, wlen = window
        st = off[:-1] + (cnt * frac).astype(np.int64)
        en = np.minimum(st + wlen, off[1:])
        wl = en - st
        woff = np.zeros(n_queries + 1, dtype=np.int64)
        np.cumsum(wl, out=woff[1:])
        gi = np.arange(int(woff[-1]), dtype=np.int64) - np.repeat(woff[:-1], wl) + np.repeat(st, wl)
        out_m = out_m[gi]
        off = woff
    return QuerySet(mask=np.ascontiguousarray(out_m), off=off, src=src)


def pick_queries(qs, pick):
    """The queries qs[i] for i in pick, in that order (repeats allowed), as a QuerySet of their own."""
    pick = np.asarray(pick, dtype=np.int64)
    lens = (qs.off[pick + 1] - qs.off[pick]).astype(np.int64)
    off = np.zeros(len(pick) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    gi = np.arange(int(off[-1]), dtype=np.int64) - np.repeat(off[:-1], lens) + np.repeat(qs.off[pick], lens)
    return QuerySet(mask=np.ascontiguousarray(qs.mask[gi]), off=off, src=qs.src[pick])


def with_repeats(qs, dup_rate, block, seed=9):
    """Amplicon-like repetition: inside every block of `block` consecutive queries, a fraction dup_rate of the
    queries is replaced by copies of other queries of the same block (chosen among those that stay)."""
    if dup_rate <= 0:
        return qs
    rng = np.random.default_rng(seed)
    pick = np.arange(qs.n, dtype=np.int64)
    for b0 in range(0, qs.n, block):
        b1 = min(qs.n, b0 + block)
        n = b1 - b0
        n_dup = min(n - 1, int(round(dup_rate * n)))
        if n_dup <= 0:
            continue
        order = rng.permutation(n)
        dups, keep = order[:n_dup], order[n_dup:]
        pick[b0 + dups] = b0 + rng.choice(keep, size=n_dup, replace=True)
    return pick_queries(qs, pick)


def aligned_string(ab, width, dna=False):
    """Render a packed aligned sequence as a gapped string ('-' for gaps)."""
    chars = b".AGRCMSVUWKDYHBN.agrcmsvuwkdyhbn" if not dna else b".AGRCMSVTWKDYHBN.agrcmsvtwkdyhbn"
    out = np.full(width, ord("-"), dtype=np.uint8)
    pos = (ab & 0xFFFFFF).astype(np.int64)
    out[pos] = np.frombuffer(chars, dtype=np.uint8)[(ab >> 24) & 31]
    return out.tobytes().decode()


def bases_string(mask):
    chars = np.frombuffer(b".AGRCMSVUWKDYHBN.agrcmsvuwkdyhbn", dtype=np.uint8)
    return chars[np.asarray(mask) & 31].tobytes().decode()
