"""GPU parity at the sizes the benchmark configurations run at: the k-mer search above one
reference tile (32 768 references) with the dense-bitmap lists active, the V4 and 23S query
shapes against >= 32 768 references, the rank != 0 start-up path (a store that arrives by
broadcast), --turn, and a 500 000-reference property test (BASELINE.json configs[3])."""
import ctypes as C
import os

import numpy as np
import pytest

from sina_amd import capi, pipeline, synth
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wide(oracle):
    """70 000 short references in six clades: three reference tiles, 200+ posting lists longer than
    1/64 of the references (the dense-bitmap path), an oracle index that builds in seconds."""
    refs = synth.make_refs(70000, length=200, width=2000, seed=5, n_clades=6)
    cs = util.cseqs_from_refs(refs)
    qs = synth.make_queries(refs, 12, seed=6, amb_rate=0.01)
    return refs, cs, qs


def _scores_and_topk_equal(ctx, idx, qs, maxes=(1, 41, 410, 4096)):
    for qi in range(qs.n):
        q = util.query_cseq(qs, qi)
        assert (ctx.kmer_scores(qs.seq(qi)) == idx.scores(q)).all(), qi
    for mx in maxes:
        gi, gs, gn = ctx.kmer_topk(qs.mask, qs.off, mx)
        for qi in range(qs.n):
            oi, os_ = idx.find(util.query_cseq(qs, qi), mx)
            assert gn[qi] == len(oi)
            assert (gi[qi, :gn[qi]] == oi).all(), (mx, qi)
            assert (gs[qi, :gn[qi]] == os_).all(), (mx, qi)


@pytest.mark.parametrize("dense_div,expect_dense", [(None, "some"), ("1", "none"), ("1000000", "many")])
def test_kmer_multi_tile_dense_equals_oracle(oracle, wide, monkeypatch, dense_div, expect_dense):
    """kmer_count_kernel + kmer_select_kernel on the code path every BASELINE config runs: several
    reference tiles and posting lists counted from bitmaps (kmer_search.cpp:366-420, idset.h:315-337).
    SINA_HIP_TEST="dense_div=N" moves the list-length threshold (n_refs / div, at least 256): 1 switches the
    bitmaps off, a huge value makes every list above 256 references a bitmap -- the results must not
    depend on it."""
    refs, cs, qs = wide
    idx = oracle.Index(cs, k=10)
    off, ids = idx.csr()
    util.set_knobs(monkeypatch, dense_div=dense_div)
    ctx = capi.Context(0)
    try:
        ctx.upload_refs(refs.ab, refs.off, refs.width)
        ctx.upload_index(10, False, off, ids)          # (bitmaps are rebuilt by the first search after this)
        _scores_and_topk_equal(ctx, idx, qs)
        nd = ctx.stats()["n_dense_lists"]
        ln = np.diff(off.astype(np.int64))
        if expect_dense == "none":
            assert nd == 0
        elif expect_dense == "some":
            assert nd == int((ln > max(256, refs.n // 32)).sum()) and nd > 100
        else:
            assert nd == int((ln > 256).sum()) and nd > 500
    finally:
        ctx.close()


@pytest.mark.parametrize("regime", ["spread", "identical", "two_groups", "unrelated"])
def test_kmer_select_regimes_equal_oracle(oracle, regime):
    """kmer_select_kernel's ways to the cut score on rows long enough for its histogram short cut
    (>= 16384 references), against the oracle's partial_sort on (score, id) (kmer_search.cpp:405-418):
    spread-out scores (sampled threshold, every score at or above the cut sorted), one giant group of
    equal scores (the short cut refuses, 8-way search, ties with the largest ids), two groups, and
    queries that share next to nothing with the references (fewer positive scores than asked for)."""
    kw = dict(length=400, width=3000, long_del_prob=0.0)
    if regime == "spread":
        refs = synth.make_refs(24000, seed=31, n_clades=50, sub_lo=0.01, sub_hi=0.2, **kw)
    elif regime == "identical":
        refs = synth.make_refs(20000, seed=32, n_clades=1, sub_lo=0.0, sub_hi=0.0, del_rate=0.0, ins_rate=0.0, **kw)
    elif regime == "two_groups":
        refs = synth.make_refs(20000, seed=33, n_clades=2, sub_lo=0.0, sub_hi=0.0, del_rate=0.0, ins_rate=0.0, **kw)
    else:
        refs = synth.make_refs(20000, seed=34, n_clades=8, **kw)
    src = synth.make_refs(64, seed=35, n_clades=4, **kw) if regime == "unrelated" else refs
    qs = synth.make_queries(src, 6, seed=36)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    ctx = capi.Context(0)
    try:
        ctx.upload_refs(refs.ab, refs.off, refs.width)
        ctx.build_index(10, False)
        _scores_and_topk_equal(ctx, idx, qs, maxes=(1, 7, 41, 410, 4096))
    finally:
        ctx.close()


@pytest.mark.parametrize("regime", ["spread", "identical", "two_groups", "unrelated"])
def test_kmer_candidate_lists_equal_oracle(oracle, monkeypatch, regime):
    """The search's candidate-list path (kmer_count_kernel<true>: at 65 536 references and more, for top-M up to 128
    the score rows never leave the chip -- every tile hands the references that reach tile 0's M-th largest
    per-thread maximum to a list, a sort of the list gives the top M) against the oracle's partial_sort on (score,
    id): spread-out scores, ONE giant group of equal scores and queries that share next to nothing with the
    references (the list overflows: the launch is repeated with the score rows), two groups; and the same searches
    with the rows forced (SINA_HIP_TEST=kmer_rows=1) give the same bytes."""
    kw = dict(length=200, width=1600, long_del_prob=0.0)
    n = 70000
    if regime == "spread":
        refs = synth.make_refs(n, seed=131, n_clades=50, sub_lo=0.01, sub_hi=0.2, **kw)
    elif regime == "identical":
        refs = synth.make_refs(n, seed=132, n_clades=1, sub_lo=0.0, sub_hi=0.0, del_rate=0.0, ins_rate=0.0, **kw)
    elif regime == "two_groups":
        refs = synth.make_refs(n, seed=133, n_clades=2, sub_lo=0.0, sub_hi=0.0, del_rate=0.0, ins_rate=0.0, **kw)
    else:
        refs = synth.make_refs(n, seed=134, n_clades=8, **kw)
    src = synth.make_refs(64, seed=135, n_clades=4, **kw) if regime == "unrelated" else refs
    qs = synth.make_queries(src, 6, seed=136)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    ctx = capi.Context(0)
    try:
        ctx.upload_refs(refs.ab, refs.off, refs.width)
        ctx.build_index(10, False)
        _scores_and_topk_equal(ctx, idx, qs, maxes=(1, 7, 41, 128, 129, 410))
        fused = [ctx.kmer_topk(qs.mask, qs.off, m) for m in (1, 41, 128)]
        util.set_knobs(monkeypatch, kmer_rows=1)
        rows = [ctx.kmer_topk(qs.mask, qs.off, m) for m in (1, 41, 128)]
        for f, r in zip(fused, rows):
            assert all((x == y).all() for x, y in zip(f, r))
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SINA_FUZZ_SEEDS", "8"))))
def test_kmer_search_fuzz(oracle, monkeypatch, seed):
    """Seeded random k-mer searches: reference counts on both sides of one tile (32 768) and of the
    select kernel's short cut (16 384), k, fast / no-fast, clade structure, the dense-list threshold,
    index built on the device or uploaded, requested counts: full score vectors and top-M against the
    oracle."""
    rng = np.random.default_rng(7000 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]  # noqa: E731
    n_refs = int(pick([300, 5000, 17000, 33000, 40000]))
    length = int(pick([150, 300]))
    refs = synth.make_refs(n_refs, length=length, width=8 * length, seed=7100 + seed, n_clades=int(pick([1, 4, 30])),
                           clade_div=float(pick([0.05, 0.2])), sub_lo=0.0, sub_hi=float(pick([0.0, 0.05, 0.2])),
                           long_del_prob=0.0, amb_rate=float(pick([0.0, 0.02])))
    k, nofast = int(pick([6, 8, 10, 10])), bool(rng.integers(0, 2))
    dd = pick([None, None, "1", "8", "1000000"])
    if dd:
        util.set_knobs(monkeypatch, dense_div=dd)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=k, nofast=nofast)
    qs = synth.make_queries(refs, 5, seed=7200 + seed, sub=float(pick([0.0, 0.03, 0.2])), amb_rate=float(pick([0.0, 0.02])))
    ctx = capi.Context(0)
    try:
        ctx.upload_refs(refs.ab, refs.off, refs.width)
        if rng.integers(0, 2):
            ctx.build_index(k, nofast)
        else:
            off, ids = idx.csr()
            ctx.upload_index(k, nofast, off, ids)
        maxes = tuple(sorted(set(int(x) for x in rng.choice([1, 2, 40, 41, 400, 1000, 4096], size=3))))
        _scores_and_topk_equal(ctx, idx, qs, maxes=maxes)
    finally:
        ctx.close()


@pytest.mark.parametrize("nofast", [False, True])
def test_device_index_multi_tile_equals_oracle_csr(oracle, wide, nofast):
    """sina_hip_build_index at 70 000 references: the CSR index itself (offsets and ids, downloaded)
    equals the oracle's IndexBuilder (kmer_search.cpp:152-211,245-276), and searching it gives the
    oracle's scores -- with all k-mers (no-fast) the lists are four times as many."""
    refs, cs, qs = wide
    idx = oracle.Index(cs, k=10, nofast=nofast)
    off, ids = idx.csr()
    ctx = capi.Context(0)
    try:
        ctx.upload_refs(refs.ab, refs.off, refs.width)
        ctx.build_index(10, nofast)
        goff, gids = ctx.download_index()
        assert (goff == off).all()
        assert len(gids) == len(ids) and (gids == ids).all()
        _scores_and_topk_equal(ctx, idx, qs, maxes=(41,))
    finally:
        ctx.close()


def _hip_runtime():
    L = C.CDLL("libamdhip64.so")
    L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.hipMemcpy.restype = C.c_int
    return L


def test_store_alloc_like_second_context(oracle, wide):
    """What every rank but 0 does at start-up (sina_amd/dist.py): sina_hip_store_alloc_like, the four
    buffers filled in place (a device-to-device copy stands in for the RCCL broadcast), then searches
    and alignments on forks of that context from several threads at once -- the bitmaps and the host
    copy of the reference offsets are rebuilt lazily by whichever fork gets there first.  Results
    must equal the builder context's."""
    import threading
    refs, cs, qs = wide
    hip = _hip_runtime()
    a = capi.Context(0)
    b = capi.Context(0)
    try:
        a.upload_refs(refs.ab, refs.off, refs.width)
        a.build_index(10, False)
        va = a.store_view()
        vb = capi.StoreView()
        C.memmove(C.byref(vb), C.byref(va), C.sizeof(va))
        b.store_alloc_like(vb)
        for name in ("ref_ab", "ref_off", "idx_offsets", "idx_ids"):
            assert getattr(vb, name) != getattr(va, name)
            n = getattr(va, name + "_bytes")
            assert getattr(vb, name + "_bytes") == n
            assert hip.hipMemcpy(getattr(vb, name), getattr(va, name), n, 3) == 0   # hipMemcpyDeviceToDevice
        want_topk = a.kmer_topk(qs.mask, qs.off, 41)
        fams = [want_topk[0][qi, :want_topk[2][qi]] for qi in range(qs.n)]
        foff = np.zeros(qs.n + 1, np.uint64)
        foff[1:] = np.cumsum([len(f) for f in fams])
        qm = qs.mask & 0x0f
        want_al = a.align_families(np.concatenate(fams), foff, qm, qs.off)
        forks = [b.fork() for _ in range(4)]
        got, errs = [None] * 4, []

        def work(i):
            try:
                # even forks align first (host copy of the offsets), odd ones search first (bitmaps)
                if i % 2 == 0:
                    al = forks[i].align_families(np.concatenate(fams), foff, qm, qs.off)
                    tk = forks[i].kmer_topk(qs.mask, qs.off, 41)
                else:
                    tk = forks[i].kmer_topk(qs.mask, qs.off, 41)
                    al = forks[i].align_families(np.concatenate(fams), foff, qm, qs.off)
                got[i] = (tk, al)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for tk, al in got:
            assert all((x == y).all() for x, y in zip(tk, want_topk))
            assert (al[0] == want_al[0]).all() and (al[1] == want_al[1]).all()
        assert b.stats()["n_dense_lists"] == a.stats()["n_dense_lists"] > 0
        for f in forks:
            f.close()
    finally:
        b.close()
        a.close()


# ---------------------------------------------------------------- query shapes of configs[2] / configs[4]

def _oracle_run(oracle, cs, idx, qs, qi):
    q = util.query_cseq(qs, qi, upper=False)
    ids, sc, fflog = idx.famfinder(q, oracle.ff_opts())
    if len(ids) == 0:
        return dict(status=2, log=fflog, ids=ids, sc=sc)
    r = oracle.align([cs[i] for i in ids], q, oracle.align_opts())
    r["log"] = fflog + r["log"]
    r["ids"], r["sc"] = ids, sc
    return r


def _pipeline_equals_oracle(oracle, refs, qs, key, min_dp, world=None):
    """world: (cs, idx) of `refs` if the caller has them already (the oracle's index of 100 000 references takes a
    while to build: two tests share one)."""
    cs, idx = world if world is not None else (None, None)
    if cs is None:
        cs = util.cseqs_from_refs(refs)
        idx = oracle.Index(cs, k=10)
    st = pipeline.Store(key, refs)
    try:
        st.build_index(10, False)
        pl = pipeline.Pipeline(st)
        pl.run(qs.mask, qs.off, batch=max(1, qs.n // 2), inflight=2)
        n_dp = 0
        for qi in range(qs.n):
            want = _oracle_run(oracle, cs, idx, qs, qi)
            got = pl.result(qi)
            assert got["status"] == want["status"], (qi, got["log"], want["log"])
            if want["status"] == 2:
                continue
            fam = "".join("ref%d.0:%.2f " % (i, s) for i, s in zip(want["ids"], want["sc"]))
            assert got["family"] == fam, qi
            assert (got["packed"] == want["packed"]).all(), qi
            assert (got["head"], got["tail"], got["qual"]) == (want["head"], want["tail"], want["qual"])
            if want["status"] == 0:
                assert got["log"] == want["log"]
                n_dp += 1
        assert n_dp >= min_dp
        assert st.stats()["n_dense_lists"] > 0
        pl.close()
    finally:
        st.close()


def test_v4_amplicons_vs_40k_references(oracle):
    """configs[2] shape: 250-base windows of full-length 16S against 40 000 full-length references
    (two reference tiles, bitmaps active): family, alignment, head / tail / quality and log text equal
    the oracle's."""
    refs = synth.make_refs(40000, length=1500, width=50000, seed=31)
    qs = synth.make_queries(refs, 32, seed=32, window=(1.0 / 3.0, 250))
    _pipeline_equals_oracle(oracle, refs, qs, ":mem:v4-40k", min_dp=28)


def test_23s_vs_33k_references(oracle):
    """configs[4] shape: ~3000-base queries, alignment width 150 000, 33 000 references (two tiles):
    the 256x12 DP geometry and the wide DAG build against the oracle."""
    refs = synth.make_refs(33000, length=3000, width=150000, seed=41)
    qs = synth.make_queries(refs, 8, seed=42)
    _pipeline_equals_oracle(oracle, refs, qs, ":mem:23s-33k", min_dp=8)


@pytest.fixture(scope="module")
def refs_100k(oracle):
    """The bench's 100 000-sequence reference (synthetic SILVA-NR-like clade model, width 50 000, seed 2 -- what
    bench.py builds) with the oracle's sequences and index of it: configs[1] and configs[2] both name it."""
    refs = synth.make_refs(100000, length=1500, width=50000, seed=2)
    cs = util.cseqs_from_refs(refs)
    return refs, (cs, oracle.Index(cs, k=10))


def test_16s_full_length_vs_100k_references(oracle, refs_100k):
    """configs[1] itself, end to end: 32 full-length 16S queries against the bench's 100 000-sequence reference: four
    reference tiles, dense bitmaps, the three-strip 8-column DP geometry, device DAG build, scout, backtrack,
    device-side assembly.  Family, aligned columns + case bits, head / tail / quality and the full log text equal
    the oracle's."""
    refs, world = refs_100k
    qs = synth.make_queries(refs, 32, seed=3)
    _pipeline_equals_oracle(oracle, refs, qs, ":mem:16s-100k", min_dp=30, world=world)


def test_v4_amplicons_vs_100k_references(oracle, refs_100k):
    """configs[2] at the reference count it names: 250-base windows (what `bench.py --window 250` cuts) against the
    same 100 000 references -- one DP strip of four columns per lane, no row skip, the candidate lists of the k-mer
    search at four tiles."""
    refs, world = refs_100k
    qs = synth.make_queries(refs, 48, seed=33, window=(1.0 / 3.0, 250))
    _pipeline_equals_oracle(oracle, refs, qs, ":mem:v4-100k", min_dp=40, world=world)


def test_16s_full_length_vs_500k_references(oracle):
    """configs[3] at its real shape: full-length 16S queries against 500 000 full-length references
    (width 50 000, seed 4 -- what `bench.py --refs 500000` builds): 16 reference tiles of 32 768 in the k-mer
    count kernel, full-length posting lists with the dense bitmaps active, the select kernel's tie order at
    the tile seams (kmer_search.cpp:405-412, greater<pair<int16, int>>), then DAG build, DP, walk and
    assembly.  Family, aligned columns + case bits, head / tail / quality and the full log text equal the
    oracle's at the FULL reference count."""
    refs = synth.make_refs(500000, length=1500, width=50000, seed=4)
    assert (refs.n + 32767) // 32768 == 16
    qs = synth.make_queries(refs, 16, seed=5)
    _pipeline_equals_oracle(oracle, refs, qs, ":mem:16s-500k", min_dp=15)


def test_turn_orientations_equal_oracle(oracle):
    """--turn none / revcomp / all (famfinder.cpp:312-378; the shape of the reference's own
    famfinder_test.cpp:89-116): queries handed in reversed, complemented, or both are recognised by
    the four top-1 k-mer searches, turned back, and then align exactly like the oracle aligns the
    sequence the oracle's turn_check picks."""
    refs = synth.make_refs(500, length=320, width=3200, seed=51)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    base = synth.make_queries(refs, 12, seed=57)
    comp = np.zeros(32, np.uint8)
    for m in range(32):   # A<->T/U, G<->C, case bit kept (aligned_base.h:117-124)
        comp[m] = ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 1) << 3) | ((m & 8) >> 3) | (m & 16)
    masks, handed = [], []
    for qi in range(base.n):
        m = base.seq(qi).copy()
        o = qi % 4
        if o & 1:
            m = m[::-1].copy()
        if o & 2:
            m = comp[m]
        masks.append(m)
        handed.append(o)
    qoff = np.zeros(base.n + 1, np.int64)
    qoff[1:] = np.cumsum([len(m) for m in masks])
    qs = synth.QuerySet(mask=np.concatenate(masks), off=qoff, src=base.src)
    st = pipeline.Store(":mem:turn", refs)
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    try:
        for mode, all_o in (("none", None), ("revcomp", False), ("all", True)):
            pl = pipeline.Pipeline(st, famfinder=dict(ff, turn=mode))
            pl.run(qs.mask, qs.off, batch=5, inflight=2)
            n_turned = 0
            for qi in range(qs.n):
                q = util.query_cseq(qs, qi, upper=False)
                got = pl.result(qi)
                if all_o is None:
                    assert pl.attr(qi, "turn") == "turn-check disabled"
                    o = 0
                else:
                    o, sc4 = idx.turn_check(q, all_o)
                    assert pl.attr(qi, "turn") == oracle.Index.TURN_NAMES[o], (mode, qi, sc4)
                    if all_o:
                        assert o == handed[qi]      # a turned 16S-like query is always recognised
                    n_turned += o != 0
                if o & 1:
                    oracle.lib().so_cseq_reverse(q.h)
                if o & 2:
                    oracle.lib().so_cseq_complement(q.h)
                ids, sc, fflog = idx.famfinder(q, oracle.ff_opts(fs_min_len=100, fs_full_len=250))
                if len(ids) == 0:
                    assert got["status"] == 2 and got["log"] == fflog
                    continue
                want = oracle.align([cs[i] for i in ids], q, oracle.align_opts())
                assert got["status"] == want["status"], (mode, qi)
                assert (got["packed"] == want["packed"]).all(), (mode, qi)
                assert (got["head"], got["tail"], got["qual"]) == (want["head"], want["tail"], want["qual"])
            if all_o is not None:
                assert n_turned >= (9 if all_o else 3)
            pl.close()
    finally:
        st.close()


def test_500k_references_properties():
    """BASELINE.json configs[3] scale for the reference side: 500 000 (short) references = 16
    reference tiles.  No oracle at this size; size-independent properties instead: the device index
    holds every reference's k-mers (a query that IS a reference scores its own k-mer count, and
    nothing scores higher), results do not depend on how the queries are cut into launches, and two
    runs are identical; through the whole pipeline such queries come back with the alignment of the
    reference they copy."""
    refs = synth.make_refs(500000, length=200, width=2000, seed=61, n_clades=40)
    qs = synth.make_queries(refs, 48, seed=62, sub=0.0, dele=0.0, ins=0.0)
    ctx = capi.Context(0)
    try:
        ctx.upload_refs(refs.ab, refs.off, refs.width)
        ctx.build_index(10, False)
        assert ctx.store_view().n_postings > 10 * refs.n
        full = ctx.kmer_topk(qs.mask, qs.off, 41)
        again = ctx.kmer_topk(qs.mask, qs.off, 41)
        assert all((x == y).all() for x, y in zip(full, again))
        half = 24
        lo = ctx.kmer_topk(qs.mask[:qs.off[half]], qs.off[:half + 1], 41)
        hi = ctx.kmer_topk(qs.mask[qs.off[half]:], qs.off[half:] - qs.off[half], 41)
        for part, sl in ((lo, slice(0, half)), (hi, slice(half, qs.n))):
            assert all((x == y[sl]).all() for x, y in zip(part, full))
        assert ctx.stats()["n_dense_lists"] > 0
        for qi in range(qs.n):
            m = qs.seq(qi) & 0x0f
            # k-mers with multiplicity: windows of 10 unambiguous bases ending before the last base
            # whose first base is A (kmer.h:69-83,122-124,188-201)
            nk = sum(1 for e in range(9, len(m) - 1) if m[e - 9] == 1)
            row = ctx.kmer_scores(qs.seq(qi))
            assert row.max() == nk == full[1][qi, 0]
            assert row[qs.src[qi]] == nk
            assert (np.sort(row)[::-1][:41] == full[1][qi]).all()
            ties = np.flatnonzero(row == nk)
            assert full[0][qi, 0] == ties.max()                  # (score desc, id desc)
    finally:
        ctx.close()
    st = pipeline.Store(":mem:500k", refs)
    try:
        st.build_index(10, False)
        pl = pipeline.Pipeline(st, famfinder={"fs-min-len": 100, "fs-full-len": 180})
        pl.run(qs.mask, qs.off, batch=16, inflight=2)
        res = [pl.result(qi) for qi in range(qs.n)]
        pl.run(qs.mask, qs.off, batch=48, inflight=1)
        n_copy = 0
        for qi in range(qs.n):
            again = pl.result(qi)
            assert again["status"] == res[qi]["status"] and (again["packed"] == res[qi]["packed"]).all()
            assert again["family"] == res[qi]["family"]
            assert res[qi]["status"] in (0, 1), (qi, res[qi]["log"], len(qs.seq(qi)))
            got_bases = (res[qi]["packed"] >> 24) & 0x0f
            assert (got_bases == (qs.seq(qi) & 0x0f)).all()       # every base placed, in order
            if res[qi]["status"] == 1:                            # copied from a reference that contains it
                n_copy += 1
                assert res[qi]["qual"] == 100
        assert n_copy >= 40
        pl.close()
    finally:
        st.close()


def test_launch_ranges_cut_by_traceback_budget(oracle):
    """A batch whose trace-back planes exceed the budget is aligned in several DP launches, each a
    whole number of rounds of wave slots (ctx.h dp_round_range): the alignments must not depend on
    where the batch is cut -- same results with a 3 GB and with the default budget -- and a sample is
    compared with the oracle.  7000 queries of ~300 bases (one strip of B = 8: 3072 wave slots)."""
    refs = synth.make_refs(600, length=300, width=1500, seed=71, n_clades=6, long_del_prob=0.0)
    qs = synth.make_queries(refs, 7000, seed=72)
    cs = util.cseqs_from_refs(refs)
    runs = []
    for gb in ("3", None):  # (3 GB: about 5000 of these queries -> cut back to one round of 3072)
        old = os.environ.get("SINA_HIP_TB_GB")
        if gb is None:
            os.environ.pop("SINA_HIP_TB_GB", None)
        else:
            os.environ["SINA_HIP_TB_GB"] = gb
        st = pipeline.Store(":mem:tbcut%s" % gb, refs)
        try:
            st.build_index(10, False)
            pl = pipeline.Pipeline(st)
            pl.run(qs.mask, qs.off, batch=7000, inflight=1)
            runs.append([pl.result(qi) for qi in range(qs.n)])
            pl.close()
        finally:
            st.close()
            if old is None:
                os.environ.pop("SINA_HIP_TB_GB", None)
            else:
                os.environ["SINA_HIP_TB_GB"] = old
    small, big = runs
    for qi in range(qs.n):
        assert small[qi]["status"] == big[qi]["status"] and (small[qi]["packed"] == big[qi]["packed"]).all(), qi
        assert (small[qi]["head"], small[qi]["tail"], small[qi]["qual"]) == (big[qi]["head"], big[qi]["tail"], big[qi]["qual"])
    assert sum(1 for r in small if r["status"] == 0) > 6000
    idx = oracle.Index(cs, k=10)
    for qi in (0, 1, 2, 3071, 3072, 3073, 3500, 6143, 6144, 6145, 6998, 6999):  # (around the launch cuts)
        want = _oracle_run(oracle, cs, idx, qs, qi)
        got = small[qi]
        assert got["status"] == want["status"], (qi, got["log"], want["log"])
        if want["status"] != 2:
            assert (got["packed"] == want["packed"]).all(), qi
            assert (got["head"], got["tail"], got["qual"]) == (want["head"], want["tail"], want["qual"])
