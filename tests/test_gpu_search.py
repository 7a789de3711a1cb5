"""GPU parity for the search stage: the comparison kernel's counters against the
oracle's literal traverse(), for every iupac rule, with and without the lower-case filter."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from sina_amd import capi, pipeline, synth
from tests import util

pytestmark = pytest.mark.gpu


def _aligned_queries(refs, n, seed, lower_rate=0.0, amb_rate=0.0):
    """Aligned query sequences: references with substitutions, a trimmed window, some bases moved to
    a free neighbouring column and some dropped -- strictly ascending columns."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        ab = refs.seq(int(rng.integers(refs.n))).copy()
        a, b = sorted(rng.integers(0, len(ab), size=2))
        if b - a < 30:
            a, b = 0, len(ab)
        ab = ab[a:b]
        keep = rng.random(len(ab)) > 0.03
        ab = ab[keep]
        pos = (ab & 0xFFFFFF).astype(np.int64)
        mask = (ab >> 24).astype(np.uint32)
        sub = rng.random(len(ab)) < 0.05
        mask[sub] = rng.choice([1, 2, 4, 8], size=int(sub.sum()))
        amb = rng.random(len(ab)) < amb_rate
        mask[amb] = rng.choice([3, 5, 7, 15, 12], size=int(amb.sum()))
        low = rng.random(len(ab)) < lower_rate
        mask[low] |= 0x10
        # shift a few bases one column to the right when that column is free
        for x in np.nonzero(rng.random(len(ab)) < 0.04)[0]:
            nxt = pos[x + 1] if x + 1 < len(pos) else refs.width
            if pos[x] + 1 < nxt:
                pos[x] += 1
        out.append((pos.astype(np.uint32) | (mask << 24)).astype(np.uint32))
    return out


@pytest.mark.parametrize("filter_lc", [False, True])
def test_compare_counts_equal_traverse(oracle, gpu_ctx, filter_lc):
    refs = synth.make_refs(150, length=300, width=2500, seed=401, amb_rate=0.02, lower_rate=0.1)
    cs = util.cseqs_from_refs(refs)
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    qs = _aligned_queries(refs, 9, 402, lower_rate=0.15, amb_rate=0.03)
    # one query without any upper-case base in the middle of a run, one fully lower case at both ends
    qs[3] = qs[3] | np.uint32(0)
    qs[4][:7] |= np.uint32(0x10 << 24)
    qs[4][-5:] |= np.uint32(0x10 << 24)
    rng = np.random.default_rng(403)
    cand = [rng.choice(refs.n, size=int(k), replace=False).astype(np.uint32) for k in (150, 1, 40, 7, 99, 3, 64, 20, 150)]
    q_off = np.zeros(len(qs) + 1, np.uint64)
    q_off[1:] = np.cumsum([len(x) for x in qs])
    c_off = np.zeros(len(qs) + 1, np.uint64)
    c_off[1:] = np.cumsum([len(x) for x in cand])
    for rule, name in enumerate(("optimistic", "pessimistic", "exact")):
        got = gpu_ctx.compare(np.concatenate(qs), q_off, np.concatenate(cand), c_off, rule, filter_lc)
        for qi, q_ab in enumerate(qs):
            q = po.Cseq.from_packed("q%d" % qi, q_ab, refs.width)
            for x, rid in enumerate(cand[qi]):
                want = po.compare_counts(q, cs[int(rid)], name, filter_lc)
                assert tuple(got[int(c_off[qi]) + x]) == want, (name, qi, int(rid))


def test_compare_reference_against_itself(oracle, gpu_ctx):
    """Properties that need no oracle: a sequence against itself is all matches; swapping a base
    for a different one moves exactly one count from match to mismatch."""
    refs = synth.make_refs(60, length=1500, width=50000, seed=411)
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    ids = np.arange(0, 60, 7, dtype=np.uint32)
    qs = [refs.seq(int(i)).copy() for i in ids]
    for q in qs[::2]:
        m = (q[100] >> 24) & 0xf
        q[100] = (q[100] & 0xFFFFFF) | (np.uint32(8 if m != 8 else 1) << 24)
    q_off = np.zeros(len(qs) + 1, np.uint64)
    q_off[1:] = np.cumsum([len(x) for x in qs])
    got = gpu_ctx.compare(np.concatenate(qs), q_off, ids, np.arange(len(ids) + 1, dtype=np.uint64), 2, False)
    for x, q in enumerate(qs):
        changed = 1 if x % 2 == 0 else 0
        assert tuple(got[x]) == (0, 0, 0, 0, len(q) - changed, changed)


def _taxonomy(i):
    phyla = ["Proteobacteria", "Firmicutes", "Bacteroidota"]
    return "Bacteria;%s;class%d;order%d;" % (phyla[i % 3], i % 6, i % 12)


@pytest.mark.parametrize("sopts,oopts", [
    ({}, {}),
    ({"search-min-sim": 0.5, "search-max-result": 7, "search-iupac": "pessimistic", "search-cover": "overlap",
      "lca-quorum": 0.5},
     dict(min_sim=0.5, max_result=7, iupac="pessimistic", cover="overlap", lca_quorum=0.5)),
    # (no --search-correction jc here: Jukes-Cantor of an IDENTITY above 0.75 is NaN, and the order
    # partial_sort leaves NaN scores in is unspecified in the reference itself)
    ({"search-kmer-candidates": 40, "search-cover": "all", "search-min-sim": -1, "search-filter-lowercase": True,
      "search-iupac": "exact"},
     dict(kmer_candidates=40, cover="all", min_sim=-1, filter_lc=1, iupac="exact")),
    ({"search-all": True, "search-ignore-super": True, "search-max-result": 5},
     dict(search_all=1, ignore_super=1, max_result=5)),
    ({"search-ignore-super": True, "search-min-sim": 0.0}, dict(ignore_super=1, min_sim=0.0)),
])
def test_pipeline_with_search_stage(oracle, sopts, oopts):
    """tray -> famfinder -> aligner -> search_filter through the C++ stage mirror (k-mer search, DP and the
    1000-candidate comparison on the GPU) vs the oracle run query by query: result ids, score bits,
    nearest_slv and the LCA classification."""
    _search_stage_case(oracle, 420, sopts, oopts)


@pytest.mark.parametrize("sopts,oopts", [
    ({}, {}),
    ({"search-all": True, "search-max-result": 12}, dict(search_all=1, max_result=12)),
])
def test_search_stage_at_20k_references(oracle, sopts, oopts):
    """The same at a reference count where the stage works as in production: the 1000 candidates are a
    real selection (top 1000 of 20 000 by k-mer score: the select kernel's histogram path with M =
    1000), and --search-all compares every query with all 20 000 references."""
    _search_stage_case(oracle, 20000, sopts, oopts, n_queries=6)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SINA_FUZZ_SEEDS", "10"))))
def test_search_stage_option_fuzz(oracle, seed):
    """Seeded random search-stage options (search_filter.cpp:244-345, cseq_comparator.cpp:240-290: all
    nine coverage rules, three IUPAC rules, lower-case filter, candidate counts below and above the
    reference count, --search-all, --search-ignore-super, quorum) against the oracle."""
    rng = np.random.default_rng(8000 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]  # noqa: E731
    iupac = pick(["optimistic", "pessimistic", "exact"])
    cover = pick(["abs", "query", "target", "overlap", "all", "average", "min", "max", "nogap"])
    o = dict(min_sim=float(pick([0.7, 0.5, 0.0, -1.0, 0.9])), max_result=int(pick([10, 1, 7, 50])), iupac=iupac, cover=cover,
             lca_quorum=float(pick([0.7, 0.5, 1.0])), kmer_candidates=int(pick([1000, 40, 300, 5])),
             filter_lc=int(pick([0, 1])), search_all=int(pick([0, 0, 1])), ignore_super=int(pick([0, 1])))
    names = dict(min_sim="search-min-sim", max_result="search-max-result", iupac="search-iupac", cover="search-cover",
                 lca_quorum="lca-quorum", kmer_candidates="search-kmer-candidates", filter_lc="search-filter-lowercase",
                 search_all="search-all", ignore_super="search-ignore-super")
    sopts = {names[k]: (bool(v) if k in ("filter_lc", "search_all", "ignore_super") else v) for k, v in o.items()}
    _search_stage_case(oracle, 420, sopts, o, n_queries=10)


def _search_stage_case(oracle, n_refs, sopts, oopts, n_queries=24):
    refs = synth.make_refs(n_refs, length=300, width=3000, seed=451, amb_rate=0.01, lower_rate=0.03)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    st = pipeline.Store(":mem:gpu-search-%d" % n_refs, refs)
    acc = ["ref%d" % i for i in range(refs.n)]
    ver = [str(1 + i % 3) for i in range(refs.n)]
    start = [str(i % 7) for i in range(refs.n)]
    stop = [str(1400 + i) for i in range(refs.n)]
    for i in range(refs.n):
        st.set_attr(i, "version", ver[i])
        st.set_attr(i, "start", start[i])
        st.set_attr(i, "stop", stop[i])
        st.set_attr(i, "tax_slv", _taxonomy(i))
    qs = synth.make_queries(refs, n_queries, seed=452, amb_rate=0.01, lower_rate=0.05)
    # three more queries that are exact pieces of references (contained: --search-ignore-super matters)
    extra = [((refs.seq(i) >> 24) & 0xff).astype(np.uint8)[a:b] for i, a, b in ((5, 10, 250), (77, 0, 200), (300, 40, 290))]
    masks = [qs.seq(i) for i in range(qs.n)] + extra
    off = np.zeros(len(masks) + 1, np.int64)
    off[1:] = np.cumsum([len(m) for m in masks])
    qs = synth.QuerySet(mask=np.concatenate(masks), off=off, src=np.zeros(len(masks), np.int64))
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    pl = pipeline.Pipeline(st, famfinder=ff, aligner={"realign": True},
                           search=dict({"lca-fields": "tax_slv"}, **sopts))
    pl.run(qs.mask, qs.off, batch=10, inflight=2)
    so = oracle.search_opts(**oopts)
    n_hits = 0
    for qi in range(qs.n):
        got = pl.result(qi)
        q = util.query_cseq(qs, qi, upper=False)
        ids, sc, _ = idx.famfinder(q, oracle.ff_opts(fs_min_len=100, fs_full_len=250))
        if len(ids) == 0:
            assert got["search_ids"] is None
            continue
        al = oracle.align([cs[i] for i in ids], q, oracle.align_opts(realign=1))
        if al["status"] not in (0, 1):
            assert got["search_ids"] is None
            continue
        aligned = po.Cseq.from_packed("query%d" % qi, al["packed"], al["width"])
        want_ids, want_sc, _ = oracle.search(idx, aligned, so)
        assert (got["search_ids"] == want_ids).all(), qi
        assert (util.f32_bits(got["search_scores"]) == util.f32_bits(want_sc)).all(), qi
        assert pl.attr(qi, "nearest_slv") == po.search_nearest(acc, ver, start, stop, want_ids, want_sc)
        assert pl.attr(qi, "lca_tax_slv") == po.search_lca([_taxonomy(int(i)) for i in want_ids], so.lca_quorum)
        n_hits += len(want_ids)
    assert n_hits > 0
    pl.close()
    st.close()



def test_fasta_database_keeps_its_index_in_a_sidx_file(oracle, tmp_path):
    """a file-backed database builds its index on the GPU once, writes it beside the
    database as <name>.sidx in the reference's format (equal to the oracle's writer), and loads it from
    there the next time -- with the same search results either way."""
    import os
    refs = synth.make_refs(260, length=300, width=2600, seed=471, amb_rate=0.01)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    db = str(tmp_path / "refs.fasta")
    with open(db, "w") as f:
        for i in range(refs.n):
            f.write(">ref%d some description\n%s\n" % (i, synth.aligned_string(refs.seq(i), refs.width)))
    qs = synth.make_queries(refs, 12, seed=472)
    ff = {"fs-min-len": 100, "fs-full-len": 250}

    def run():
        st = pipeline.Store.open(db)
        pl = pipeline.Pipeline(st, famfinder=ff)
        pl.run(qs.mask, qs.off)
        fam = [pl.result(q)["family"] for q in range(qs.n)]
        origin = st.index_origin()
        pl.close()
        st.close()
        return fam, origin

    fam1, origin1 = run()
    sidx = str(tmp_path / "refs.sidx")
    assert origin1 == "built" and os.path.exists(sidx)
    want = str(tmp_path / "want.sidx")
    idx.write_sidx(want)
    a, b = open(want, "rb").read(), open(sidx, "rb").read()
    keep = lambda x: x[:10] + x[12:18] + x[24:]
    assert keep(a) == keep(b)
    fam2, origin2 = run()
    assert origin2 == "loaded " + sidx
    assert fam1 == fam2
    for q in range(qs.n):
        ids, sc, _ = idx.famfinder(util.query_cseq(qs, q, upper=False), oracle.ff_opts(fs_min_len=100, fs_full_len=250))
        assert fam1[q] == "".join("ref%d.0:%.2f " % (i, s) for i, s in zip(ids, sc))
    # an index file older than the database is not trusted
    os.utime(sidx, (1, 1))
    _, origin3 = run()
    assert origin3 == "built"


def test_fasta_database_in_arb_id_order(oracle, tmp_path):
    """SURVEY 8f-2, second half: a database opened with the reference's id order (the walk of its
    unordered_map<string, ..., boost::hash<string>>, host/id_order.cpp) numbers, indexes and reports its sequences
    in that order: the .sidx it writes is the oracle's for the permuted references, names included, and the
    families are the oracle's over the permuted references (ties between equal k-mer scores go to the larger id)."""
    import os
    refs = synth.make_refs(300, length=300, width=2600, seed=481, amb_rate=0.01)
    names = ["Acc%05d" % (7919 * i % 100000) for i in range(refs.n)]
    db = str(tmp_path / "arbdb.fasta")
    with open(db, "w") as f:
        for i in range(refs.n):
            f.write(">%s\n%s\n" % (names[i], synth.aligned_string(refs.seq(i), refs.width)))
    order, _, _ = pipeline.reference_order(names)
    assert sorted(order.tolist()) == list(range(refs.n)) and order.tolist() != list(range(refs.n))
    cs = [oracle.Cseq.from_packed(names[j], refs.seq(int(j)), refs.width) for j in order]
    idx = oracle.Index(cs, k=10)
    qs = synth.make_queries(refs, 16, seed=482)
    st = pipeline.Store.open(db, id_order="arb")
    assert [st.name(i) for i in range(refs.n)] == [names[j] for j in order]
    pl = pipeline.Pipeline(st, famfinder={"fs-min-len": 100, "fs-full-len": 250})
    pl.run(qs.mask, qs.off)
    for q in range(qs.n):
        ids, sc, _ = idx.famfinder(util.query_cseq(qs, q, upper=False), oracle.ff_opts(fs_min_len=100, fs_full_len=250))
        assert pl.result(q)["family"] == "".join("%s.0:%.2f " % (names[order[i]], s) for i, s in zip(ids, sc))
    assert st.index_origin() == "built"
    pl.close()
    st.close()
    want = str(tmp_path / "want.sidx")
    idx.write_sidx(want, names=[names[j] for j in order])
    a, b = open(want, "rb").read(), open(str(tmp_path / "arbdb.sidx"), "rb").read()
    keep = lambda x: x[:10] + x[12:18] + x[24:]
    assert keep(a) == keep(b)


def test_sidx_written_under_one_id_order_is_renumbered_under_the_other(oracle, tmp_path):
    """A <db>.sidx holds ids that count ITS OWN name list (the reference resolves them through those names,
    kmer_search.cpp try_load).  The same database opened first in file order (index built + cached), then in
    ARB order, must not take the cached posting ids at face value: they are renumbered by name (or the cache is
    rebuilt) and the families are those of an index built for that order -- and the other way round; a cache
    whose names are not this database's is rebuilt."""
    import os
    refs = synth.make_refs(280, length=300, width=2600, seed=491, amb_rate=0.01)
    names = ["Acc%05d" % (104729 * i % 100000) for i in range(refs.n)]
    db = str(tmp_path / "both.fasta")
    with open(db, "w") as f:
        for i in range(refs.n):
            f.write(">%s\n%s\n" % (names[i], synth.aligned_string(refs.seq(i), refs.width)))
    order, _, _ = pipeline.reference_order(names)
    qs = synth.make_queries(refs, 12, seed=492)
    ffo = {"fs-min-len": 100, "fs-full-len": 250}

    def run(id_order):
        st = pipeline.Store.open(db, id_order=id_order) if id_order else pipeline.Store.open(db)
        pl = pipeline.Pipeline(st, famfinder=ffo)
        pl.run(qs.mask, qs.off)
        fam = [pl.result(q)["family"] for q in range(qs.n)]
        origin = st.index_origin()
        pl.close()
        st.close()
        return fam, origin

    def want(perm):
        cs = [oracle.Cseq.from_packed(names[j], refs.seq(int(j)), refs.width) for j in perm]
        idx = oracle.Index(cs, k=10)
        out = []
        for q in range(qs.n):
            ids, sc, _ = idx.famfinder(util.query_cseq(qs, q, upper=False), oracle.ff_opts(fs_min_len=100, fs_full_len=250))
            out.append("".join("%s.0:%.2f " % (names[perm[i]], s) for i, s in zip(ids, sc)))
        return out

    sidx = str(tmp_path / "both.sidx")
    want_file, want_arb = want(list(range(refs.n))), want([int(j) for j in order])
    fam, origin = run(None)
    assert origin == "built" and fam == want_file
    fam, origin = run("arb")                     # the cache is in file order
    assert origin == "loaded (ids renumbered by name) " + sidx and fam == want_arb
    os.remove(sidx)
    fam, origin = run("arb")
    assert origin == "built" and fam == want_arb
    fam, origin = run(None)                      # the cache is in ARB order now
    assert origin == "loaded (ids renumbered by name) " + sidx and fam == want_file
    fam, origin = run("arb")
    assert origin == "loaded " + sidx and fam == want_arb
    # a cache of the right size whose names belong to another database: rebuilt
    blob = open(sidx, "rb").read()
    at = blob.index(names[int(order[0])].encode())
    open(sidx, "wb").write(blob[:at] + b"Zzz" + blob[at + 3:])
    fam, origin = run("arb")
    assert origin == "built" and fam == want_arb


def test_fasta_pipeline_and_show_dist_metrics(oracle, tmp_path):
    """FASTA in -> famfinder -> aligner -> FASTA out with --show-dist, run the way the
    reference's accuracy test does (tests/accuracy_kmer.test: the database's own sequences, --realign
    --fs-leave-query-out): written alignments and avg_sps / avg_cpm / avg_idty against the oracle."""
    refs = synth.make_refs(300, length=320, width=3200, seed=481, amb_rate=0.01)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    st = pipeline.Store(":mem:gpu-fasta", refs)
    pick = list(range(0, 300, 12))
    src = str(tmp_path / "in.fasta")
    with open(src, "w") as f:
        for i in pick:
            f.write(">ref%d\n%s\n" % (i, synth.aligned_string(refs.seq(i), refs.width)))
    dst = str(tmp_path / "out.fasta")
    ffo = {"fs-min-len": 100, "fs-full-len": 250, "fs-leave-query-out": True}
    got = pipeline.run_fasta(st, src, dst, famfinder=ffo, aligner={"realign": True}, show_dist=True, batch=7,
                             log_path=str(tmp_path / "log.txt"))
    assert got["read"] == len(pick) and got["skipped"] == 0
    out_lines = open(dst).read().splitlines()
    tot_sps = tot_cpm = tot_idty = 0.0
    n = 0
    for i in pick:
        orig = cs[i]
        ids, sc, _ = idx.famfinder(orig, oracle.ff_opts(fs_min_len=100, fs_full_len=250, fs_leave_query_out=1))
        assert i not in ids
        al = oracle.align([cs[j] for j in ids], orig, oracle.align_opts(realign=1))
        assert al["status"] == 0 and "removed from family" not in al["log"]
        assert out_lines[2 * n] == ">ref%d" % i and out_lines[2 * n + 1] == al["aligned"].replace(".", "-")
        aligned = po.Cseq.from_packed("ref%d" % i, al["packed"], al["width"])
        sps = po.compare(orig, aligned, "exact", "none", "query")
        scored = sorted((float(po.compare(orig, cs[int(j)], "optimistic", "none", "query")), "ref%d" % j, int(j))
                        for j in ids)
        closest = scored[-1]
        orig_idty = np.float32(closest[0])
        aligned_idty = po.compare(aligned, cs[closest[2]], "optimistic", "none", "query")
        tot_sps += float(sps)
        tot_idty += float(orig_idty)
        tot_cpm += float(np.float32(orig_idty - aligned_idty))
        n += 1
    assert got["aligned"] == n and got["written"] == n
    assert got["avg_sps"] == tot_sps / n and got["avg_cpm"] == tot_cpm / n and got["avg_idty"] == tot_idty / n
    assert got["avg_sps"] > 0.97   # (the reference's own accuracy test asks for > 0.996 on real rRNA)
    st.close()


@pytest.mark.parametrize("fasta,search", [({}, None), ({"fasta-write-dots": True, "line-length": 70, "meta-fmt": "comment"}, {}),
                                          ({"meta-fmt": "csv", "fasta-write-dna": True}, None)])
def test_fasta_driver_with_concurrent_stages_writes_the_serial_drivers_bytes(oracle, tmp_path, fasta, search):
    """SURVEY 8f-3 at speed: the FASTA driver runs reader, famfinder, aligner (+ search) and the sink as concurrent
    nodes (src/sina.cpp:452-586) -- the sink renders the log reports and composes the records of a batch on the loop
    pool and hands the text over in input order.  Output file, csv side file, log and summary are byte for byte
    those of the one-batch-at-a-time driver (the aligned_slv time stamp aside); batches of 16 make 20 hand-overs."""
    import os
    refs = synth.make_refs(400, length=320, width=3200, seed=495, amb_rate=0.01, lower_rate=0.02)
    st = pipeline.Store(":mem:gpu-fasta-conc", refs)
    qs = synth.make_queries(refs, 310, seed=496, ins=0.01, dele=0.01, lower_rate=0.03)
    src = str(tmp_path / "in.fasta")
    with open(src, "w") as f:
        for i in range(qs.n):
            text = synth.bases_string(qs.seq(i))
            if i == 17:
                text = text[:40] + "!" + text[41:]          # a record the reader drops
            if i == 23:
                text = "ACGU"                               # one without relatives
            f.write(">q%d sample %d\n%s\n" % (i, i % 7, text))
    ffo = {"fs-min-len": 100, "fs-full-len": 250}
    if search is not None:
        search = {"search-min-sim": 0.5, "search-kmer-candidates": 50}
    outs = {}
    for serial in (True, False):
        tag = "serial" if serial else "conc"
        dst, logp = str(tmp_path / (tag + ".fasta")), str(tmp_path / (tag + ".log"))
        got = pipeline.run_fasta(st, src, dst, famfinder=ffo, search=search, fasta=fasta, show_dist=False, batch=16,
                                 log_path=logp, serial=serial)
        import re
        # (the aligned_slv time stamp: its line in logs and meta comments, its cell in the csv rows)
        strip = lambda text: re.sub(r"\d\d:\d\d:\d\d", "hh:mm:ss", "\n".join(l for l in text.splitlines() if "aligned_slv" not in l))
        csv = str(tmp_path / (tag + ".csv"))
        outs[tag] = (got, strip(open(dst).read()), strip(open(logp).read()),
                     strip(open(csv).read()) if os.path.exists(csv) else None)
    assert outs["serial"][0] == outs["conc"][0]
    assert outs["serial"][0]["read"] == qs.n - 1 and outs["serial"][0]["skipped"] == 1
    assert outs["serial"][0]["written"] >= qs.n - 12
    assert outs["serial"][1] == outs["conc"][1]
    assert outs["serial"][2] == outs["conc"][2]
    assert outs["serial"][3] == outs["conc"][3] and (outs["serial"][3] is not None) == (fasta.get("meta-fmt") == "csv")
    st.close()


def test_fasta_driver_stops_cleanly_when_its_source_fails_in_mid_run(oracle, tmp_path):
    """The concurrent FASTA driver with an input that breaks off (a truncated gzip file: the reader thread throws
    after it has handed over several batches): the error reaches the caller, no thread is left waiting on a queue,
    and the store and its contexts serve the next run as if nothing had happened."""
    import gzip
    refs = synth.make_refs(300, length=320, width=3200, seed=497)
    st = pipeline.Store(":mem:gpu-fasta-broken", refs)
    qs = synth.make_queries(refs, 400, seed=498)
    good, bad = str(tmp_path / "in.fasta.gz"), str(tmp_path / "broken.fasta.gz")
    with gzip.open(good, "wt", compresslevel=1) as f:
        for i in range(qs.n):
            f.write(">q%d\n%s\n" % (i, synth.bases_string(qs.seq(i))))
    whole = open(good, "rb").read()
    open(bad, "wb").write(whole[:(2 * len(whole)) // 3])
    ffo = {"fs-min-len": 100, "fs-full-len": 250}
    with pytest.raises(pipeline.HostError):
        pipeline.run_fasta(st, bad, str(tmp_path / "out1.fasta"), famfinder=ffo, batch=16)
    got = pipeline.run_fasta(st, good, str(tmp_path / "out2.fasta"), famfinder=ffo, batch=16)
    ser = pipeline.run_fasta(st, good, str(tmp_path / "out3.fasta"), famfinder=ffo, batch=16, serial=True)
    assert got == ser and got["read"] == qs.n and got["written"] >= qs.n - 10
    assert open(str(tmp_path / "out2.fasta")).read() == open(str(tmp_path / "out3.fasta")).read()
    st.close()
