"""GPU end-to-end: tray -> famfinder -> aligner through the C++ stage mirror (k-mer
search and mesh DP on the GPU) vs the CPU oracle run query by query."""
import os

import numpy as np
import pytest

from sina_amd import pipeline, synth
from tests import util

pytestmark = pytest.mark.gpu


def _oracle_run(oracle, cs, idx, qs, qi, ff=None, al=None):
    q = util.query_cseq(qs, qi, upper=False)
    ids, sc, fflog = idx.famfinder(q, oracle.ff_opts(**(ff or {})))
    if len(ids) == 0:
        return dict(status=2, log=fflog, ids=ids, sc=sc)
    r = oracle.align([cs[i] for i in ids], q, oracle.align_opts(**(al or {})))
    r["log"] = fflog + r["log"]
    r["ids"], r["sc"] = ids, sc
    return r


def _check(oracle, refs, qs, pl, cs, idx, ff=None, al=None):
    n_dp = n_copy = 0
    for qi in range(qs.n):
        want = _oracle_run(oracle, cs, idx, qs, qi, ff, al)
        got = pl.result(qi)
        fam = "".join("ref%d.0:%.2f " % (i, s) for i, s in zip(want["ids"], want["sc"]))
        if want["status"] == 2:
            assert got["status"] == 2 and got["log"] == want["log"]
            continue
        assert got["family"] == fam
        assert got["status"] == want["status"], (qi, got["log"], want["log"])
        assert synth.aligned_string(got["packed"], got["width"]) == want["aligned"].replace(".", "-")
        assert (got["packed"] == want["packed"]).all()          # columns AND case bits
        assert (got["head"], got["tail"], got["qual"]) == (want["head"], want["tail"], want["qual"])
        if want["status"] == 0:
            assert got["log"] == want["log"]                      # NAST + scoring text
            n_dp += 1
        else:
            n_copy += 1
    return n_dp, n_copy


@pytest.fixture(scope="module")
def world(oracle):
    refs = synth.make_refs(500, length=320, width=3200, seed=51, amb_rate=0.01, lower_rate=0.02)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    st = pipeline.Store(":mem:gpu-e2e", refs)
    yield refs, cs, idx, st
    st.close()


def test_pipeline_defaults(oracle, world):
    refs, cs, idx, st = world
    qs = synth.make_queries(refs, 40, seed=52, amb_rate=0.01, lower_rate=0.05)
    pl = pipeline.Pipeline(st, famfinder={"fs-min-len": 100, "fs-full-len": 250})
    pl.run(qs.mask, qs.off, batch=16, inflight=2)
    n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250))
    assert n_dp >= 35
    pl.close()


@pytest.mark.parametrize("al,oal", [
    ({"overhang": "remove"}, dict(overhang=1)),
    ({"overhang": "edge", "lowercase": "unaligned"}, dict(overhang=2, lowercase=2)),
    ({"insertion": "forbid"}, dict(insertion=1)),
    ({"lowercase": "original", "fs-weight": 0.5, "pen-gap": 4, "pen-gapext": 1.5, "match-score": 3,
      "mismatch-score": -2}, dict(lowercase=1, fs_weight=0.5, gap_penalty=4, gap_ext_penalty=1.5, match_score=3,
                                  mismatch_score=-2)),
])
def test_pipeline_aligner_options(oracle, world, al, oal):
    refs, cs, idx, st = world
    # V4-like windows give head/tail overhang; extra indels give insertions to place
    qs = synth.make_queries(refs, 24, seed=53, window=(0.3, 120), ins=0.02, dele=0.02, lower_rate=0.05)
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    pl = pipeline.Pipeline(st, famfinder=ff, aligner=al)
    pl.run(qs.mask, qs.off, batch=24, inflight=1)
    n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250), al=oal)
    assert n_dp >= 20
    pl.close()


@pytest.mark.parametrize("geom", [None, "128,12", "128,4"])
@pytest.mark.parametrize("al,oal", [
    ({}, {}),
    ({"insertion": "forbid", "overhang": "remove"}, dict(insertion=1, overhang=1)),
    ({"lowercase": "unaligned", "overhang": "edge", "pen-gap": 4, "pen-gapext": 1.5, "match-score": 3,
      "mismatch-score": -2}, dict(lowercase=2, overhang=2, gap_penalty=4, gap_ext_penalty=1.5, match_score=3,
                                  mismatch_score=-2)),
])
def test_pipeline_fs_no_graph_profile(oracle, world, monkeypatch, geom, al, oal):
    """--fs-no-graph (src/align.cpp:428-433): the family as a profile (pseq) scored with
    scoring_scheme_profile.  The host builds the column chain and tabulates base_profile::comp per node
    and iupac code; the DP kernel reads the match term from that table."""
    refs, cs, idx, st = world
    if geom:
        util.set_knobs(monkeypatch, geom=geom)
    qs = synth.make_queries(refs, 24, seed=57, window=(0.3, 120), ins=0.02, dele=0.02, lower_rate=0.05, amb_rate=0.02)
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    pl = pipeline.Pipeline(st, famfinder=ff, aligner=dict(al, **{"fs-no-graph": True}))
    pl.run(qs.mask, qs.off, batch=24, inflight=1)
    n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250),
                     al=dict(oal, fs_no_graph=1))
    assert n_dp >= 20
    pl.close()


@pytest.mark.parametrize("geom", ["128,12", "256,12", "128,8", "384,4"])
@pytest.mark.parametrize("insertion", ["forbid", "shift"])
def test_pipeline_forced_multi_strip_geometries(oracle, world, monkeypatch, geom, insertion):
    """The same 24 queries under DP geometries forced with SINA_HIP_TEST="geom=T,B" -- several strips of 12, 8
    and 4 columns per lane, with and without --insertion=forbid (32-bit trace-back cells).  The
    12-column forbid kernels once restored garbage for a scalar load that was spilled while in flight
    (mesh_dp.hip, sload16): 17 or more queries in a launch crashed the GPU."""
    refs, cs, idx, st = world
    util.set_knobs(monkeypatch, geom=geom)
    qs = synth.make_queries(refs, 24, seed=53, window=(0.3, 120), ins=0.02, dele=0.02, lower_rate=0.05)
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    pl = pipeline.Pipeline(st, famfinder=ff, aligner={"insertion": insertion})
    pl.run(qs.mask, qs.off, batch=24, inflight=1)
    n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250),
                     al=dict(insertion=1 if insertion == "forbid" else 0))
    assert n_dp >= 20
    pl.close()


def test_pipeline_calc_idty(oracle, world):
    """--calc-idty (align.cpp:380-382,443-453): align_ident_slv = 100 x the best overlap identity of the
    aligned query with a member of its family (one comparison launch per batch); 100 for copied
    alignments; absent without the option."""
    refs, cs, idx, st = world
    some = synth.make_queries(refs, 20, seed=58, amb_rate=0.01, lower_rate=0.05)
    exact = synth.make_queries(refs, 6, seed=59, sub=0.0, dele=0.0, ins=0.0)
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    for qs in (some, exact):
        pl = pipeline.Pipeline(st, famfinder=ff, aligner={"calc-idty": True})
        pl.run(qs.mask, qs.off, batch=8, inflight=2)
        n = 0
        for qi in range(qs.n):
            want = _oracle_run(oracle, cs, idx, qs, qi, dict(fs_min_len=100, fs_full_len=250))
            got = pl.result(qi)
            if want["status"] in (0, 1):
                assert util.f32_bits(got["idty"]) == util.f32_bits(want["idty"]), qi
                n += 1
        assert n >= 5
        pl.close()
    pl = pipeline.Pipeline(st, famfinder=ff)
    pl.run(some.mask, some.off)
    assert pl.result(0)["idty"] == -1
    pl.close()


def test_pipeline_copy_shortcut_and_realign(oracle, world):
    """Queries that ARE (substrings of) references: alignment is copied (align.cpp:349-388) unless --realign."""
    refs, cs, idx, st = world
    qs = synth.make_queries(refs, 16, seed=54, sub=0.0, dele=0.0, ins=0.0)
    half = synth.make_queries(refs, 16, seed=54, sub=0.0, dele=0.0, ins=0.0, window=(0.25, 150))
    for queries in (qs, half):
        ff = {"fs-min-len": 100, "fs-full-len": 250}
        pl = pipeline.Pipeline(st, famfinder=ff)
        pl.run(queries.mask, queries.off)
        _, n_copy = _check(oracle, refs, queries, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250))
        assert n_copy >= 12
        pl.close()
        pl = pipeline.Pipeline(st, famfinder=ff, aligner={"realign": True})
        pl.run(queries.mask, queries.off)
        n_dp, n_copy = _check(oracle, refs, queries, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250),
                              al=dict(realign=1))
        assert n_copy == 0 and n_dp >= 12
        pl.close()


def test_pipeline_exact_relatives_among_mutated_queries(oracle, world):
    """One batch in which every third query is an exact copy of its reference, every third an exact window of it and
    the rest are mutated (bench.py --exact-rate): copies and DP alignments side by side, short windows that several
    family members hold (the aligner's containment search, host/stages.cpp find_bases) -- all against the oracle."""
    refs, cs, idx, st = world
    full = synth.make_queries(refs, 60, seed=91, sub=[0.0, 0.03, 0.03], dele=[0.0, 0.005, 0.005], ins=[0.0, 0.003, 0.003])
    wins = synth.make_queries(refs, 60, seed=92, sub=[0.03, 0.0, 0.03], dele=[0.005, 0.0, 0.005], ins=[0.003, 0.0, 0.003],
                              window=(0.3, 60))
    ff = {"fs-min-len": 40, "fs-full-len": 250}
    for queries, min_copy in ((full, 18), (wins, 15)):
        pl = pipeline.Pipeline(st, famfinder=ff)
        pl.run(queries.mask, queries.off, batch=32, inflight=2)
        n_dp, n_copy = _check(oracle, refs, queries, pl, cs, idx, ff=dict(fs_min_len=40, fs_full_len=250))
        assert n_copy >= min_copy and n_dp >= 20, (n_dp, n_copy)
        pl.close()


def test_pipeline_family_escalation_and_rejects(oracle, world):
    """Default fs-min-len/full-len reject most of these short references: the candidate list
    escalates 41 -> 410 -> all (famfinder.cpp:591-608) and some queries end with no relatives."""
    refs, cs, idx, st = world
    qs = synth.make_queries(refs, 12, seed=55)
    pl = pipeline.Pipeline(st, famfinder={"fs-full-len": 318, "fs-min-len": 150})
    pl.run(qs.mask, qs.off)
    _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_full_len=318, fs_min_len=150))
    pl.close()
    pl = pipeline.Pipeline(st)   # SINA defaults: nothing is "full length" (>=1400) here
    pl.run(qs.mask, qs.off)
    _check(oracle, refs, qs, pl, cs, idx)
    pl.close()


@pytest.mark.parametrize("geom,insertion", [(None, "shift"), ("128,12", "shift"), ("128,12", "forbid"),
                                            ("128,8", "forbid"), ("384,4", "shift")])
def test_pipeline_weighted_scheme(oracle, world, monkeypatch, geom, insertion):
    """scoring_scheme_weighted (positional-variability weights, align.cpp:410-414) through the pipeline,
    also under forced multi-strip geometries and combined with --insertion=forbid: 32 queries, enough
    to fill more than one round of the 12-column kernels' scalar-register-starved variants."""
    refs, cs, idx, st = world
    if geom:
        util.set_knobs(monkeypatch, geom=geom)
    rng = np.random.default_rng(8)
    w = rng.uniform(0.3, 1.4, size=refs.width).astype(np.float32)
    st.add_filter("posvar", w)
    qs = synth.make_queries(refs, 32, seed=56)
    ff = {"fs-min-len": 100, "fs-full-len": 250, "filter": "posvar"}
    pl = pipeline.Pipeline(st, famfinder=ff, aligner={"insertion": insertion})
    pl.run(qs.mask, qs.off)
    n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250),
                     al=dict(weights=w, insertion=1 if insertion == "forbid" else 0))
    assert n_dp >= 28
    pl.close()


@pytest.mark.parametrize("fs_min,fs_max", [(100, 128), (5, 5), (1, 1)])
def test_pipeline_family_sizes(oracle, world, fs_min, fs_max):
    """Families of 128 members (the device DAG build's limit: wide LDS tables, long predecessor
    lists), of five and of a single reference -- a DAG that is a chain."""
    refs, cs, idx, st = world
    qs = synth.make_queries(refs, 16, seed=58, ins=0.01, dele=0.01)
    ff = {"fs-min-len": 100, "fs-full-len": 250, "fs-min": fs_min, "fs-max": fs_max}
    pl = pipeline.Pipeline(st, famfinder=ff)
    pl.run(qs.mask, qs.off)
    n_dp, _ = _check(oracle, refs, qs, pl, cs, idx,
                     ff=dict(fs_min_len=100, fs_full_len=250, fs_min=fs_min, fs_max=fs_max))
    assert n_dp >= 12
    pl.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SINA_FUZZ_SEEDS", "12"))))
def test_pipeline_option_fuzz(oracle, world, monkeypatch, seed):
    """Seeded random combinations of the aligner's options, the scoring parameters (incl. gap extension
    above gap opening: the general chain path of the kernel), the weighted scheme, the host / device DAG
    build, the DP geometry, the walk kernel and the query shape -- every case against the oracle, everything compared."""
    refs, cs, idx, st = world
    rng = np.random.default_rng(1000 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]  # noqa: E731
    geom = pick([None, None, "64,8", "128,8", "128,12", "192,4", "256,4"])
    if geom:
        util.set_knobs(monkeypatch, geom=geom)
    # (the cell walk: one wave per query, or one lane per query as launches of 8192 queries and more do it)
    util.set_knobs(monkeypatch, bt_lanes=seed % 2)
    overhang, lowercase, insertion = pick(["attach", "remove", "edge"]), pick(["none", "original", "unaligned"]), pick(["shift", "forbid"])
    ms, mms = float(pick([2, 3, 1.5])), float(pick([-1, -2, -0.5]))
    gp, gpe = pick([(5, 2), (4, 1.5), (2, 3), (3, 3), (6, 0.5)])
    fsw = float(pick([1.0, 0.0, 0.5, 2.5]))
    weighted = bool(rng.integers(0, 2))
    al = {"overhang": overhang, "lowercase": lowercase, "insertion": insertion, "match-score": ms, "mismatch-score": mms,
          "pen-gap": gp, "pen-gapext": gpe, "fs-weight": fsw, "device-graph": bool(rng.integers(0, 2))}
    oal = dict(overhang={"attach": 0, "remove": 1, "edge": 2}[overhang], lowercase={"none": 0, "original": 1, "unaligned": 2}[lowercase],
               insertion=1 if insertion == "forbid" else 0, match_score=ms, mismatch_score=mms, gap_penalty=gp,
               gap_ext_penalty=gpe, fs_weight=fsw)
    if seed % 4 == 3:  # every fourth case: the family as a profile (positional weights then play no part)
        al["fs-no-graph"] = True
        oal["fs_no_graph"] = 1
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    if weighted:
        w = rng.uniform(0.3, 1.4, size=refs.width).astype(np.float32)
        st.add_filter("fuzzvar", w)
        ff["filter"] = "fuzzvar"
        oal["weights"] = w
    window = pick([None, (0.3, 120), (0.1, 200)])
    qs = synth.make_queries(refs, 20, seed=2000 + seed, window=window, ins=float(pick([0.003, 0.02])),
                            dele=float(pick([0.005, 0.02])), lower_rate=0.05, amb_rate=0.01)
    pl = pipeline.Pipeline(st, famfinder=ff, aligner=al)
    pl.run(qs.mask, qs.off, batch=int(pick([20, 7])), inflight=int(pick([1, 2])))
    n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250), al=oal)
    assert n_dp >= 15, (geom, al)
    pl.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SINA_FUZZ_SEEDS", "12"))))
def test_famfinder_option_fuzz(oracle, world, seed):
    """Seeded random famfinder options (famfinder.cpp:497-612: family size bounds, score and identity
    cut-offs, full-length / gap / coverage requirements, leave-query-out) with whole or partial queries,
    exact copies of references among them: family, alignment and log against the oracle."""
    refs, cs, idx, st = world
    rng = np.random.default_rng(5000 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]  # noqa: E731
    fs_min = int(pick([40, 10, 3, 60]))
    o = dict(fs_min=fs_min, fs_max=int(pick([fs_min, fs_min + 5, 2 * fs_min])), fs_msc=float(pick([0.7, 0.2, 50.0, 150.0])),
             fs_msc_max=float(pick([2.0, 2.0, 0.98, 0.9])), fs_leave_query_out=int(pick([0, 0, 1])),
             fs_req=int(pick([1, 1, 5, 30])), fs_req_full=int(pick([1, 0, 3])), fs_full_len=int(pick([250, 290, 310])),
             fs_req_gaps=int(pick([10, 0, 400])), fs_min_len=int(pick([100, 150, 280])), fs_cover_gene=int(pick([0, 0, 2])))
    ff = {k.replace("_", "-"): v for k, v in o.items()}
    window = pick([None, None, (0.2, 150)])
    qs = synth.make_queries(refs, 16, seed=6000 + seed, window=window, sub=float(pick([0.03, 0.0, 0.1])),
                            dele=float(pick([0.005, 0.0])), ins=float(pick([0.003, 0.0])))
    pl = pipeline.Pipeline(st, famfinder=ff)
    pl.run(qs.mask, qs.off, batch=int(pick([16, 5])), inflight=int(pick([1, 2])))
    _check(oracle, refs, qs, pl, cs, idx, ff=o)
    pl.close()


@pytest.mark.parametrize("geom", [None, "128,4", "64,8", "128,12"])
@pytest.mark.parametrize("insertion", ["shift", "forbid"])
def test_pipeline_mixed_query_lengths(oracle, world, monkeypatch, geom, insertion):
    """Queries of 40 to 310 bases in ONE launch: the geometry follows the longest, every query sweeps
    only the strips up to its own last column (mesh_dp_kernel: S per query)."""
    refs, cs, idx, st = world
    if geom:
        util.set_knobs(monkeypatch, geom=geom)
    parts = [synth.make_queries(refs, 6, seed=70 + i, window=w, ins=0.01, dele=0.01)
             for i, w in enumerate([(0.4, 40), (0.2, 130), (0.1, 260), None])]
    masks = [p.seq(i) for p in parts for i in range(p.n)]
    order = np.random.default_rng(77).permutation(len(masks))
    masks = [masks[i] for i in order]
    off = np.zeros(len(masks) + 1, np.int64)
    off[1:] = np.cumsum([len(m) for m in masks])
    qs = synth.QuerySet(mask=np.concatenate(masks), off=off, src=np.zeros(len(masks), np.int64))
    pl = pipeline.Pipeline(st, famfinder={"fs-min-len": 100, "fs-full-len": 250}, aligner={"insertion": insertion})
    pl.run(qs.mask, qs.off, batch=qs.n, inflight=1)
    n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250),
                     al=dict(insertion=1 if insertion == "forbid" else 0))
    assert n_dp >= 20
    pl.close()


def test_driver_worker_handles_many_batches(oracle, world):
    """One worker of the host driver taking batch after batch (20 queries in batches of 3, one in
    flight): every result -- log text and status included -- is that query's alone.  (The driver reuses
    its tray objects; their logs once carried over from the previous batch.)"""
    refs, cs, idx, st = world
    qs = synth.make_queries(refs, 20, seed=61, ins=0.02, dele=0.01)
    pl = pipeline.Pipeline(st, famfinder={"fs-min-len": 100, "fs-full-len": 250})
    pl.run(qs.mask, qs.off, batch=3, inflight=1)
    n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250))
    assert n_dp >= 16
    pl.close()


def test_single_tray_batching_shim(oracle, world):
    """batch=1: every tray goes through the stages alone, as SINA's TBB nodes would call them."""
    refs, cs, idx, st = world
    qs = synth.make_queries(refs, 6, seed=57)
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    pl = pipeline.Pipeline(st, famfinder=ff)
    pl.run(qs.mask, qs.off, batch=1, inflight=3)
    _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250))
    pl.close()


@pytest.mark.parametrize("al,oal", [({}, {}), ({"lowercase": "original", "overhang": "edge"}, dict(lowercase=1, overhang=2)),
                                    ({"device-graph": False, "insertion": "forbid"}, dict(insertion=1))])
def test_repeated_queries_go_to_the_device_once(oracle, world, al, oal):
    """Batch-level memoisation (the device-side analogue of kmer_search's cache of the base strings it has just
    seen, src/kmer_search.cpp:105,377-378,419): 60 queries of which only 24 are distinct -- same bases, same case.
    The trays equal the oracle's and those of a run with the memoisation switched off, field by field, and the
    device has searched and aligned 24 where the other run searched and aligned 60."""
    refs, cs, idx, st = world
    base = synth.make_queries(refs, 24, seed=63, ins=0.01, dele=0.01, lower_rate=0.05, window=(0.2, 200))
    rng = np.random.default_rng(64)
    pick = np.concatenate([np.arange(24), rng.integers(0, 24, size=36)])
    rng.shuffle(pick)
    qs = synth.pick_queries(base, pick)
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    runs = {}
    for dedup in (True, False):
        pl = pipeline.Pipeline(st, famfinder=ff, aligner=al, dedup=dedup)
        s0 = st.stats()
        pl.run(qs.mask, qs.off, batch=60, inflight=1)
        s1 = st.stats()
        runs[dedup] = ([pl.result(q) for q in range(qs.n)], s1["dp_cells"] - s0["dp_cells"], s1["postings"] - s0["postings"])
        n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250), al=oal)
        assert n_dp >= 50
        pl.close()
    for a, b in zip(runs[True][0], runs[False][0]):
        assert a["status"] == b["status"] and a["family"] == b["family"] and a["log"] == b["log"]
        assert (a["packed"] == b["packed"]).all() and (a["head"], a["tail"], a["qual"]) == (b["head"], b["tail"], b["qual"])
    # the device's share: the 24 distinct queries' cells and postings -- what a run over just those costs
    pl = pipeline.Pipeline(st, famfinder=ff, aligner=al, dedup=False)
    s0 = st.stats()
    pl.run(base.mask, base.off, batch=24, inflight=1)
    s1 = st.stats()
    pl.close()
    assert runs[True][1] == s1["dp_cells"] - s0["dp_cells"] and runs[True][2] == s1["postings"] - s0["postings"]
    assert runs[False][1] > 2 * runs[True][1] and runs[False][2] > 2 * runs[True][2]


def test_batched_stage_shim_under_32_concurrent_single_tray_callers(oracle, world):
    """INTEGRATION.md section 1 as it is bound: sina::batched<famfinder> -> batched<aligner>, called with ONE
    tray per call from 40 threads at once (SINA's unlimited-concurrency function_nodes, src/sina.cpp:497-519).
    Every caller gets its own tray back, equal to the oracle's, whatever batches the shim happened to form."""
    refs, cs, idx, st = world
    qs = synth.make_queries(refs, 150, seed=58, ins=0.01, dele=0.01, lower_rate=0.03)
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    pl = pipeline.Pipeline(st, famfinder=ff)
    failed, err = pl.run_single_trays(qs.mask, qs.off, threads=40, max_batch=64, linger_us=500)
    assert not failed.any() and err == ""
    n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250))
    assert n_dp >= 130
    # ... a second run on the same pipeline object with other queries, small batches (max_batch 4: full batches
    # go at once, the rest after the linger), and nothing of the first run's results shows through
    qs2 = synth.make_queries(refs, 37, seed=59)
    failed, err = pl.run_single_trays(qs2.mask, qs2.off, threads=33, max_batch=4, linger_us=200)
    assert not failed.any()
    _check(oracle, refs, qs2, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250))
    with pytest.raises(pipeline.HostError):          # (beyond this run's 37 results: nothing of the first run's)
        pl.result(40)
    pl.close()


def test_batched_stage_shim_hands_a_stage_exception_to_the_callers_of_that_batch(oracle, world):
    """One tray reaches the aligner with a family member that is not of the reference store (a stale pointer):
    the device call rejects its batch, the stage throws std::runtime_error, and the shim hands that exception to
    every caller whose tray travelled in the batch -- nobody hangs, nobody gets another caller's tray, and the
    trays of all other batches are the oracle's."""
    refs, cs, idx, st = world
    qs = synth.make_queries(refs, 96, seed=60)
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    pl = pipeline.Pipeline(st, famfinder=ff)
    wants = [_oracle_run(oracle, cs, idx, qs, qi, dict(fs_min_len=100, fs_full_len=250)) for qi in range(qs.n)]
    poison = next(qi for qi in range(40, qs.n) if wants[qi]["status"] == 0)   # (one that goes through the DP)
    failed, err = pl.run_single_trays(qs.mask, qs.off, threads=32, max_batch=8, linger_us=300, poison=poison)
    assert failed[poison] and "reference id out of range" in err
    assert 1 <= failed.sum() <= 8                      # its batch, no more than max_batch callers
    ok = 0
    for qi in range(qs.n):
        got, want = pl.result(qi), wants[qi]
        if failed[qi]:
            assert got["status"] == 2 and got["log"] == ""
            continue
        assert got["status"] == want["status"]
        if want["status"] != 2:
            assert (got["packed"] == want["packed"]).all()
            if want["status"] == 0:
                assert got["log"] == want["log"]
            ok += 1
    assert ok >= 80
    # the pipeline (store, contexts, shim) is as good as new afterwards
    failed, err = pl.run_single_trays(qs.mask, qs.off, threads=32, max_batch=64, linger_us=300)
    assert not failed.any()
    _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250))
    pl.close()


def test_batched_stage_shim_serves_a_lone_caller_after_the_linger(oracle, world):
    """A single caller is not kept waiting for a batch to fill: it is served one linger period after it arrived
    (twice per tray: famfinder and aligner)."""
    import time
    refs, cs, idx, st = world
    qs = synth.make_queries(refs, 3, seed=62)
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    pl = pipeline.Pipeline(st, famfinder=ff)
    pl.run_single_trays(qs.mask, qs.off, threads=1, max_batch=1024, linger_us=100)       # (warm: contexts, scratch)
    t0 = time.time()
    failed, _ = pl.run_single_trays(qs.mask, qs.off, threads=1, max_batch=1024, linger_us=100000)
    dt = time.time() - t0
    assert not failed.any()
    assert 3 * 2 * 0.1 <= dt < 3 * 2 * 0.1 + 1.5, dt
    _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250))
    pl.close()


@pytest.mark.parametrize("shape", ["v4_amplicon", "lsu_23s"])
def test_baseline_config_shapes(oracle, shape):
    """BASELINE.json configs[2] (250 bp V4 amplicons vs full-length 16S references) and configs[4]
    (23S-like ~3000 bp, width 150k): the short-query and the long-query / wide-graph DP shapes."""
    if shape == "v4_amplicon":
        refs = synth.make_refs(300, length=1500, width=50000, seed=81)
        qs = synth.make_queries(refs, 32, seed=82, window=(1.0 / 3.0, 250))
    else:
        refs = synth.make_refs(120, length=3000, width=150000, seed=83)
        qs = synth.make_queries(refs, 32, seed=84)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    st = pipeline.Store(":mem:gpu-shape-" + shape, refs)
    try:
        pl = pipeline.Pipeline(st)
        pl.run(qs.mask, qs.off)
        n_dp, _ = _check(oracle, refs, qs, pl, cs, idx)
        assert n_dp == qs.n
        pl.close()
    finally:
        st.close()


def test_overlong_query_fails_alone(oracle, world):
    """A sequence beyond the device limit (SINA_HIP_MAX_QUERY_LEN = 10240 bases) is a soft
    failure of THAT tray -- reason in the log, no aligned sequence -- and the rest of its batch is
    aligned as usual (the reference aligns any length; this engine does not abort the run)."""
    refs, cs, idx, st = world
    qs = synth.make_queries(refs, 6, seed=61)
    rng = np.random.default_rng(62)
    long_q = synth.CODE_TO_MASK[rng.integers(0, 4, size=11000)]
    masks = [qs.seq(i) for i in range(3)] + [long_q] + [qs.seq(i) for i in range(3, 6)]
    off = np.zeros(len(masks) + 1, np.int64)
    off[1:] = np.cumsum([len(m) for m in masks])
    ff = {"fs-min-len": 100, "fs-full-len": 250}
    pl = pipeline.Pipeline(st, famfinder=ff)
    pl.run(np.concatenate(masks), off, batch=7, inflight=1)
    got = [pl.result(i) for i in range(7)]
    assert got[3]["status"] == 2 and "longer than 10240 bases" in got[3]["log"]
    sub = synth.QuerySet(mask=qs.mask, off=qs.off, src=qs.src)
    for k, qi in enumerate([0, 1, 2, None, 3, 4, 5]):
        if qi is None:
            continue
        want = _oracle_run(oracle, cs, idx, sub, qi, dict(fs_min_len=100, fs_full_len=250))
        assert got[k]["status"] == want["status"]
        if want["status"] != 2:
            assert (got[k]["packed"] == want["packed"]).all()
    pl.close()


def test_pipeline_9000_base_queries_and_families_of_200(oracle):
    """What round 2 failed softly: queries longer than 8191 bases (18 strips of 512 columns; the k-mer
    kernel's cursor list grows with the query) and families larger than the DAG-build kernel's 128
    (--fs-max 200: the host builds those DAGs, the device aligns them) -- whole pipeline against the
    oracle, with and without --insertion=forbid (32-bit cells: the widened value_sidx field)."""
    refs = synth.make_refs(260, length=9400, width=30000, seed=771, n_clades=2)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    st = pipeline.Store(":mem:gpu-long", refs)
    try:
        qs = synth.make_queries(refs, 3, seed=772)
        assert max(len(qs.seq(i)) for i in range(qs.n)) > 9000
        for al, oal in (({}, {}), ({"insertion": "forbid"}, dict(insertion=1))):
            ff = {"fs-min-len": 100, "fs-full-len": 250}
            pl = pipeline.Pipeline(st, famfinder=ff, aligner=al)
            pl.run(qs.mask, qs.off, batch=3, inflight=1)
            n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250), al=oal)
            assert n_dp == 3
            pl.close()
    finally:
        st.close()
    # families of 200
    refs = synth.make_refs(400, length=300, width=2400, seed=773, n_clades=1)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    st = pipeline.Store(":mem:gpu-bigfam", refs)
    try:
        qs = synth.make_queries(refs, 12, seed=774)
        ff = {"fs-min-len": 100, "fs-full-len": 250, "fs-min": 200, "fs-max": 200}
        pl = pipeline.Pipeline(st, famfinder=ff)
        pl.run(qs.mask, qs.off, batch=12, inflight=1)
        n_dp, _ = _check(oracle, refs, qs, pl, cs, idx, ff=dict(fs_min_len=100, fs_full_len=250, fs_min=200, fs_max=200))
        assert n_dp >= 10
        assert all(pl.result(i)["family"].count(":") >= 190 for i in range(qs.n) if pl.result(i)["status"] == 0)
        pl.close()
    finally:
        st.close()
