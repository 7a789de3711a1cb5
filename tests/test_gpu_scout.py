"""GPU tests of the scout pass (scout.hip): the bound U every query's certified row skip starts
from is the cost of a real path -- the query's banded alignment against the chain of its family's first member.
Results never depend on it -- the
skipping kernel certifies against whatever U it is given -- so: (a) with the scout on, off, forced too bold and
forced too loose the trays are the oracle's; (b) on bench-shaped queries the scout's value lets (nearly) every query
pass its first certificate and sweeps fewer rows than the store-wide guess; (c) a launch that mixes near-identical
and distant queries is as exact and needs no more second attempts than a homogeneous one."""
import numpy as np
import pytest

from sina_amd import pipeline, synth
from tests import util

pytestmark = pytest.mark.gpu


def _run_and_check(oracle, refs, qs, cs, idx, key, batch):
    st = pipeline.Store(key, refs)
    try:
        st.build_index(10, False)
        pl = pipeline.Pipeline(st)
        pl.run(qs.mask, qs.off, batch=batch, inflight=1)
        n_dp = 0
        for qi in range(qs.n):
            q = util.query_cseq(qs, qi, upper=False)
            ids, sc, fflog = idx.famfinder(q, oracle.ff_opts())
            want = oracle.align([cs[i] for i in ids], q, oracle.align_opts())
            got = pl.result(qi)
            assert got["status"] == want["status"], (qi, got["log"], want["log"])
            assert (got["packed"] == want["packed"]).all(), qi
            assert (got["head"], got["tail"], got["qual"]) == (want["head"], want["tail"], want["qual"])
            if want["status"] == 0:
                assert got["log"] == fflog + want["log"]
                n_dp += 1
        s = st.stats()
        pl.close()
        return s, n_dp
    finally:
        st.close()


@pytest.fixture(scope="module")
def world(oracle):
    refs = synth.make_refs(3000, length=1500, width=50000, seed=61)
    cs = util.cseqs_from_refs(refs)
    return refs, cs, oracle.Index(cs, k=10)


@pytest.mark.parametrize("mode", ["on", "off", "bold", "loose", "absurd"])
def test_scout_pipeline_equals_oracle(oracle, world, monkeypatch, mode):
    """Full-length 16S queries end to end with the scout left alone, switched off (the store's guess), and forced
    wrong: 300 units too bold (every certificate fails: second attempts under what the first found), 400 units too
    loose (a wide band), and absurd (-1e5: beyond the exact range, the guess stands)."""
    refs, cs, idx = world
    knobs = {"on": {}, "off": {"scout": "0"}, "bold": {"scout_add": "-300"}, "loose": {"scout_add": "400"},
             "absurd": {"scout_add": "-200000"}}[mode]
    if knobs:
        util.set_knobs(monkeypatch, **knobs)
    qs = synth.make_queries(refs, 24, seed=62)
    s, n_dp = _run_and_check(oracle, refs, qs, cs, idx, ":mem:scout_%s" % mode, 24)
    assert n_dp >= 20
    assert s["dp_queries_pruned"] >= n_dp
    if mode == "off":
        assert s["scout_launches"] == 0
    else:
        assert s["scout_launches"] >= 1
    if mode == "on":       # the scout's path costs a little more than the optimum: nobody sweeps twice, few rows are swept
        assert s["dp_second_attempts"] + s["dp_full_sweeps"] <= 1
        assert s["dp_rows_swept"] < 0.5 * s["dp_rows"]
    if mode == "bold":
        assert s["dp_second_attempts"] >= n_dp - 1
        assert s["dp_full_sweeps"] == 0


def test_scout_rows_swept_beat_the_guess(oracle, world, monkeypatch):
    """Two launches of the same queries: the scout's bounds sweep fewer rows than the store-wide guess."""
    refs, cs, idx = world
    qs = synth.make_queries(refs, 48, seed=63)
    swept = {}
    for mode in ("off", "on"):
        util.set_knobs(monkeypatch, scout="0" if mode == "off" else None)
        s, n_dp = _run_and_check(oracle, refs, qs, cs, idx, ":mem:scout_cmp_%s" % mode, 48)
        swept[mode] = s["dp_rows_swept"] / max(1, s["dp_rows"])
    assert swept["on"] < swept["off"], swept


def test_scout_mixed_divergence_in_one_launch(oracle, world):
    """Queries at 0.5 / 3 / 10 / 20 % substitutions (indels in proportion) within ONE launch: every tray is the
    oracle's and nobody is swept in full.  The store's guess guards the scout's values; where it is too bold for a
    distant query the first attempt dies early and the second runs under the scout's value -- until the store has
    seen such queries: a second launch of the same mix needs (next to) no second attempts."""
    refs, cs, idx = world
    qs = synth.make_queries(refs, 48, seed=64, sub=[0.005, 0.03, 0.10, 0.20], dele=[0.001, 0.005, 0.015, 0.03],
                            ins=[0.001, 0.003, 0.01, 0.02])
    st = pipeline.Store(":mem:scout_mix", refs)
    try:
        st.build_index(10, False)
        pl = pipeline.Pipeline(st)
        seen = None
        for launch in range(2):
            pl.run(qs.mask, qs.off, batch=48, inflight=1)
            n_dp = 0
            for qi in range(qs.n):
                q = util.query_cseq(qs, qi, upper=False)
                ids, sc, fflog = idx.famfinder(q, oracle.ff_opts())
                want = oracle.align([cs[i] for i in ids], q, oracle.align_opts())
                got = pl.result(qi)
                assert got["status"] == want["status"], (qi, got["log"], want["log"])
                assert (got["packed"] == want["packed"]).all(), qi
                n_dp += want["status"] == 0
            s = st.stats()
            assert n_dp >= 36
            assert s["dp_full_sweeps"] == 0
            if seen is not None:
                assert s["dp_second_attempts"] - seen <= 2, (s["dp_second_attempts"], seen)
            seen = s["dp_second_attempts"]
        pl.close()
    finally:
        st.close()
