"""GPU parity of the DP kernel's certified row skip (mesh_dp.hip PRUNE): whatever bound a launch
guesses, (a) every cell whose reference value is at or below the bound the kernel ended up using is the reference's
bit for bit -- value, value_midx, value_sidx -- and every other cell is above its bound, (b) the finished alignments
are the oracle's, (c) a guess that is too bold is caught by the certificate and the query is swept again."""
import os

import numpy as np
import pytest

from sina_amd import capi, pipeline, synth
from tests import util

pytestmark = pytest.mark.gpu


def _bound_plane(info, rg_cols, N, L):
    """T(m, s) = U + min(a * r, R(m) - gmin * max(0, C(m) - r)), r = L-1-s, in units of 1/64, as float64 (common.h,
    "the bound")."""
    rg, cols = rg_cols
    a, gmin = float(info["prune_step"]), float(info["prune_gmin"])
    r = (L - 1 - np.arange(L, dtype=np.float64))[None, :]
    second = rg.astype(np.float64)[:, None] - gmin * np.maximum(0.0, cols.astype(np.float64)[:, None] - r)
    return info["ubound"] + np.minimum(a * r, second) / 64.0


def _check_planes(ctx, cells, vm, vs, val):
    N, L = val.shape
    info = ctx.dp_info(0)
    assert info["attempts"] >= 1, "the launch did not go through the skipping kernel"
    if np.isinf(info["ubound"]):     # third attempt: swept in full
        alive = np.ones((N, L), bool)
    else:
        T = _bound_plane(info, ctx.rgain(N), N, L)
        alive = cells["value"].astype(np.float64) <= T
        assert (val[~alive].astype(np.float64) > T[~alive]).all(), "a cell the reference has above its bound came out at or below it"
    assert (util.f32_bits(val)[alive] == util.f32_bits(cells["value"])[alive]).all()
    assert (vm[alive] == cells["value_midx"][alive]).all()
    assert (vs[alive] == cells["value_sidx"][alive]).all()
    return info, alive


SIMPLE_CASES = [i for i, c in enumerate(util.MESH_CASES)
                if not c["scheme"]["weighted"] and not c["scheme"]["forbid"] and c["scheme"]["gap"] >= c["scheme"]["gapext"]]


@pytest.mark.parametrize("rho", [None, "0.3", "0.9", "0.98", "2"])
def test_row_skip_planes_equal_oracle_at_or_below_the_bound(oracle, gpu_ctx, monkeypatch, rho):
    """The reference-parts mesh cases of the simple scheme (families of 1..41, fs-weight 0 / 1 / 2.5, one full
    16S family) under forced multi-strip geometries, with the launch's guess left to the library (None) or
    forced: timid (0.3), plausible (0.9), bold (0.98) and impossible (2: no path gains twice the bound -- every
    query fails its first certificate and is swept again under the bound the first attempt found)."""
    if rho is not None:
        util.set_knobs(monkeypatch, rho=rho)
    n_skipping = n_second = 0
    for ci in SIMPLE_CASES:
        case = util.MESH_CASES[ci]
        fam, qa, width, w, sch = util.mesh_case_inputs(case)
        cs = [oracle.Cseq.from_packed("f%d" % i, a, width) for i, a in enumerate(fam)]
        q = oracle.Cseq.from_packed("q", qa, len(qa))
        opts = oracle.align_opts(match_score=sch["match"], mismatch_score=sch["mismatch"], gap_penalty=sch["gap"],
                                 gap_ext_penalty=sch["gapext"], fs_weight=sch["fs_weight"])
        cells = oracle.mesh_compute(cs, q, opts, weight=sch["fs_weight"])
        g = util.graph_dict(cs, sch["fs_weight"])
        L = len(qa)
        geoms = [None] if L > 1000 else ["%d,4" % (64 * ((L + 255) // 256 + (ci % 2))), "%d,8" % (64 * ((L + 511) // 512 + 1))]
        for geom in geoms:
            if geom:
                util.set_knobs(monkeypatch, geom=geom)
            else:
                util.set_knobs(monkeypatch, geom=None)
            if geom and int(geom.split(",")[0]) < 128:
                continue   # (a single strip: the kernel does not skip)
            gb = gpu_ctx.graph_batch([g], width)
            p = gpu_ctx.params(match_score=sch["match"], mismatch_score=sch["mismatch"], gap_penalty=sch["gap"],
                               gap_ext_penalty=sch["gapext"], fs_weight=sch["fs_weight"])
            vm, vs, val = gpu_ctx.debug_mesh(gb, (qa >> 24).astype(np.uint8), p, prune=True)
            info, alive = _check_planes(gpu_ctx, cells, vm, vs, val)
            n_second += info["attempts"] >= 2
            strips = (L - 1) // (64 * int((geom or "192,8").split(",")[1])) + 1
            n_skipping += info["rows_swept"] < strips * g["n"] * info["attempts"]
            # the end cell and its value are the oracle's
            want = oracle.align(cs, q, oracle.align_opts(match_score=sch["match"], mismatch_score=sch["mismatch"],
                                                         gap_penalty=sch["gap"], gap_ext_penalty=sch["gapext"],
                                                         fs_weight=sch["fs_weight"], realign=1))
            if want["status"] == 0:
                assert info["status"] == 0
    if rho == "2":
        assert n_second >= len(SIMPLE_CASES)        # every launch was caught by its certificate
    if rho in (None, "0.9"):
        assert n_skipping >= 3                        # ... and rows are actually skipped


@pytest.mark.parametrize("rho", [None, "0.5", "0.97", "3"])
def test_row_skip_pipeline_equals_oracle(oracle, monkeypatch, rho):
    """Full-length 16S queries end to end (device DAG build with its column bound, three-strip DP, walk, assembly)
    with the guess left alone or forced: family, alignment, head / tail / quality, log text equal the oracle's;
    rows are skipped; a bold guess costs second attempts, never a different result."""
    if rho is not None:
        util.set_knobs(monkeypatch, rho=rho)
    refs = synth.make_refs(3000, length=1500, width=50000, seed=52)
    qs = synth.make_queries(refs, 24, seed=53)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    st = pipeline.Store(":mem:prune%s" % rho, refs)
    try:
        st.build_index(10, False)
        pl = pipeline.Pipeline(st)
        pl.run(qs.mask, qs.off, batch=24, inflight=1)
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
        assert n_dp >= 20
        s = st.stats()
        assert s["dp_queries_pruned"] >= n_dp
        assert s["dp_rows_swept"] > 0 and s["dp_cells_swept"] > 0
        if rho == "3":
            assert s["dp_second_attempts"] >= n_dp
        elif rho != "0.97":     # (bold: many of these queries pay a second attempt)
            assert s["dp_rows_swept"] < 0.75 * s["dp_rows"]
        if rho is None:
            assert s["dp_second_attempts"] + s["dp_full_sweeps"] <= n_dp // 4
        pl.close()
    finally:
        st.close()


def test_device_column_bound_covers_every_path(oracle, gpu_ctx):
    """The bound R(m) the device DAG build emits: for every edge p -> m of the DAG, R(p) >= gain(m) + R(m), gain(m)
    at least 64 * 2 * weight(m) + 1 units -- what the skip's induction needs (common.h)."""
    refs = synth.make_refs(300, length=600, width=6000, seed=91, long_del_prob=0.4)
    ctx = capi.Context(0)
    try:
        ctx.upload_refs(refs.ab, refs.off, refs.width)
        rng = np.random.default_rng(92)
        for F in (1, 7, 40):
            ids = rng.choice(refs.n, size=F, replace=False).astype(np.uint32)
            g = ctx.debug_family_graph(ids, fs_weight=1.0)
            rg, cols = ctx.rgain(g["n"])
            rg, cols = rg.astype(np.int64), cols.astype(np.int64)
            gain = np.ceil(64.0 * 2.0 * g["weight"].astype(np.float64)) + 1
            for m in range(g["n"]):
                for p in g["pred"][g["pred_off"][m]:g["pred_off"][m + 1]]:
                    assert rg[p] >= gain[m] + rg[m], (F, m, p)
            assert (np.diff(rg) <= 0).all()      # ids ascend with the column
            assert rg[-1] == 0
            # C(m) = occupied columns right of the node's
            ucols = np.unique(g["pos"])
            assert (cols == len(ucols) - 1 - np.searchsorted(ucols, g["pos"])).all()
    finally:
        ctx.close()


@pytest.mark.parametrize("rho", [None, "0.6", "0.95", "2"])
def test_row_skip_narrow_band_after_wide_ones(oracle, gpu_ctx, monkeypatch, rho):
    """A query that IS a member of its family (the optimum runs along one reference: the narrowest band there is,
    strips cut early, later strips sweeping rows the strip to their left never reached) right after queries with
    wide bands on the same context -- whatever earlier sweeps left in the edge records, spill rows and row slots
    must not be read as this sweep's.  Planes against the oracle's where at or below the bound, alignments through
    sina_hip_align_graphs against the oracle's."""
    if rho is not None:
        util.set_knobs(monkeypatch, rho=rho)
    refs = synth.make_refs(400, length=1500, width=50000, seed=61, long_del_prob=0.5)
    cs = util.cseqs_from_refs(refs)
    rng = np.random.default_rng(62)
    qs = synth.make_queries(refs, 6, seed=63, sub=0.08, dele=0.02, ins=0.02)
    fam_ids = rng.choice(refs.n, size=40, replace=False)
    fam = [cs[i] for i in fam_ids]
    g = util.graph_dict(fam)
    gb = gpu_ctx.graph_batch([g], refs.width)
    cases = [util.query_cseq(qs, i) for i in range(3)]                       # divergent queries: wide bands
    member = cs[int(fam_ids[7])]
    mm = ((refs.seq(int(fam_ids[7])) >> 24) & 0x0f).astype(np.uint8)
    exact = oracle.Cseq.from_packed("member", np.arange(len(mm), dtype=np.uint32) | (mm.astype(np.uint32) << 24), len(mm))
    cases += [exact, util.query_cseq(qs, 3), exact]
    for q in cases:
        qm = (q.packed() >> 24).astype(np.uint8)
        cells = oracle.mesh_compute(fam, q)
        vm, vs, val = gpu_ctx.debug_mesh(gb, qm, gpu_ctx.params(), prune=True)
        _check_planes(gpu_ctx, cells, vm, vs, val)
        out, pos = gpu_ctx.align_graphs(gb, qm, np.array([0, len(qm)], np.uint64), gpu_ctx.params())
        ref = oracle.align(fam, q, oracle.align_opts(realign=1)) if q is not exact else None
        assert out[0]["status"] == 0
        if ref is not None and ref["status"] == 0:
            aligned, _ = util.finish_alignment(qm, out[0], pos[:len(qm)], refs.width)
            assert aligned == ref["aligned"]
        else:   # (the aligner would copy the member's alignment: compare the walk with the oracle's planes instead)
            end_v = np.float32(out[0]["raw"])
            snk = g["snk"]
            best = min(cells["value"][:, -1].min(), cells["value"][snk].min())
            assert util.f32_bits(np.float32(best)) == util.f32_bits(end_v)
            got_cols = refs.width - 1 - pos[:len(qm)][::-1]
            own = (refs.seq(int(fam_ids[7])) & 0xFFFFFF).astype(np.uint32)
            assert (got_cols == own).mean() > 0.97


@pytest.mark.parametrize("seed", range(int(os.environ.get("SINA_FUZZ_SEEDS", "12"))))
def test_row_skip_plane_fuzz(oracle, monkeypatch, seed):
    """The mesh plane fuzz of test_gpu_edges.py with the row skip in play: seeded random families (1 - 60 members,
    divergence, long deletions, ambiguity codes, lower case), scoring parameters, fs-weight, LDS budget, a forced
    multi-strip geometry and a random guess (left alone, timid, bold, impossible).  Where the launch went through
    the skipping kernel: every cell at or below its bound is the oracle's, every other one above it; where it did
    not (insertion=forbid, extend > open ...): the planes are the oracle's cell for cell."""
    rng = np.random.default_rng(7700 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]  # noqa: E731
    for attempt in range(8):
        length = int(pick([150, 320, 700]))
        refs = synth.make_refs(int(pick([8, 40, 90])), length=length, width=int(length * pick([3, 8])), seed=7800 + seed + 1000 * attempt,
                               n_clades=int(pick([1, 3, 8])), clade_div=float(pick([0.05, 0.2, 0.4])),
                               sub_hi=float(pick([0.02, 0.1, 0.3])), del_rate=float(pick([0.0, 0.01, 0.08])),
                               ins_rate=float(pick([0.0, 0.005, 0.05])), long_del_prob=float(pick([0.0, 0.3, 1.0])),
                               amb_rate=float(pick([0.0, 0.03])), lower_rate=float(pick([0.0, 0.1])))
        cs = util.cseqs_from_refs(refs)
        usable = [i for i in range(refs.n) if cs[i].size >= 40]
        if usable:
            break
    assert usable
    fam = [cs[i] for i in list(rng.permutation(usable)[:int(pick([1, 2, 7, 40, 60]))])]
    src = (refs.seq(usable[int(rng.integers(0, len(usable)))]) >> 24) & 0x0f
    lo = int(rng.integers(0, max(1, len(src) // 4)))
    qm = src[lo:].copy()
    mut = rng.random(len(qm)) < float(pick([0.0, 0.05, 0.3]))
    qm[mut] = rng.choice([1, 2, 4, 8, 15, 3], size=int(mut.sum()))
    qm = qm.astype(np.uint8)
    q = oracle.Cseq.from_packed("fz%d" % seed, np.arange(len(qm), dtype=np.uint32) | (qm.astype(np.uint32) << 24), len(qm))
    b = int(pick([4, 8]))
    strips = (len(qm) + 64 * b - 1) // (64 * b) + int(pick([0, 1]))
    # (narrow lanes make several strips out of these short queries)
    if strips < 2:
        b, strips = 4, max(2, (len(qm) + 255) // 256)
    if len(qm) > 64 * b * strips:
        strips = (len(qm) + 64 * b - 1) // (64 * b)
    util.set_knobs(monkeypatch, geom="%d,%d" % (64 * strips, b))
    rho = pick([None, None, "0.2", "0.7", "0.97", "2.5"])
    if rho:
        util.set_knobs(monkeypatch, rho=rho)
    if rng.integers(0, 3) == 0:
        util.set_knobs(monkeypatch, lds_kb=pick([5, 9]))
    gp, gpe = pick([(5, 2), (4, 1.5), (3, 3), (6, 0.5), (0.3, 0.1), (2, 3)])
    opts = dict(match_score=float(pick([2, 3, 0.7])), mismatch_score=float(pick([-1, -2, -0.1])), gap_penalty=float(gp),
                gap_ext_penalty=float(gpe), insertion=int(pick([0, 0, 0, 1])), fs_weight=float(pick([1.0, 0.0, 2.5])))
    ctx = capi.Context(0)
    try:
        cells = oracle.mesh_compute(fam, q, oracle.align_opts(**opts), weight=opts["fs_weight"])
        gb = ctx.graph_batch([util.graph_dict(fam, weight=opts["fs_weight"])], refs.width)
        popts = {k: v for k, v in opts.items() if k != "fs_weight"}
        vm, vs, val = ctx.debug_mesh(gb, qm, ctx.params(**popts), prune=True)
        if ctx.dp_info(0)["attempts"] == 0:
            assert (util.f32_bits(val) == util.f32_bits(cells["value"])).all()
            assert (vm == cells["value_midx"]).all() and (vs == cells["value_sidx"]).all()
        else:
            _check_planes(ctx, cells, vm, vs, val)
    finally:
        ctx.close()


@pytest.mark.parametrize("frac,rho,lanes", [(0.0, None, 0), (0.3, None, 1), (0.46, None, 0), (0.3, "0.97", 0), (0.46, "3", 1), (0.0, None, 1)])
def test_row_skip_partial_queries_equal_oracle(oracle, monkeypatch, frac, rho, lanes):
    """Queries that cover only part of the alignment -- 800-base windows at its start, in its middle and at its end:
    two strips, so the row skip is in play, and the alignment begins and ends somewhere inside the DAG: free starts
    (column 0 of any row), free ends (the last column of any row, any column of a sink), long stretches of rows on
    either side of the window that only the free-start rule keeps in play.  Trays against the oracle's -- walked back
    by either kernel (lanes: one lane per query, what launches of 8192 queries and more use)."""
    util.set_knobs(monkeypatch, bt_lanes=lanes)
    if rho is not None:
        util.set_knobs(monkeypatch, rho=rho)
    refs = synth.make_refs(3000, length=1500, width=50000, seed=72)
    qs = synth.make_queries(refs, 16, seed=73, window=(frac, 800))
    assert all(700 <= len(qs.seq(i)) <= 800 for i in range(qs.n))
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    st = pipeline.Store(":mem:prune-part-%s-%s-%s" % (frac, rho, lanes), refs)
    try:
        st.build_index(10, False)
        pl = pipeline.Pipeline(st, famfinder={"fs-min-len": 100})
        pl.run(qs.mask, qs.off, batch=16, inflight=1)
        n_dp = 0
        for qi in range(qs.n):
            q = util.query_cseq(qs, qi, upper=False)
            ids, sc, fflog = idx.famfinder(q, oracle.ff_opts(fs_min_len=100))
            got = pl.result(qi)
            if len(ids) == 0:
                assert got["status"] == 2
                continue
            want = oracle.align([cs[i] for i in ids], q, oracle.align_opts())
            assert got["status"] == want["status"], (qi, got["log"], want["log"])
            assert (got["packed"] == want["packed"]).all(), qi
            assert (got["head"], got["tail"], got["qual"]) == (want["head"], want["tail"], want["qual"])
            if want["status"] == 0:
                assert got["log"] == fflog + want["log"]
                n_dp += 1
        assert n_dp >= 12
        assert st.stats()["dp_queries_pruned"] >= n_dp
        pl.close()
    finally:
        st.close()


def _bench_like_cases(oracle, n):
    refs = synth.make_refs(1200, length=1500, width=50000, seed=91)
    qs = synth.make_queries(refs, n, seed=92)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    for qi in range(n):
        q = util.query_cseq(qs, qi)
        ids, sc, _ = idx.famfinder(q, oracle.ff_opts())
        fam = [cs[i] for i in ids]
        yield fam, q, (q.packed() >> 24).astype(np.uint8), refs.width


@pytest.mark.parametrize("slack", [0.0, 60.0, 1500.0, -30.0])
def test_scout_value_planes_equal_oracle(oracle, gpu_ctx, monkeypatch, slack):
    """Full-length 16S queries whose scout value is forced (SINA_HIP_TEST=scout_set) to the optimum plus a slack --
    exact (0), what the chain scout leaves (60), very loose (1500) and refuted (-30: below the optimum, the certificate
    fails and the query is swept again under what the first attempt found).  Whatever the value, every cell at or
    below the bound the final attempt used is the oracle's (value bits, value_midx, value_sidx), every other cell above
    its bound, and the tighter the value the fewer rows are swept."""
    swept = 0
    for fam, q, qm, width in _bench_like_cases(oracle, 4):
        cells = oracle.mesh_compute(fam, q, oracle.align_opts())
        g = util.graph_dict(fam)
        gb = gpu_ctx.graph_batch([g], width)
        util.set_knobs(monkeypatch, scout_set=None)
        vm, vs, val = gpu_ctx.debug_mesh(gb, qm, gpu_ctx.params(), prune=True)
        base = gpu_ctx.dp_info(0)
        optimum = base["raw"]
        util.set_knobs(monkeypatch, scout_set=repr(float(optimum) + slack))
        vm, vs, val = gpu_ctx.debug_mesh(gb, qm, gpu_ctx.params(), prune=True)
        info, alive = _check_planes(gpu_ctx, cells, vm, vs, val)
        assert util.f32_bits(np.float32(info["raw"])) == util.f32_bits(np.float32(optimum))
        if slack in (0.0, 60.0):
            assert info["attempts"] == 1
        if slack == 0.0:     # (no bound is tighter than the optimum itself)
            assert info["rows_swept"] <= base["rows_swept"], (info["rows_swept"], base["rows_swept"])
        if slack == -30.0:
            assert info["attempts"] == 2
        swept += info["rows_swept"]
    util.set_knobs(monkeypatch, scout_set=None)
