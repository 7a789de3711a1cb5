"""GPU parity at the edges of the path: degenerate sizes, the largest DP geometry, error returns of
the C ABI, and size-independent properties at full-length scale (self-alignment, determinism)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from sina_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu


def _cseq(name, masks):
    m = np.asarray(masks, np.uint8)
    ab = np.arange(len(m), dtype=np.uint32) | (m.astype(np.uint32) << 24)
    return po.Cseq.from_packed(name, ab, len(m)), m


def _planes_equal(oracle, gpu_ctx, fam, q, qm, width, **opts):
    cells = oracle.mesh_compute(fam, q, oracle.align_opts(**opts) if opts else None)
    gb = gpu_ctx.graph_batch([util.graph_dict(fam)], width)
    vm, vs, val = gpu_ctx.debug_mesh(gb, qm, gpu_ctx.params(**opts) if opts else None)
    assert (util.f32_bits(val) == util.f32_bits(cells["value"])).all()
    assert (vm == cells["value_midx"]).all() and (vs == cells["value_sidx"]).all()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SINA_FUZZ_SEEDS", "10"))))
def test_mesh_plane_fuzz(oracle, gpu_ctx, monkeypatch, seed):
    """Seeded random families (1 - 60 members, heavy to no divergence, long deletions that make
    far-away predecessors, ambiguity codes, lower case), scoring parameters, insertion rule, node-weight
    scale and DP geometry (incl. a single LDS row slot): value / value_midx / value_sidx planes of the
    production kernel bit-exact against the oracle's mesh."""
    for attempt in range(8):  # (short references under long deletions can all come out empty: draw again)
        rng = np.random.default_rng(9000 + seed + 100000 * attempt)
        pick = lambda xs: xs[int(rng.integers(0, len(xs)))]  # noqa: E731
        length = int(pick([60, 150, 320, 700]))
        refs = synth.make_refs(int(pick([8, 40, 90])), length=length, width=int(length * pick([3, 8])), seed=9100 + seed,
                               n_clades=int(pick([1, 3, 8])), clade_div=float(pick([0.05, 0.2, 0.4])),
                               sub_hi=float(pick([0.02, 0.1, 0.3])), del_rate=float(pick([0.0, 0.01, 0.08])),
                               ins_rate=float(pick([0.0, 0.005, 0.05])), long_del_prob=float(pick([0.0, 0.3, 1.0])),
                               amb_rate=float(pick([0.0, 0.03])), lower_rate=float(pick([0.0, 0.1])))
        cs = util.cseqs_from_refs(refs)
        usable = [i for i in range(refs.n) if cs[i].size >= 20]
        if usable:
            break
    assert usable
    nfam = int(pick([1, 2, 7, 40, 60]))
    fam = [cs[i] for i in list(rng.permutation(usable)[:nfam])]
    src = (refs.seq(usable[int(rng.integers(0, len(usable)))]) >> 24) & 0x0f
    lo = int(rng.integers(0, max(1, len(src) // 3)))
    hi = int(rng.integers(min(len(src), lo + 5), len(src) + 1))
    qm = src[lo:hi].copy()
    mut = rng.random(len(qm)) < float(pick([0.0, 0.05, 0.3]))
    qm[mut] = rng.choice([1, 2, 4, 8, 15, 3], size=int(mut.sum()))
    q, qm = _cseq("fuzz%d" % seed, qm)
    geom = pick([None, None, "64,4", "128,4", "64,8", "128,8", "64,12", "128,12"])
    if geom and len(qm) <= int(geom.split(",")[0]) * int(geom.split(",")[1]):
        util.set_knobs(monkeypatch, geom=geom)
    if rng.integers(0, 3) == 0:
        util.set_knobs(monkeypatch, lds_kb=pick([5, 9]))
    gp, gpe = pick([(5, 2), (4, 1.5), (2, 3), (3, 3), (6, 0.5), (0.3, 0.1)])
    opts = dict(match_score=float(pick([2, 3, 0.7])), mismatch_score=float(pick([-1, -2, -0.1])), gap_penalty=float(gp),
                gap_ext_penalty=float(gpe), insertion=int(pick([0, 0, 1])), fs_weight=float(pick([1.0, 0.0, 2.5])))
    if rng.integers(0, 3) == 0:
        opts["weights"] = rng.uniform(0.2, 1.6, size=refs.width).astype(np.float32)
    ctx = capi.Context(0)   # (a context of its own: the LDS budget is read when it is created)
    try:
        cells = oracle.mesh_compute(fam, q, oracle.align_opts(**opts), weight=opts["fs_weight"])
        gb = ctx.graph_batch([util.graph_dict(fam, weight=opts["fs_weight"])], refs.width)
        popts = {k: v for k, v in opts.items() if k != "fs_weight"}
        vm, vs, val = ctx.debug_mesh(gb, qm, ctx.params(**popts))
        assert (util.f32_bits(val) == util.f32_bits(cells["value"])).all()
        assert (vm == cells["value_midx"]).all() and (vs == cells["value_sidx"]).all()
    finally:
        ctx.close()


def test_tiny_queries_and_tiny_families(oracle, gpu_ctx):
    """Queries of 1, 2, 3, 5 and 13 bases; families of one and of two (identical) references."""
    refs = synth.make_refs(30, length=120, width=900, seed=301)
    cs = util.cseqs_from_refs(refs)
    rng = np.random.default_rng(302)
    n_aligned = 0
    for L in (1, 2, 3, 5, 13, 14):
        q, qm = _cseq("tiny%d" % L, rng.choice([1, 2, 4, 8], size=L))
        for fam in ([cs[3]], [cs[4], cs[4]], [cs[i] for i in (1, 7, 9, 20)]):
            _planes_equal(oracle, gpu_ctx, fam, q, qm, refs.width)
            want = oracle.align(fam, q, oracle.align_opts(realign=1))
            gb = gpu_ctx.graph_batch([util.graph_dict(fam)], refs.width)
            out, pos = gpu_ctx.align_graphs(gb, qm, np.array([0, L], np.uint64), gpu_ctx.params())
            assert out[0]["status"] == 0
            if want["status"] != 0:
                continue  # the aligner drops family members that CONTAIN the query (align.cpp:337-348)
            n_aligned += 1
            aligned, _ = util.finish_alignment(qm, out[0], pos[:L], refs.width)
            assert aligned == want["aligned"]
    assert n_aligned >= 3


def test_ragged_batch_mixed_lengths(oracle, gpu_ctx):
    """One launch with query lengths from 17 to 700 against families of 1..40: every query picks its
    own rows of the batch-wide geometry."""
    refs = synth.make_refs(200, length=600, width=5000, seed=311, amb_rate=0.01)
    cs = util.cseqs_from_refs(refs)
    rng = np.random.default_rng(312)
    graphs, qms, fams, qcs = [], [], [], []
    for i, L in enumerate((19, 700, 17, 350, 33, 512, 64, 129)):
        src = (refs.seq(int(rng.integers(refs.n))) >> 24) & 0x0f
        # (short ones random: a piece of a reference would be CONTAINED in family members, which the
        # aligner then removes from the family, align.cpp:337-348 -- not what this test is about)
        m = np.resize(src, L).astype(np.uint8) if L >= 100 else rng.choice([1, 2, 4, 8], size=L).astype(np.uint8)
        m[m == 0] = 1
        q, qm = _cseq("rag%d" % i, m)
        fam = [cs[j] for j in rng.choice(refs.n, size=int(rng.integers(1, 41)), replace=False)]
        graphs.append(util.graph_dict(fam)); qms.append(qm); fams.append(fam); qcs.append(q)
    qoff = np.zeros(len(qms) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(m) for m in qms])
    out, pos = gpu_ctx.align_graphs(gpu_ctx.graph_batch(graphs, refs.width), np.concatenate(qms), qoff,
                                    gpu_ctx.params())
    for i, (fam, q) in enumerate(zip(fams, qcs)):
        want = oracle.align(fam, q, oracle.align_opts(realign=1))
        assert out[i]["status"] == want["status"] == 0
        assert util.f32_bits(np.float32(out[i]["raw"]) / np.float32(out[i]["sum_weight"])) == util.f32_bits(want["score"])
        aligned, _ = util.finish_alignment(qms[i], out[i], pos[int(qoff[i]):int(qoff[i + 1])], refs.width)
        assert aligned == want["aligned"]


@pytest.mark.parametrize("ref_len,lo,hi", [(6000, 100, 5900), (8400, 60, 8251), (10400, 30, 10270)])
def test_largest_geometry(oracle, gpu_ctx, ref_len, lo, hi):
    """A 5800-base query (768x8: 12 strips), one of 8191 bases (16 strips of 512 columns, round 2's limit)
    and one of 10240 = SINA_HIP_MAX_QUERY_LEN -- against a 3-member family of references that long: 70 - 140 M cells,
    planes bit-exact."""
    refs = synth.make_refs(6, length=ref_len, width=8 * ref_len, seed=321, n_clades=2)
    cs = util.cseqs_from_refs(refs)
    src = (refs.seq(2) >> 24) & 0x0f
    assert hi <= len(src)
    q, qm = _cseq("long", src[lo:hi])
    _planes_equal(oracle, gpu_ctx, [cs[0], cs[2], cs[5]], q, qm, refs.width)


def test_error_returns(oracle, gpu_ctx):
    """The C ABI reports what it cannot do instead of computing something else."""
    refs = synth.make_refs(140, length=100, width=700, seed=331)
    cs = util.cseqs_from_refs(refs)
    g = util.graph_dict([cs[0], cs[1]])
    too_long = np.ones(10241, np.uint8)
    with pytest.raises(capi.SinaHipError):
        gpu_ctx.align_graphs(gpu_ctx.graph_batch([g], refs.width), too_long, np.array([0, 10241], np.uint64),
                             gpu_ctx.params())
    with pytest.raises(capi.SinaHipError):   # empty query
        gpu_ctx.align_graphs(gpu_ctx.graph_batch([g], refs.width), np.ones(1, np.uint8), np.array([0, 0], np.uint64),
                             gpu_ctx.params())
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    ids = np.arange(129, dtype=np.uint32)     # families are capped at 128 members
    with pytest.raises(capi.SinaHipError):
        gpu_ctx.align_families(ids, np.array([0, 129], np.uint64), np.ones(20, np.uint8),
                               np.array([0, 20], np.uint64), gpu_ctx.params())
    with pytest.raises(capi.SinaHipError):   # reference id out of range
        gpu_ctx.align_families(np.array([5, 9999], np.uint32), np.array([0, 2], np.uint64), np.ones(20, np.uint8),
                               np.array([0, 20], np.uint64), gpu_ctx.params())
    fresh = capi.Context(0)
    with pytest.raises(capi.SinaHipError):   # k-mer search without an index
        fresh.kmer_topk(np.ones(30, np.uint8), np.array([0, 30], np.uint64), 5)
    fresh.close()


def test_kmer_search_degenerate_queries(oracle, gpu_ctx):
    """Queries shorter than k, all-ambiguous queries, a query that IS a reference, and max larger
    than the store: scores and (score desc, id desc) order as the oracle's."""
    refs = synth.make_refs(300, length=200, width=1500, seed=341, amb_rate=0.02)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    gpu_ctx.build_index(10, False)
    rng = np.random.default_rng(342)
    masks = [rng.choice([1, 2, 4, 8], size=5), rng.choice([1, 2, 4, 8], size=10), np.full(40, 15),
             (refs.seq(17) >> 24) & 0x0f, rng.choice([1, 2, 4, 8, 5, 15], size=90)]
    qoff = np.zeros(len(masks) + 1, np.int64)
    qoff[1:] = np.cumsum([len(m) for m in masks])
    flat = np.concatenate(masks).astype(np.uint8)
    for mx in (1, 7, 300, 1000):
        gi, gs, gn = gpu_ctx.kmer_topk(flat, qoff, mx)
        for qi, m in enumerate(masks):
            q, _ = _cseq("d%d" % qi, m)
            oi, os_ = idx.find(q, mx)
            assert gn[qi] == len(oi)
            assert (gi[qi, :gn[qi]] == oi).all() and (gs[qi, :gn[qi]] == os_).all()
    assert gpu_ctx.kmer_topk(flat, qoff, 1)[0][3, 0] == 17     # the reference finds itself first


def test_full_length_properties(oracle, gpu_ctx):
    """Size-independent properties on full-length 16S-shaped inputs (no oracle run: too large to be
    quick on the CPU): a reference aligned against a family that contains it lands (almost
    entirely) on its own columns, in order; the same batch twice gives identical bytes; results do not
    depend on how the batch is cut into launches."""
    refs = synth.make_refs(3000, length=1500, width=50000, seed=351)
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    gpu_ctx.build_index(10, False)
    pick = np.arange(0, 3000, 125)                        # 24 references as queries
    masks = [((refs.seq(int(i)) >> 24) & 0x0f).astype(np.uint8) for i in pick]
    qoff = np.zeros(len(masks) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(m) for m in masks])
    flat = np.concatenate(masks)
    ids, sc, n = gpu_ctx.kmer_topk(flat, qoff, 40)
    assert all(ids[q, 0] == pick[q] for q in range(len(pick)))          # each finds itself first
    fam = [np.asarray(ids[q, :n[q]], np.uint32) for q in range(len(pick))]
    foff = np.zeros(len(pick) + 1, np.uint64)
    foff[1:] = np.cumsum([len(f) for f in fam])
    out, pos = gpu_ctx.align_families(np.concatenate(fam), foff, flat, qoff, gpu_ctx.params())
    for q, i in enumerate(pick):
        cols = (refs.seq(int(i)) & 0xFFFFFF).astype(np.uint32)
        a, b = int(qoff[q]), int(qoff[q + 1])
        assert out[q]["status"] == 0 and out[q]["n_out"] == b - a
        # backtrack emits the columns from the last base to the first, counted from the right edge.
        # (Not necessarily ALL on its own columns: a better-conserved neighbouring column can score
        # higher than the reference's own, the node weight grows with the members sharing it.)
        got = refs.width - 1 - pos[a:b][::-1]
        assert (got == cols).mean() > 0.98
        assert (np.diff(got.astype(np.int64)) >= 0).all()
        assert out[q]["cutoff_head"] == 0 and out[q]["cutoff_tail"] == 0
    out2, pos2 = gpu_ctx.align_families(np.concatenate(fam), foff, flat, qoff, gpu_ctx.params())
    assert out.tobytes() == out2.tobytes() and (pos == pos2).all()
    for q in (0, 7, 23):                                                   # one query per launch
        o1, p1 = gpu_ctx.align_families(fam[q], np.array([0, len(fam[q])], np.uint64), masks[q],
                                        np.array([0, len(masks[q])], np.uint64), gpu_ctx.params())
        assert o1[0].tobytes() == out[q].tobytes()
        assert (p1[:len(masks[q])] == pos[int(qoff[q]):int(qoff[q + 1])]).all()


@pytest.mark.parametrize("scores", [
    (2.0, -1.0, 5.0, 2.0),          # the defaults
    (1.7, -0.9, 3.3, 0.7),          # nothing exactly representable: every sum rounds
    (2.0, -1.0, 2.0, 2.0),          # open == extend: ties between opening and extending everywhere
    (0.1, -0.1, 0.3, 0.1),          # small values, long runs of equal-cost choices
    (3.0, -2.0, 1.0, 0.25),         # cheap gaps: insertion runs cross many lanes
    (2.0, -1.0, 1.0, 3.0),          # extend > open: the general chain path
    (2.0, -1.0, 2000.0, 700.0),     # gap costs so large that values reach the 1e6 initial value of a cell
])
def test_scoring_parameter_sets_planes(oracle, gpu_ctx, scores):
    """All three planes bit-equal for scoring parameters that are not small integers (the insertion
    chain guesses run structure with single adds and must still end at the reference's repeated
    float adds), for a query with long insertions against its family, in a one-wave and a two-wave
    geometry."""
    ms, mms, gp, gpe = scores
    for length, width, seed in ((500, 4000, 331), (1450, 50000, 332)):
        refs = synth.make_refs(120, length=length, width=width, seed=seed)
        cs = util.cseqs_from_refs(refs)
        rng = np.random.default_rng(seed + 1)
        fam = [cs[i] for i in rng.choice(refs.n, size=12, replace=False)]
        src = ((refs.seq(int(rng.integers(refs.n))) >> 24) & 0x0f).astype(np.uint8)
        # two inserted stretches (40 and 150 bases) and one deleted stretch
        ins1 = rng.choice([1, 2, 4, 8], size=40).astype(np.uint8)
        ins2 = rng.choice([1, 2, 4, 8], size=150).astype(np.uint8)
        a, b, c = len(src) // 5, len(src) // 2, (3 * len(src)) // 4
        qmask = np.concatenate([src[:a], ins1, src[a:b], ins2, src[b:c], src[c + 60:]])
        q, qm = _cseq("ins%d" % length, qmask)
        _planes_equal(oracle, gpu_ctx, fam, q, qm, refs.width, match_score=ms, mismatch_score=mms,
                      gap_penalty=gp, gap_ext_penalty=gpe)


def test_malformed_graphs_are_rejected(oracle, gpu_ctx):
    """sina_hip_align_graphs validates host-supplied DAGs instead of truncating them: a predecessor
    that is not an earlier node, or more than 255 predecessors of one node, is an error."""
    refs = synth.make_refs(40, length=100, width=700, seed=332)
    cs = util.cseqs_from_refs(refs)
    g = util.graph_dict([cs[0], cs[1], cs[2]])
    q = np.ones(30, np.uint8)
    qoff = np.array([0, 30], np.uint64)
    bad = dict(g)
    bad["pred"] = g["pred"].copy()
    m = int(np.flatnonzero(np.diff(g["pred_off"]) > 0)[0])
    bad["pred"][g["pred_off"][m]] = m            # a node as its own predecessor
    with pytest.raises(capi.SinaHipError):
        gpu_ctx.align_graphs(gpu_ctx.graph_batch([bad], refs.width), q, qoff, gpu_ctx.params())
    n = 300                                       # node 299 with 256 predecessors
    poff = np.zeros(n + 1, np.uint32)
    poff[1:299] = np.arange(0, 298)               # a chain: node i <- i-1
    npred_last = 256
    pred = list(range(0, 297)) + list(range(0, npred_last))
    poff[299] = 297
    poff[300] = 297 + npred_last
    wide = dict(n=n, pos=np.arange(n, dtype=np.uint32), mask=np.ones(n, np.uint8), weight=np.ones(n, np.float32),
                pred_off=poff, pred=np.array(pred, np.uint32), succ_minpos=np.arange(1, n + 1, dtype=np.uint32))
    with pytest.raises(capi.SinaHipError):
        gpu_ctx.align_graphs(gpu_ctx.graph_batch([wide], 400), q, qoff, gpu_ctx.params())
