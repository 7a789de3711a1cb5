"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest

from sina_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu


def _family(oracle, refs, cs, idx, qs, qi):
    q = util.query_cseq(qs, qi)
    ids, sc, _ = idx.famfinder(q)
    fam = [cs[i] for i in ids]
    return q, fam, ids


@pytest.fixture(scope="module")
def small(oracle):
    refs = synth.make_refs(400, length=300, width=3000, seed=11, amb_rate=0.01, lower_rate=0.02)
    qs = synth.make_queries(refs, 12, seed=12, amb_rate=0.01)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    return refs, qs, cs, idx


@pytest.mark.parametrize("mode", ["simple", "weighted", "forbid", "weighted_forbid"])
def test_mesh_planes_bit_exact(oracle, gpu_ctx, small, mode):
    refs, qs, cs, idx = small
    rng = np.random.default_rng(5)
    weights = rng.uniform(0.2, 1.5, size=refs.width).astype(np.float32) if "weighted" in mode else None
    ins = 1 if "forbid" in mode else 0
    for qi in range(4):
        q, fam, _ = _family(oracle, refs, cs, idx, qs, qi)
        oo = oracle.align_opts(weights=weights, insertion=ins)
        cells = oracle.mesh_compute(fam, q, oo)
        g = util.graph_dict(fam)
        gb = gpu_ctx.graph_batch([g], refs.width)
        qm = (q.packed() >> 24).astype(np.uint8)
        vm, vs, val = gpu_ctx.debug_mesh(gb, qm, gpu_ctx.params(weights=weights, insertion=ins))
        assert (util.f32_bits(val) == util.f32_bits(cells["value"])).all()
        assert (vm == cells["value_midx"]).all()
        assert (vs == cells["value_sidx"]).all()


@pytest.mark.parametrize("overhang,lowercase", [(0, 0), (1, 0), (2, 2), (0, 2)])
def test_align_matches_oracle(oracle, gpu_ctx, small, overhang, lowercase):
    refs, qs, cs, idx = small
    graphs, qms, fams, qcs = [], [], [], []
    for qi in range(qs.n):
        q, fam, _ = _family(oracle, refs, cs, idx, qs, qi)
        if not fam:
            continue
        graphs.append(util.graph_dict(fam))
        qms.append((q.packed() >> 24).astype(np.uint8))
        fams.append(fam)
        qcs.append(q)
    qoff = np.zeros(len(qms) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(m) for m in qms])
    gb = gpu_ctx.graph_batch(graphs, refs.width)
    out, pos = gpu_ctx.align_graphs(gb, np.concatenate(qms), qoff, gpu_ctx.params(overhang=overhang))
    for i, (fam, q) in enumerate(zip(fams, qcs)):
        ref = oracle.align(fam, q, oracle.align_opts(overhang=overhang, lowercase=lowercase, realign=1))
        assert ref["status"] == 0
        o = out[i]
        assert o["status"] == 0
        assert o["cutoff_head"] == ref["head"] and o["cutoff_tail"] == ref["tail"]
        score = np.float32(o["raw"]) / np.float32(o["sum_weight"])
        assert util.f32_bits(score) == util.f32_bits(ref["score"])
        aligned, log = util.finish_alignment(qms[i], o, pos[int(qoff[i]):int(qoff[i + 1])], refs.width,
                                             lowercase_unaligned=(lowercase == 2))
        assert aligned == ref["aligned"]


def test_long_deletions_on_the_path(oracle, gpu_ctx, small):
    """Queries with 5..60 base chunks cut out: the optimal path crosses runs of gap-EXTENDING
    deletion cells, whose value_midx the kernel does not carry but resolves in backtrack
    (common.h kTbExt / kTbOpLast).  Planes (host-side resolution) and final alignments (device-side
    resolution) against the oracle."""
    from oracle import pyoracle as po
    refs, qs, cs, idx = small
    rng = np.random.default_rng(77)
    graphs, qms, fams, qcs = [], [], [], []
    for qi in range(6):
        q0, fam, _ = _family(oracle, refs, cs, idx, qs, qi)
        m = (q0.packed() >> 24).astype(np.uint8)
        keep = np.ones(len(m), bool)
        for _ in range(3):
            a = int(rng.integers(10, len(m) - 70))
            keep[a:a + int(rng.integers(5, 60))] = False
        m = m[keep]
        ab = np.arange(len(m), dtype=np.uint32) | (m.astype(np.uint32) << 24)
        q = po.Cseq.from_packed("cut%d" % qi, ab, len(m))
        graphs.append(util.graph_dict(fam))
        qms.append(m)
        fams.append(fam)
        qcs.append(q)
    n_ext = 0
    for g, m, fam, q in list(zip(graphs, qms, fams, qcs))[:3]:
        cells = oracle.mesh_compute(fam, q)
        vm, vs, val = gpu_ctx.debug_mesh(gpu_ctx.graph_batch([g], refs.width), m)
        assert (util.f32_bits(val) == util.f32_bits(cells["value"])).all()
        assert (vm == cells["value_midx"]).all() and (vs == cells["value_sidx"]).all()
    qoff = np.zeros(len(qms) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(m) for m in qms])
    out, pos = gpu_ctx.align_graphs(gpu_ctx.graph_batch(graphs, refs.width), np.concatenate(qms), qoff,
                                    gpu_ctx.params())
    for i, (fam, q) in enumerate(zip(fams, qcs)):
        ref = oracle.align(fam, q, oracle.align_opts(realign=1))
        o = out[i]
        assert ref["status"] == 0 and o["status"] == 0
        score = np.float32(o["raw"]) / np.float32(o["sum_weight"])
        assert util.f32_bits(score) == util.f32_bits(ref["score"])
        aligned, log = util.finish_alignment(qms[i], o, pos[int(qoff[i]):int(qoff[i + 1])], refs.width)
        assert aligned == ref["aligned"]
        # the cut really produces multi-column gaps in the aligned query
        gaps = np.diff(np.sort(pos[int(qoff[i]):int(qoff[i + 1])][:o["n_out"]]))
        n_ext += int((gaps > 8).sum())
    assert n_ext > 0


def test_forked_contexts_run_concurrently(oracle, gpu_ctx, small):
    """sina_hip_fork: contexts that share the store and index but own their stream and scratch give
    the parent's results when driven from several threads at once; a fork cannot change the store."""
    import threading
    refs, qs, cs, idx = small
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    gpu_ctx.build_index(10, False)
    want = gpu_ctx.kmer_topk(qs.mask, qs.off, 41)
    fam_ids = [np.asarray(want[0][qi, :want[2][qi]], np.uint32) for qi in range(qs.n)]
    foff = np.zeros(qs.n + 1, np.uint64)
    foff[1:] = np.cumsum([len(f) for f in fam_ids])
    masks = (qs.mask & 0x0f).astype(np.uint8)
    want_al = gpu_ctx.align_families(np.concatenate(fam_ids), foff, masks, qs.off, gpu_ctx.params())
    forks = [gpu_ctx.fork() for _ in range(3)]
    got, errs = [None] * 3, []

    def work(i):
        try:
            for _ in range(4):
                k = forks[i].kmer_topk(qs.mask, qs.off, 41)
                a = forks[i].align_families(np.concatenate(fam_ids), foff, masks, qs.off, forks[i].params())
            got[i] = (k, a)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for k, a in got:
        assert all((x == y).all() for x, y in zip(k, want))
        assert (a[0] == want_al[0]).all() and (a[1] == want_al[1]).all()
    with pytest.raises(capi.SinaHipError):
        forks[0].build_index(10, False)
    for f in forks:
        f.close()


def test_full_length_16s_geometry(oracle, gpu_ctx):
    """One full-length 16S-shaped problem (T=256,B=6 geometry, ~4.5 M cells)."""
    refs = synth.make_refs(600, length=1500, width=50000, seed=21)
    qs = synth.make_queries(refs, 2, seed=22)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    for qi in range(2):
        q, fam, _ = _family(oracle, refs, cs, idx, qs, qi)
        cells = oracle.mesh_compute(fam, q)
        g = util.graph_dict(fam)
        gb = gpu_ctx.graph_batch([g], refs.width)
        qm = (q.packed() >> 24).astype(np.uint8)
        vm, vs, val = gpu_ctx.debug_mesh(gb, qm)
        assert (util.f32_bits(val) == util.f32_bits(cells["value"])).all()
        assert (vm == cells["value_midx"]).all() and (vs == cells["value_sidx"]).all()


def test_kmer_scores_and_topk(oracle, gpu_ctx, small):
    refs, qs, cs, idx = small
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    off, ids = idx.csr()
    gpu_ctx.upload_index(10, False, off, ids)
    for qi in range(qs.n):
        q = util.query_cseq(qs, qi)
        assert (gpu_ctx.kmer_scores(qs.seq(qi)) == idx.scores(q)).all()
    for mx in (1, 41, 400, 5000):
        gi, gs, gn = gpu_ctx.kmer_topk(qs.mask, qs.off, mx)
        for qi in range(qs.n):
            oi, os_ = idx.find(util.query_cseq(qs, qi), mx)
            assert gn[qi] == len(oi)
            assert (gi[qi, :gn[qi]] == oi).all()
            assert (gs[qi, :gn[qi]] == os_).all()


@pytest.mark.parametrize("k", [4, 8, 12])
def test_device_index_build_equals_oracle_csr(oracle, gpu_ctx, small, k):
    """Index built on the device for k = 4 (256 long lists: all of them dense), 8 and 12 (the largest
    the C ABI takes: 16.8 M list slots), fast and no-fast: scores and top-41 equal the oracle's."""
    refs, qs, cs, idx = small
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    for nofast in (False, True):
        oidx = oracle.Index(cs, k=k, nofast=nofast)
        gpu_ctx.build_index(k, nofast)
        for qi in range(qs.n):
            q = util.query_cseq(qs, qi)
            assert (gpu_ctx.kmer_scores(qs.seq(qi)) == oidx.scores(q)).all()
        gi, gs, gn = gpu_ctx.kmer_topk(qs.mask, qs.off, 41)
        for qi in range(qs.n):
            oi, os_ = oidx.find(util.query_cseq(qs, qi), 41)
            assert gn[qi] == len(oi) and (gi[qi, :gn[qi]] == oi).all() and (gs[qi, :gn[qi]] == os_).all()


def test_device_family_graph_equals_oracle(oracle, gpu_ctx):
    """The DAG built on the GPU (sina_hip_align_families' first stage) vs mseq in the oracle: node order,
    columns, masks, weight bits, predecessor lists, successor minimum, sinks, and where each finished
    DP row is kept (LDS slot by liveness or spill row: a design detail of the kernel, checked against a
    straight Python model of the same greedy rule)."""
    refs = synth.make_refs(300, length=400, width=4000, seed=71, amb_rate=0.03, lower_rate=0.05, long_del_prob=0.4)
    cs = util.cseqs_from_refs(refs)
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    rng = np.random.default_rng(4)
    for fsw in (1.0, 0.0, 2.5):
        for F in (1, 2, 7, 40, 41, 100):
            ids = rng.choice(refs.n, size=F, replace=False).astype(np.uint32)
            for ring in (1, 3, 8):
                g = gpu_ctx.debug_family_graph(ids, fsw, ring)
                o = util.graph_dict([cs[i] for i in ids], fsw)
                assert g["n"] == o["n"]
                assert (g["pos"] == o["pos"]).all() and (g["mask"] == o["mask"]).all()
                assert (util.f32_bits(g["weight"]) == util.f32_bits(o["weight"])).all()
                assert (g["pred_off"] == o["pred_off"]).all() and (g["pred"] == o["pred"]).all()
                assert (g["succ_minpos"] == o["succ_minpos"]).all()
                sink = np.zeros(o["n"], np.uint8)
                sink[o["snk"]] = 1
                assert (g["sink"] == sink).all()
                assert (g["spill"] == util.row_store_model(o["pred_off"], o["pred"], ring)).all()


def test_device_family_graph_many_characters_per_column(oracle, gpu_ctx):
    """Families whose columns hold a dozen and more different characters (nine in ten bases an ambiguity code, half
    of them lower case): a tile of the DAG build then has more nodes than it keeps per-node words for in LDS and
    takes them in windows of whole columns; many nodes have more predecessors than a register list ever held, and
    the long deletions put predecessors further back than the 64 ids the per-node bit set spans."""
    refs = synth.make_refs(200, length=300, width=3000, seed=171, amb_rate=0.9, lower_rate=0.5, long_del_prob=0.5)
    cs = util.cseqs_from_refs(refs)
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    rng = np.random.default_rng(14)
    seen_wide = False
    for F in (3, 40, 128):
        ids = rng.choice(refs.n, size=F, replace=False).astype(np.uint32)
        g = gpu_ctx.debug_family_graph(ids, 1.0, 4)
        o = util.graph_dict([cs[i] for i in ids], 1.0)
        assert g["n"] == o["n"]
        assert (g["pos"] == o["pos"]).all() and (g["mask"] == o["mask"]).all()
        assert (util.f32_bits(g["weight"]) == util.f32_bits(o["weight"])).all()
        assert (g["pred_off"] == o["pred_off"]).all() and (g["pred"] == o["pred"]).all()
        assert (g["succ_minpos"] == o["succ_minpos"]).all()
        ncol = len(np.unique(o["pos"]))
        npred = np.diff(o["pred_off"])
        far = o["pred"].size and (np.repeat(np.arange(o["n"]), npred) - o["pred"]).max()
        seen_wide = seen_wide or (o["n"] > 6 * ncol and npred.max() > 8 and far > 64)
    assert seen_wide


@pytest.mark.parametrize("seed", range(int(os.environ.get("SINA_FUZZ_SEEDS", "12"))))
def test_device_family_graph_fuzz(oracle, gpu_ctx, seed):
    """Seeded random reference sets (length, alignment width -- the occupied-column bitmap's size --, ambiguity codes,
    lower case, long deletions, indel rates, clades) and families of 1 .. 128 members in random order: the DAG built on
    the GPU against mseq in the oracle, everything compared."""
    rng = np.random.default_rng(7000 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]  # noqa: E731
    length = int(pick([60, 150, 300, 700, 1300]))
    width = int(length * pick([2, 5, 33, 60]))
    refs = synth.make_refs(140, length=length, width=width, seed=7100 + seed, n_clades=int(pick([1, 3, 8])),
                           clade_div=float(pick([0.02, 0.12, 0.3])), del_rate=float(pick([0.0, 0.01, 0.1])),
                           ins_rate=float(pick([0.0, 0.005, 0.05])), long_del_prob=float(pick([0.0, 0.3, 0.9])),
                           amb_rate=float(pick([0.0, 0.02, 0.5])), lower_rate=float(pick([0.0, 0.1, 0.6])))
    cs = util.cseqs_from_refs(refs)
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    for F in (int(pick([1, 2, 3, 5])), int(pick([17, 40, 64])), int(pick([100, 127, 128]))):
        ids = rng.choice(refs.n, size=F, replace=False).astype(np.uint32)
        fsw, ring = float(pick([1.0, 0.0, 2.5])), int(pick([1, 3, 4, 8]))
        o = util.graph_dict([cs[i] for i in ids], fsw)
        try:
            g = gpu_ctx.debug_family_graph(ids, fsw, ring)
        except Exception as e:  # noqa: BLE001
            # the one documented refusal a random family can run into (32 768 spill rows per query -- a
            # family of 128 half-ambiguous members on a ring of one slot; seed 138 of a 400-seed run): it has to be
            # that limit, and the model has to agree that the DAG is beyond it
            if "too many spill rows" not in str(e):
                raise
            model = util.row_store_model(o["pred_off"], o["pred"], ring)
            assert int(((model != 0xFFFFFFFF) & (model >= 0x80000000)).sum()) > 32768, (seed, F)
            continue
        assert g["n"] == o["n"], (seed, F)
        assert (g["pos"] == o["pos"]).all() and (g["mask"] == o["mask"]).all(), (seed, F)
        assert (util.f32_bits(g["weight"]) == util.f32_bits(o["weight"])).all(), (seed, F)
        assert (g["pred_off"] == o["pred_off"]).all() and (g["pred"] == o["pred"]).all(), (seed, F)
        assert (g["succ_minpos"] == o["succ_minpos"]).all(), (seed, F)
        sink = np.zeros(o["n"], np.uint8)
        sink[o["snk"]] = 1
        assert (g["sink"] == sink).all(), (seed, F)
        assert (g["spill"] == util.row_store_model(o["pred_off"], o["pred"], ring)).all(), (seed, F)


def test_device_family_graph_at_the_widest_alignment(oracle, gpu_ctx):
    """The device DAG build at the widest alignment it takes (524 288 columns: the occupied-column bitmap and its ranks
    are 96 KB of the workgroup's LDS) with families of 40 and of 128 members (the entry table's 32 KB on top): the DAG
    against the oracle's."""
    refs = synth.make_refs(130, length=300, width=524288, seed=271, amb_rate=0.02, long_del_prob=0.3)
    assert refs.width == 524288
    cs = util.cseqs_from_refs(refs)
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    rng = np.random.default_rng(24)
    for F in (40, 128):
        ids = rng.choice(refs.n, size=F, replace=False).astype(np.uint32)
        g = gpu_ctx.debug_family_graph(ids, 1.0, 4)
        o = util.graph_dict([cs[i] for i in ids], 1.0)
        assert g["n"] == o["n"]
        assert (g["pos"] == o["pos"]).all() and (g["mask"] == o["mask"]).all()
        assert (util.f32_bits(g["weight"]) == util.f32_bits(o["weight"])).all()
        assert (g["pred_off"] == o["pred_off"]).all() and (g["pred"] == o["pred"]).all()
        assert o["pos"].max() > 500000


def test_align_families_equals_align_graphs(oracle, gpu_ctx, small):
    refs, qs, cs, idx = small
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    fams, graphs, qms = [], [], []
    for qi in range(qs.n):
        q, fam, ids = _family(oracle, refs, cs, idx, qs, qi)
        if len(ids) == 0:
            continue
        fams.append(ids.astype(np.uint32))
        graphs.append(util.graph_dict(fam))
        qms.append((q.packed() >> 24).astype(np.uint8))
    qoff = np.zeros(len(qms) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(m) for m in qms])
    foff = np.zeros(len(fams) + 1, np.uint64)
    foff[1:] = np.cumsum([len(f) for f in fams])
    for ins in (0, 1):
        p = gpu_ctx.params(insertion=ins)
        o1, p1 = gpu_ctx.align_graphs(gpu_ctx.graph_batch(graphs, refs.width), np.concatenate(qms), qoff, p)
        o2, p2 = gpu_ctx.align_families(np.concatenate(fams), foff, np.concatenate(qms), qoff, p)
        assert (o1 == o2).all() and (p1 == p2).all()


def test_queries_with_the_same_ordered_family_share_one_dag(oracle, gpu_ctx, small):
    """sina_hip_align_families builds ONE DAG per distinct ordered family of a launch and aligns every query that
    names it against that DAG (own trace-back cells, spill rows, edge records).  Eighteen queries over three
    families, interleaved, plus the same members in another ORDER (a different DAG: node order follows the family
    order): results equal those of per-query graphs handed over by the host (sina_hip_align_graphs, no sharing),
    plain and with --insertion=forbid (the kernel then also keeps succ_min), and the device built 4 DAGs for 20."""
    refs, qs, cs, idx = small
    gpu_ctx.upload_refs(refs.ab, refs.off, refs.width)
    base = []
    for qi in range(qs.n):
        q, fam, ids = _family(oracle, refs, cs, idx, qs, qi)
        if len(ids) >= 5:
            base.append((fam, ids.astype(np.uint32)))
        if len(base) == 3:
            break
    assert len(base) == 3
    fams, graphs, qms = [], [], []
    for qi in range(18):
        fam, ids = base[qi % 3]
        fams.append(ids)
        graphs.append(util.graph_dict(fam))
        qms.append((util.query_cseq(qs, (5 * qi) % qs.n).packed() >> 24).astype(np.uint8))
    for qi in (18, 19):  # family 0 with two members swapped: same set, another DAG
        fam, ids = base[0]
        perm = list(range(len(ids)))
        perm[0], perm[1] = perm[1], perm[0]
        fams.append(ids[perm])
        graphs.append(util.graph_dict([fam[i] for i in perm]))
        qms.append((util.query_cseq(qs, (5 * qi) % qs.n).packed() >> 24).astype(np.uint8))
    qoff = np.zeros(len(qms) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(m) for m in qms])
    foff = np.zeros(len(fams) + 1, np.uint64)
    foff[1:] = np.cumsum([len(f) for f in fams])
    for ins in (0, 1):
        p = gpu_ctx.params(insertion=ins)
        o1, p1 = gpu_ctx.align_graphs(gpu_ctx.graph_batch(graphs, refs.width), np.concatenate(qms), qoff, p)
        s0 = gpu_ctx.stats()
        o2, p2 = gpu_ctx.align_families(np.concatenate(fams), foff, np.concatenate(qms), qoff, p)
        s1 = gpu_ctx.stats()
        assert (o1 == o2).all() and (p1 == p2).all()
        assert s1["dags_used"] - s0["dags_used"] == 20 and s1["dags_built"] - s0["dags_built"] == 4


def test_mesh_planes_equal_reference_parts_hashes(oracle, gpu_ctx):
    """The DP kernel against planes the REFERENCE's scoring schemes and dag<T> produced (oracle/_ref cell
    loop, tests/golden/make_ref_vectors.py): value / value_midx / value_sidx of 25 families -- 1..41
    members, fs-weight 0 / 1 / 2.5, simple and weighted scheme, --insertion=forbid, gap-open ==
    gap-extend, one full-length 16S family -- compared by plane hash."""
    import os
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_vectors.npz"))
    hashes = ref["mesh_case_hash"]
    col = {f: i for i, f in enumerate(util.MESH_PLANES)}
    for ci, case in enumerate(util.MESH_CASES):
        fam, qa, width, w, sch = util.mesh_case_inputs(case)
        cs = [oracle.Cseq.from_packed("f%d" % i, a, width) for i, a in enumerate(fam)]
        g = util.graph_dict(cs, sch["fs_weight"])
        gb = gpu_ctx.graph_batch([g], width)
        p = gpu_ctx.params(weights=w, match_score=sch["match"], mismatch_score=sch["mismatch"],
                           gap_penalty=sch["gap"], gap_ext_penalty=sch["gapext"], fs_weight=sch["fs_weight"],
                           insertion=1 if sch["forbid"] else 0)
        vm, vs, val = gpu_ctx.debug_mesh(gb, (qa >> 24).astype(np.uint8), p)
        assert val.shape == tuple(ref["mesh_case_shape"][ci])
        assert util.plane_hash(val) == hashes[ci][col["value"]], (ci, case["scheme"])
        assert util.plane_hash(vm) == hashes[ci][col["value_midx"]], (ci, case["scheme"])
        assert util.plane_hash(vs) == hashes[ci][col["value_sidx"]], (ci, case["scheme"])
