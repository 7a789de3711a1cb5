"""CPU tests: the oracle against the reference's own known-answer tables and
against vectors produced by the real reference parts (tests/golden/)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import util

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(GOLD, "kmer_kat.json")))


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(GOLD, "ref_vectors.npz"))


# ------------------------------------------------------------------ k-mer KATs
# reference: src/unit_tests/kmer_test.cpp (kmer_generator_test, *_unique_*, *_prefix_*)

def test_kmer_generator_tables(oracle, kat):
    seq = kat["sequence"]
    for n, k in enumerate(kat["try_k"]):
        good, val = oracle.kmer_trace(seq, k)
        assert good.tolist() == kat["valid_k"][n]
        for i, g in enumerate(good):
            if g:
                assert val[i] == kat["kmers_k"][n][i]
        good_u, val_u = oracle.kmer_trace(seq, k, unique=True)
        assert good_u.tolist() == kat["first_k"][n]


def test_kmer_prefix_tables(oracle, kat):
    seq = kat["sequence"]
    for n, k in enumerate(kat["try_k"]):
        for p in range(2):
            plen, pval = kat["prefix_lens"][n][p], kat["prefixes"][n][p]
            good, val = oracle.kmer_trace(seq, k, plen, pval)
            expect = [int(v and (km >> ((k - plen) * 2)) == pval)
                      for v, km in zip(kat["valid_k"][n], kat["kmers_k"][n])]
            assert good.tolist() == expect
            assert int(good.sum()) == kat["prefix_counts"][n][p]


def test_kmer_iterables(oracle, kat):
    seq = kat["sequence"]
    c = oracle.Cseq("t", seq)
    ab = c.packed()
    for n, k in enumerate(kat["try_k"]):
        expect = [km for v, km in zip(kat["valid_k"][n], kat["kmers_k"][n]) if v]
        assert oracle.kmers(ab, k).tolist() == expect
        first = [km for v, km in zip(kat["first_k"][n], kat["kmers_k"][n]) if v]
        assert oracle.kmers(ab, k, unique=True).tolist() == first
        for p in range(2):
            plen, pval = kat["prefix_lens"][n][p], kat["prefixes"][n][p]
            assert len(oracle.kmers(ab, k, plen, pval)) == kat["prefix_counts"][n][p]
            assert len(oracle.kmers(ab, k, plen, pval, unique=True)) == kat["unique_prefix_counts"][n][p]


def test_kmer_generator_bounds(oracle):
    # kmer_generator(0) / (17) throw in the reference; the C ABI accepts 1..12 (4^k offsets in HBM)
    from sina_amd import capi
    assert capi  # bound is enforced in sina_hip_build_index / upload_index (see test_abi.py)


def test_final_kmer_is_dropped(oracle):
    """all_kmers("ACGTACGTAC", 4) yields 6 k-mers, the one ending on the last base is lost."""
    g = json.load(open(os.path.join(GOLD, "probe_dp.json")))
    ab = oracle.Cseq("t", "ACGTACGTAC").packed()
    assert oracle.kmers(ab, 4).tolist() == g["all_kmers_ACGTACGTAC_k4"]
    assert oracle.kmers(ab, 4, 1, 0).tolist() == g["prefix_kmers_ACGTACGTAC_k4_A"]


# ------------------------------------------------------------------ vectors from the real reference parts

def test_kmers_equal_reference(oracle, vec):
    modes = vec["kmer_modes"]
    off = vec["kmer_off"]
    ooff = vec["kmer_out_off"]
    j = 0
    for i in range(len(off) - 1):
        ab = vec["kmer_ab"][off[i]:off[i + 1]]
        for (k, pl, pv, u) in modes:
            got = oracle.kmers(ab, int(k), int(pl), int(pv), bool(u))
            assert got.tolist() == vec["kmer_out"][ooff[j]:ooff[j + 1]].tolist()
            j += 1


def _vl_parse(ser):
    inc, last, bytesize, size = np.frombuffer(ser[:16].tobytes(), np.uint32)
    return int(np.int16(inc & 0xffff)), int(last), int(size), ser[16:16 + bytesize]


def test_vlimap_bytes_equal_reference(oracle, vec):
    L = oracle.lib()
    for i, size in enumerate(vec["vl_sizes"]):
        data = vec["vl_set_%d" % i]
        v = C.c_void_p(L.so_vlimap_new(int(size)))
        for x in data:
            L.so_vlimap_push_back(v, int(x))
        ptr = oracle.u8p()
        nb = L.so_vlimap_bytes(v, C.byref(ptr))
        mine = np.ctypeslib.as_array(ptr, shape=(nb,)).copy() if nb else np.zeros(0, np.uint8)
        inc, last, sz, payload = _vl_parse(vec["vl_ser_%d" % i])
        assert inc == 1 and sz == len(data)
        assert mine.tolist() == payload.tolist()
        L.so_vlimap_invert(v)
        nb = L.so_vlimap_bytes(v, C.byref(ptr))
        mine = np.ctypeslib.as_array(ptr, shape=(nb,)).copy() if nb else np.zeros(0, np.uint8)
        inc, last, sz, payload = _vl_parse(vec["vl_inv_%d" % i])
        assert inc == -1
        assert mine.tolist() == payload.tolist()
        # inverted increment(): decrements members of the complement and reports offset 1
        cnt = np.ones(max(int(size), 1), np.int16)
        assert L.so_vlimap_increment(v, cnt.ctypes.data_as(oracle.i16p)) == 1
        expect = np.zeros(max(int(size), 1), np.int16)
        expect[data] = 1
        assert (cnt[:size] == expect[:size]).all()
        L.so_vlimap_free(v)


def _dag_family(oracle, vec):
    off = vec["dag_off"]
    return [oracle.Cseq.from_packed("f%d" % i, vec["dag_ab"][off[i]:off[i + 1]], int(vec["dag_width"]))
            for i in range(len(off) - 1)]


def test_family_dag_equals_reference(oracle, vec):
    fam = _dag_family(oracle, vec)
    for wi in range(3):
        p = "dag%d_" % wi
        g = oracle.mseq_build(fam, float(vec[p + "fs_weight"]))
        assert (vec[p + "ids"] == np.arange(g["n"])).all()          # id == list order == mesh row
        assert (g["pos"] == vec[p + "pos"]).all()
        assert (g["mask"] == vec[p + "mask"]).all()
        assert (util.f32_bits(g["weight"]) == util.f32_bits(vec[p + "weight"])).all()
        assert (g["pred_off"] == vec[p + "pred_off"]).all()
        assert (g["pred"] == vec[p + "pred"]).all()
        assert (g["src"] == vec[p + "src"]).all() and (g["snk"] == vec[p + "snk"]).all()


def test_mesh_cells_equal_reference_scheme(oracle, vec):
    """All seven cell fields, bit for bit, against the cell loop that runs the REAL
    scoring_scheme_simple on the REAL dag<T> (oracle/ref_parts.cpp)."""
    fam = _dag_family(oracle, vec)
    for qi in range(3):
        qa = vec["mesh_q%d" % qi]
        q = oracle.Cseq.from_packed("q", qa, len(qa))
        cells = oracle.mesh_compute(fam, q)
        for f in ("value_midx", "value_sidx", "gapm_idx", "gaps_idx"):
            assert (cells[f] == vec["mesh%d_%s" % (qi, f)]).all(), f
        for f in ("value", "gapm_val", "gaps_val"):
            assert (cells[f].view(np.uint32) == vec["mesh%d_%s" % (qi, f)]).all(), f


def test_scoring_ops_equal_reference(oracle, vec):
    L = oracle.lib()
    w = np.ascontiguousarray(vec["score_weights"])
    for pr, want in zip(vec["score_probes"], vec["score_vals"]):
        op, prev_b, mpos, mc, mw_b, sc, offs, weighted = [int(x) for x in pr]
        prev = np.uint32(prev_b).view(np.float32)
        mw = np.uint32(mw_b).view(np.float32)
        got = L.so_score_op(op, prev, mpos, L.so_char_to_mask(mc), mw, L.so_char_to_mask(sc), offs, -2.0, 1.0, 5.0,
                            2.0, w.ctypes.data_as(oracle.f32p) if weighted else None, len(w))
        assert np.float32(got).view(np.uint32) == want


# ------------------------------------------------------------------ the one recorded end-to-end reference vector

def test_probe_alignment_vector(oracle):
    g = json.load(open(os.path.join(GOLD, "probe_dp.json")))
    fam = [oracle.Cseq("r%d" % i, s) for i, s in enumerate(g["refs"])]
    q = oracle.Cseq("q", g["query"])
    assert oracle.mseq_build(fam)["n"] == g["nodes"]
    r = oracle.align(fam, q)
    assert r["aligned"] == g["aligned"]
    assert r["head"] == g["head"] and r["tail"] == g["tail"]
    assert "%08x" % int(util.f32_bits(r["score"])) == g["score_bits"]
    assert "raw=%s, weight=%s," % (g["raw_str"], g["weight_str"]) in r["log"]


# ------------------------------------------------------------------ cseq KATs (src/unit_tests/cseq_test.cpp)

def test_cseq_kats_oracle(oracle):
    k = json.load(open(os.path.join(GOLD, "cseq_kat.json")))
    L = oracle.lib()
    c = oracle.Cseq("x", k["rna_aligned"])
    assert c.size == len(k["rna"]) and c.width == len(k["rna_aligned"])
    assert c.bases() == k["rna"] and c.aligned(nodots=True) == k["rna_aligned"]
    assert c.aligned(nodots=False) == k["rna_aligned_dots"]
    for w, s in k["setwidth_chain"]:
        assert L.so_cseq_set_width(c.h, w) == 0
        assert c.aligned(nodots=True) == s
    assert L.so_cseq_set_width(c.h, k["setwidth_throws"]) == -1
    c = oracle.Cseq("x", k["rna_aligned"])
    L.so_cseq_reverse(c.h)
    assert c.aligned(nodots=True) == k["rna_aligned"][::-1]
    L.so_cseq_reverse(c.h)
    assert c.aligned(nodots=True) == k["rna_aligned"]
    c = oracle.Cseq("x", k["rna"])
    L.so_cseq_complement(c.h)
    assert c.bases() == k["complement_bases"]
    low = oracle.Cseq("x", k["rna_aligned"].lower())
    assert low.aligned(nodots=True, dna=True) == k["rna_aligned"].lower().replace("u", "t")
    L.so_cseq_upper(low.h)
    assert low.aligned(nodots=True) == k["rna_aligned"]
    with pytest.raises(ValueError):
        oracle.Cseq("x", "ACGX")


# ------------------------------------------------------------------ posting-list properties (src/unit_tests/idset_test.cpp)

@pytest.mark.parametrize("size", [0, 255, 256, 257, 10000])
@pytest.mark.parametrize("fill", [0, 10, 50, 100])
@pytest.mark.parametrize("seed", [132456, 54321, 242424])
def test_vlimap_properties(oracle, size, fill, seed):
    L = oracle.lib()
    rng = np.random.default_rng(seed)
    n = size * fill // 100
    data = np.sort(rng.choice(size, n, replace=False)).astype(np.uint32) if n else np.zeros(0, np.uint32)
    expected = np.zeros(max(size, 1), np.int16)
    expected[data] = 1
    a, b = C.c_void_p(L.so_vlimap_new(size)), C.c_void_p(L.so_vlimap_new(size))
    mid = len(data) // 2
    for x in data[:mid]:
        L.so_vlimap_push_back(a, int(x))
    for x in data[mid:]:
        L.so_vlimap_push_back(b, int(x))
    cnt = np.zeros(max(size, 1), np.int16)
    assert L.so_vlimap_increment(a, cnt.ctypes.data_as(oracle.i16p)) == 0
    L.so_vlimap_increment(b, cnt.ctypes.data_as(oracle.i16p))
    assert (cnt == expected).all()
    L.so_vlimap_append(a, b)
    cnt[:] = 0
    L.so_vlimap_increment(a, cnt.ctypes.data_as(oracle.i16p))
    assert (cnt == expected).all()
    L.so_vlimap_invert(a)
    cnt[:] = 1
    assert L.so_vlimap_increment(a, cnt.ctypes.data_as(oracle.i16p)) == 1
    assert (cnt[:size] == expected[:size]).all()
    L.so_vlimap_free(a)
    L.so_vlimap_free(b)


# ------------------------------------------------------------------ independent definition of the k-mer score

def _brute_kmers(mask, k, fast):
    out = []
    for e in range(k - 1, len(mask) - 1):           # window may not end on the last base
        w = mask[e - k + 1:e + 1] & 0xf
        if any(bin(int(x)).count("1") != 1 for x in w):
            continue
        v = 0
        for x in w:
            v = (v << 2) | (int(x).bit_length() - 1)
        if fast and (v >> (2 * (k - 1))) != 0:
            continue
        out.append(v)
    return out


@pytest.mark.parametrize("k,nofast", [(4, False), (6, True), (10, False)])
def test_index_scores_match_definition(oracle, k, nofast):
    from sina_amd import synth
    refs = synth.make_refs(60, length=120, width=600, seed=31, amb_rate=0.03, n_clades=2, clade_div=0.05)
    qs = synth.make_queries(refs, 6, seed=32, amb_rate=0.02)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=k, nofast=nofast)
    sets = [set(_brute_kmers((refs.seq(i) >> 24).astype(np.uint8), k, not nofast)) for i in range(refs.n)]
    for qi in range(qs.n):
        qk = _brute_kmers(qs.seq(qi), k, not nofast)
        want = np.array([sum(1 for v in qk if v in s) for s in sets], np.int16)
        got = idx.scores(util.query_cseq(qs, qi, upper=False))
        assert (got == want).all()
        ids, sc = idx.find(util.query_cseq(qs, qi, upper=False), 7)
        order = sorted(range(refs.n), key=lambda r: (-int(want[r]), -r))[:7]
        assert ids.tolist() == order and sc.tolist() == [float(want[r]) for r in order]


# ------------------------------------------------------------------ reference-parts harness present in this checkout?

def test_oracle_matches_live_reference_parts(oracle):
    """When oracle/_ref was built (build container, or shipped to the GPU box), re-check a
    fresh random case live instead of only the committed vectors."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built in this checkout")
    from sina_amd import synth
    R = oracle.ref()
    refs = synth.make_refs(30, length=200, width=1200, seed=77, amb_rate=0.02, lower_rate=0.03)
    for i in range(refs.n):
        ab = np.ascontiguousarray(refs.seq(i))
        for (k, pl, pv, u) in [(10, 1, 0, 1), (10, 1, 0, 0), (7, 0, 0, 0)]:
            buf = np.zeros(len(ab) + 1, np.uint32)
            n = R.ref_kmers(ab.ctypes.data_as(oracle.u32p), len(ab), k, pl, pv, u, buf.ctypes.data_as(oracle.u32p))
            assert oracle.kmers(ab, k, pl, pv, bool(u)).tolist() == buf[:n].tolist()


def test_turn_check_orientations(oracle):
    """so_turn_check (famfinder.cpp:344-378): a query handed in reversed / complemented / both is told
    apart by its top-1 k-mer scores; without --turn=all only "as is" and "reversed and complemented"
    are searched (the other two score 0); ties and all-zero fall back to the lowest orientation."""
    import numpy as np
    from sina_amd import synth
    from tests import util
    refs = synth.make_refs(200, length=300, width=3000, seed=91)
    cs = util.cseqs_from_refs(refs)
    idx = oracle.Index(cs, k=10)
    qs = synth.make_queries(refs, 6, seed=92)
    L = oracle.lib()
    for qi in range(qs.n):
        for o in range(4):
            q = util.query_cseq(qs, qi)
            if o & 1:
                L.so_cseq_reverse(q.h)
            if o & 2:
                L.so_cseq_complement(q.h)
            best, sc = idx.turn_check(q, True)
            assert best == o and sc[o] == sc.max() and sc[o] > 4 * np.delete(sc, o).max()
            best2, sc2 = idx.turn_check(q, False)
            assert sc2[1] == 0 and sc2[2] == 0
            assert best2 == (o if o in (0, 3) else (3 if sc2[3] > sc2[0] else 0))
    # a query without any k-mer: every score 0 -> orientation 0
    q = oracle.Cseq.from_packed("tiny", np.arange(5, dtype=np.uint32) | (np.uint32(1) << 24), 5)
    assert idx.turn_check(q, True)[0] == 0


def _mesh_case_oracle_inputs(oracle, case):
    from tests import util
    fam, qa, width, w, sch = util.mesh_case_inputs(case)
    cs = [oracle.Cseq.from_packed("f%d" % i, a, width) for i, a in enumerate(fam)]
    q = oracle.Cseq.from_packed("q", qa, len(qa))
    opts = oracle.align_opts(weights=w, match_score=sch["match"], mismatch_score=sch["mismatch"],
                             gap_penalty=sch["gap"], gap_ext_penalty=sch["gapext"], fs_weight=sch["fs_weight"],
                             insertion=1 if sch["forbid"] else 0)
    return cs, q, opts


def test_mesh_planes_equal_reference_parts_hashes(oracle):
    """All seven cell planes of 25 families (1..41 members, fs-weight 0 / 1 / 2.5, simple and weighted
    scheme, --insertion=forbid, gap-open == gap-extend, one full-length 16S family) equal what the
    reference's own scoring schemes and dag<T> produce in the oracle/_ref cell loop -- compared by
    plane hash (tests/golden/make_ref_vectors.py; inputs regenerated from the synth seeds)."""
    import numpy as np
    from tests import util
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_vectors.npz"))
    shapes, hashes = ref["mesh_case_shape"], ref["mesh_case_hash"]
    assert len(shapes) == len(util.MESH_CASES) >= 25
    for ci, case in enumerate(util.MESH_CASES):
        cs, q, opts = _mesh_case_oracle_inputs(oracle, case)
        cells = oracle.mesh_compute(cs, q, opts)
        assert cells.shape == tuple(shapes[ci]), ci
        got = [util.plane_hash(cells[f]) for f in util.MESH_PLANES]
        assert got == list(hashes[ci]), (ci, case["scheme"])


# ---- --fs-no-graph: pseq + scoring_scheme_profile (src/pseq.{h,cpp}, src/scoring_schemes.h:37-100).
# The reference holds no test or fixture for this option and pseq.h cannot be compiled here (it
# includes cseq.h, which needs Boost): the restatement is checked against values worked by hand from
# the published formulas -- "parity unpinned" for this option, as DESIGN.md says.
def test_profile_columns_by_hand(oracle):
    fam = [oracle.Cseq("a", "AC-GU"), oracle.Cseq("b", "A--GR"), oracle.Cseq("c", "-CNG-")]
    p = oracle.pseq_build(fam)
    assert p["width"] == 5 and list(p["pos"]) == [0, 1, 2, 3, 4]
    f = np.float32
    # column 0: A, A, leading gap (extended): 12, 12 points + 12 for the gap
    assert np.array_equal(p["prof"][0], np.array([24, 0, 0, 0, 0, 12], f) / f(36))
    # column 1: C, gap opened (b), C
    assert np.array_equal(p["prof"][1], np.array([0, 0, 24, 0, 12, 0], f) / f(36))
    # column 2: gap opened (a), gap extended (b), N = 3 points each
    assert np.array_equal(p["prof"][2], np.array([3, 3, 3, 3, 12, 12], f) / f(36))
    # column 3: G G G
    assert np.array_equal(p["prof"][3], np.array([0, 36, 0, 0, 0, 0], f) / f(36))
    # column 4: U, R = A|G 6 points each, gap opened (c)
    assert np.array_equal(p["prof"][4], np.array([6, 6, 0, 12, 12, 0], f) / f(36))


def test_profile_skips_unoccupied_columns_but_keeps_column_zero(oracle):
    fam = [oracle.Cseq("a", "--A---C-"), oracle.Cseq("b", "--A---G-")]
    p = oracle.pseq_build(fam)
    assert list(p["pos"]) == [0, 2, 6]  # column 0 first, occupied or not (pseq.cpp:67-69)
    assert np.array_equal(p["prof"][0], np.array([0, 0, 0, 0, 0, 1], np.float32))
    assert np.array_equal(p["prof"][2], np.array([0, .5, .5, 0, 0, 0], np.float32))


def test_profile_comp_by_hand(oracle):
    f = np.float32
    col = np.array([6, 6, 0, 12, 12, 0], f) / f(36)
    ms, mms, gp, gpe = f(-2), f(1), f(5), f(2)
    for mask, shares in [(1, [1, 0, 0, 0]), (8, [0, 0, 0, 1]), (3, [.5, .5, 0, 0]), (15, [.25] * 4),
                         (14, [0, f(1) / f(3), f(1) / f(3), f(1) / f(3)])]:
        res = f(0)
        for i in range(4):          # sixteen products, i outer and j inner, each rounded to float
            for j in range(4):
                res = f(res + f(f((ms if i == j else mms) * col[i]) * f(shares[j])))
        want = f(f(res + f(gp * col[4])) + f(gpe * col[5]))
        assert oracle.profile_comp(col, mask, ms, mms, gp, gpe) == want
    # a base against itself: the "had there been a match" weight of backtrack()
    assert oracle.profile_comp(None, 1, ms, mms, gp, gpe) == f(-2)
    assert oracle.profile_comp(None, 3, ms, mms, gp, gpe) == f(f(-2 * .25) * 2 + f(1 * .25) * 2)


def test_profile_alignment_without_family_gaps_keeps_the_columns(oracle):
    fam = [oracle.Cseq("a", "--ACGGUUAGCAAUGCAGGCU-"), oracle.Cseq("b", "--ACAGUUCGCAAUGCAGGCU-"),
           oracle.Cseq("c", "--ACGGUUAGCUAUGCAGCCU-")]
    q = oracle.Cseq("q", "ACGGUUAGCAAUGGAGGCU")  # (one substitution: no member contains it, no copy shortcut)
    r = oracle.align(fam, q, oracle.align_opts(fs_no_graph=1))
    assert r["status"] == 0
    assert r["aligned"].replace(".", "-") == "--ACGGUUAGCAAUGGAGGCU-"
    # 18 of 19 bases scored (the first one starts the path): raw = 1 + sum of comp(), weight = 19 * -2
    assert "weight=-38, query-len=19, aligned-bases=19" in r["log"]
