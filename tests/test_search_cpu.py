"""CPU tests of the search stage: the oracle's cseq_comparator and the host
stage's restatement against the reference's own known-answer table
(src/unit_tests/cseq_comparator_test.cpp, extracted to tests/golden/cseq_comparator_kat.json), and the
pieces of search_filter::operator() that need no GPU (ranking order, nearest string, LCA vote)."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from sina_amd import pipeline

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT = json.load(open(os.path.join(GOLD, "cseq_comparator_kat.json")))


def _bits(x):
    return int(np.float32(x).view(np.uint32))


def test_oracle_comparator_reference_kats(oracle):
    cs = {k: po.Cseq(k, v) for k, v in KAT["sequences"].items()}
    assert len(KAT["checks"]) == 121
    for c in KAT["checks"]:
        got = po.compare(cs[c["a"]], cs[c["b"]], c["iupac"], c["dist"], c["cover"], c["filter_lc"])
        if "same_as" in c:
            o = c["same_as"]
            want = po.compare(cs[o["a"]], cs[o["b"]], o["iupac"], o["dist"], o["cover"], o["filter_lc"])
            assert _bits(got) == _bits(want), c
        else:
            assert _bits(got) == c["expect_bits"], c


def test_host_comparator_reference_kats():
    seqs = KAT["sequences"]
    for c in KAT["checks"]:
        args = (po.IUPAC_RULES[c["iupac"]], po.DIST_RULES[c["dist"]], po.COVER_RULES[c["cover"]], c["filter_lc"])
        got, _ = pipeline.host_compare(seqs[c["a"]], seqs[c["b"]], *args)
        if "same_as" in c:
            o = c["same_as"]
            want, _ = pipeline.host_compare(seqs[o["a"]], seqs[o["b"]], po.IUPAC_RULES[o["iupac"]],
                                            po.DIST_RULES[o["dist"]], po.COVER_RULES[o["cover"]], o["filter_lc"])
            assert _bits(got) == _bits(want), c
        else:
            assert _bits(got) == c["expect_bits"], c


def test_host_and_oracle_counters_agree_on_random_pairs(oracle):
    rng = np.random.default_rng(7)
    alphabet = np.array(list("ACGUacgu-----MRN"))
    for it in range(300):
        n = int(rng.integers(8, 60))
        a = "".join(rng.choice(alphabet, size=n))
        b = "".join(rng.choice(alphabet, size=n))
        if not any(ch.isupper() for ch in a) or not any(ch.isupper() for ch in b):
            continue  # (the reference dereferences end() when a side has no unfiltered base)
        for iupac in range(3):
            for flc in (False, True):
                _, got = pipeline.host_compare(a, b, iupac, 0, 4, flc)
                want = po.compare_counts(po.Cseq("a", a), po.Cseq("b", b), list(po.IUPAC_RULES)[iupac], flc)
                assert got == want, (a, b, iupac, flc)


def test_jukes_cantor_and_cover_rules(oracle):
    counts = (3, 5, 2, 7, 80, 9)  # only_a_overhang, only_b_overhang, only_a, only_b, match, mismatch
    base = {"query": 80 + 9 + 2 + 3, "target": 80 + 9 + 7 + 5, "overlap": 80 + 9 + 2 + 7, "all": 80 + 9 + 17,
            "average": 80 + 9 + 17 // 2, "min": 89 + 5, "max": 89 + 12, "nogap": 89}
    for cover, b in base.items():
        frac = np.float32(80) / np.float32(b)
        assert _bits(po.compare_score(counts, cover, "none")) == _bits(frac)
        with np.errstate(invalid="ignore"):  # (identity > 0.75: log of a negative number, NaN on both sides)
            jc = np.float32(-3.0 / 4 * np.log(1.0 - 4.0 / 3 * float(frac)))
        assert _bits(po.compare_score(counts, cover, "jc")) == _bits(jc)
    assert po.compare_score(counts, "abs", "none") == 80


def test_lca_vote(oracle):
    A = "Bacteria;Proteobacteria;Gamma;Entero;"
    B = "Bacteria;Proteobacteria;Gamma;Pseudo;"
    C_ = "Bacteria;Firmicutes;Bacilli;"
    assert po.search_lca([A, A, A], 0.7) == A
    assert po.search_lca([A, B, A, B], 0.7) == "Bacteria;Proteobacteria;Gamma;"
    # quorum .7 of 10 results tolerates 3 outliers
    assert po.search_lca([A] * 7 + [C_] * 3, 0.7) == A
    assert po.search_lca([A] * 6 + [C_] * 4, 0.7) == "Bacteria;"
    assert po.search_lca([], 0.7) == "Unclassified;"
    assert po.search_lca(["Unclassified;", "Unclassified;"], 0.7) == "Unclassified;"
    assert po.search_lca([A, "Archaea;Eury;"], 1.0) == "Unclassified;"
    assert po.search_lca(["Bacteria; ", "Bacteria; "], 0.7) == "Bacteria;"


def test_nearest_string(oracle):
    txt = po.search_nearest(["AB1", "CD2"], ["1", "2"], ["5", "0"], ["1500", "1400"], [1, 0],
                            np.array([0.98765, 0.5], np.float32))
    assert txt == "CD2.2.0.1400~0.988 AB1.1.5.1500~0.500 "


# ---- the reference's .sidx index cache

@pytest.mark.parametrize("k,nofast", [(4, False), (4, True), (6, True), (10, False)])
def test_sidx_files_match_the_oracle_writer_and_reader(oracle, tmp_path, k, nofast):
    """The host's writer produces the bytes kmer_search::impl::store would (restated in the oracle,
    whose vlimaps are byte-exact against the real idset.h) -- inverted lists included -- and its reader
    recovers the CSR index from the oracle's file."""
    from sina_amd import synth
    from tests import util
    refs = synth.make_refs(90, length=220, width=1600, seed=77 + k, amb_rate=0.01)
    cs = util.cseqs_from_refs(refs)
    idx = po.Index(cs, k=k, nofast=nofast)
    off, ids = idx.csr()
    want = str(tmp_path / "want.sidx")
    got = str(tmp_path / "got.sidx")
    idx.write_sidx(want)
    pipeline.sidx_store(got, refs.n, off, ids, k=k, nofast=nofast)
    a, b = open(want, "rb").read(), open(got, "rb").read()
    assert len(a) == len(b)
    # bytes 10-11 and 18-23 of the header are struct padding (uninitialised in the reference)
    keep = lambda x: x[:10] + x[12:18] + x[24:]
    assert keep(a) == keep(b)
    n, off2, ids2 = pipeline.sidx_load(want, k=k, nofast=nofast)
    assert n == refs.n and (off2 == off).all() and (ids2 == ids).all()
    if k == 4:  # short k-mers: many lists are longer than n/2 and stored inverted
        assert b.count(b"\xff\xff\xff\xff") > 20
    # wrong k / fast setting are refused, as try_load does
    with pytest.raises(pipeline.HostError):
        pipeline.sidx_load(want, k=k + 1, nofast=nofast)
    with pytest.raises(pipeline.HostError):
        pipeline.sidx_load(want, k=k, nofast=not nofast)
    # and the oracle's reader accepts the host's file
    again = po.Index.read_sidx(got, cs, k=k, nofast=nofast)
    o3, i3 = again.csr()
    assert (o3 == off).all() and (i3 == ids).all()


# ---- FASTA reader / writer rules (src/rw_fasta.cpp:229-315,437-541)

def test_fasta_reader_and_writer_rules(tmp_path):
    src = str(tmp_path / "in.fasta")
    open(src, "w").write(
        "junk before the first record\n"
        ">seq1 Escherichia coli K12\r\n"
        "; acc = AB000001 \n"
        ";just a remark\n"
        "--AGCU\n"
        "agcu..\n"
        ">seq2\n"
        "ACGTNRY-acgu\n"
        ">bad has a digit\n"
        "ACG7ACG\n"
        "ACGU\n"
        ">seq3\tdescription after a tab\n"
        "A-C\n")
    dst = str(tmp_path / "out.fasta")
    assert pipeline.fasta_roundtrip(src, dst) == (3, 1)
    assert open(dst).read() == (">seq1 Escherichia coli K12\n--AGCUagcu--\n"
                                ">seq2\nACGUNRY-acgu\n"
                                ">seq3 description after a tab\nA-C\n")
    # dots for leading/trailing gaps, DNA alphabet, wrapped lines, attributes as comments
    assert pipeline.fasta_roundtrip(src, dst, {"fasta-write-dots": True, "fasta-write-dna": True, "line-length": 5,
                                               "meta-fmt": "comment"}) == (3, 1)
    assert open(dst).read() == (">seq1 Escherichia coli K12\n; acc=AB000001\n..AGC\nTagct\n..\n"
                                ">seq2\nACGTN\nRY-ac\ngt\n"
                                ">seq3 description after a tab\nA-C\n")
    assert pipeline.fasta_roundtrip(src, dst, {"meta-fmt": "header"}) == (3, 1)
    assert open(dst).read().splitlines()[0] == ">seq1 Escherichia coli K12 [acc=AB000001]"
    assert pipeline.fasta_roundtrip(src, dst, {"meta-fmt": "csv"}) == (3, 1)
    csv = open(str(tmp_path / "out.csv"), newline="").read()
    assert csv.startswith("name,acc,full_name\r\nseq1,AB000001,Escherichia coli K12\r\nseq2\r\n")


def test_fasta_reader_edge_cases(tmp_path):
    """Line-cutter corners of the FASTA source (behaviour of src/rw_fasta.cpp:229-315 on std::istream):
    a last sequence line without a line feed is data, a header line without one is not a record, a line
    longer than the reader's block, CR LF files, records cut by --fasta-block, csv quoting, damaged gzip."""
    import gzip
    src, dst = str(tmp_path / "in.fasta"), str(tmp_path / "out.fasta")
    # (1) the file ends inside a sequence line / inside a header line
    open(src, "w").write(">a\nAC-GU\n>b x y\nAC\nGU")
    assert pipeline.fasta_roundtrip(src, dst) == (2, 0)
    assert open(dst).read() == ">a\nAC-GU\n>b x y\nACGU\n"
    open(src, "w").write(">a\nACGU\n>b")
    assert pipeline.fasta_roundtrip(src, dst) == (1, 0)
    open(src, "w").write("")
    assert pipeline.fasta_roundtrip(src, dst) == (0, 0) and open(dst).read() == ""
    # (2) CR LF throughout, empty lines inside a record, a comment line with '=' in its value
    open(src, "wb").write(b">a  two blanks\r\n; k = v=w \r\nAC\r\n\r\nGU\r\n>c\r\n\r\n")
    assert pipeline.fasta_roundtrip(src, dst, {"meta-fmt": "comment"}) == (2, 0)
    assert open(dst).read() == ">a  two blanks\n; k=v=w\nACGU\n>c\n\n"
    # (3) one 9 MB sequence line (longer than the 4 MB block) between two ordinary records
    big = "-" * 4_500_000 + "ACGU" + "-" * 4_500_000
    open(src, "w").write(">s\nAC\n>big\n" + big + "\n>t\nGU\n")
    assert pipeline.fasta_roundtrip(src, dst) == (3, 0)
    out = open(dst).read().split("\n")
    assert out[0:2] == [">s", "AC"] and out[2] == ">big" and out[3] == big and out[4:6] == [">t", "GU"]
    # (4) --fasta-block / --fasta-idx: every record is read by exactly one slice
    recs = "".join(">r%d\n%s\n" % (i, "ACGU" * (3 + i % 5)) for i in range(40))
    open(src, "w").write(recs)
    seen = []
    for idx in range((len(recs) + 99) // 100):
        n, _ = pipeline.fasta_roundtrip(src, dst, {"fasta-block": 100, "fasta-idx": idx})
        seen += [l[1:] for l in open(dst).read().splitlines() if l.startswith(">")]
    # (a slice takes the records that START in it, plus the one its last byte falls into -- the
    # reference's rule, rw_fasta.cpp:239-242 -- so neighbours may both hold a boundary record)
    assert sorted(set(seen), key=lambda s: int(s[1:])) == ["r%d" % i for i in range(40)]
    # (5) csv quoting: quotes doubled, fields with comma / quote / line break in quotes
    open(src, "w").write(">q say \"hi\", twice\n; note = a,b\nACGU\n")
    assert pipeline.fasta_roundtrip(src, dst, {"meta-fmt": "csv"}) == (1, 0)
    assert open(str(tmp_path / "out.csv"), newline="").read() == \
        'name,full_name,note\r\nq,"say ""hi"", twice","a,b"\r\n'
    # (6) a truncated gzip file is an error, not a short input
    gz = str(tmp_path / "in.fasta.gz")
    with gzip.open(gz, "wt") as f:
        f.write("".join(">r%d\n%s\n" % (i, "ACGU" * 500) for i in range(200)))
    whole = open(gz, "rb").read()
    open(gz, "wb").write(whole[:len(whole) // 2])
    with pytest.raises(pipeline.HostError):
        pipeline.fasta_roundtrip(gz, dst)


# ---- reference id order of an ARB database (host/id_order.h; query_arb.cpp:160,470-474,732-739)
def _boost_hash_string(s):
    """boost::hash<std::string>, Boost 1.62 ... 1.80 on a 64-bit target: hash_range over the characters with
    the 64-bit hash_combine (boost/container_hash/hash.hpp), written here from the published algorithm."""
    M64 = (1 << 64) - 1
    m = 0xc6a4a7935bd1e995
    seed = 0
    for ch in s.encode():
        k = ch if ch < 128 else ch | 0xFFFFFFFFFFFFFF00  # (plain char is signed on x86-64)
        k = (k * m) & M64
        k ^= k >> 47
        k = (k * m) & M64
        seed ^= k
        seed = (seed * m) & M64
        seed = (seed + 0xe6546b64) & M64
    return seed


def test_reference_order_is_the_walk_of_a_hash_table():
    from sina_amd import pipeline
    rng = np.random.default_rng(5)
    names = ["%s%05d.%d" % ("".join(chr(65 + int(c)) for c in rng.integers(0, 26, 2)), int(rng.integers(0, 99999)), i)
             for i in range(5000)]
    order, hashes, buckets = pipeline.reference_order(names)
    assert sorted(order.tolist()) == list(range(5000))          # a permutation of the database order
    assert order.tolist() != list(range(5000))
    assert [int(h) for h in hashes[:200]] == [_boost_hash_string(s) for s in names[:200]]
    assert _boost_hash_string("") == 0 and int(pipeline.reference_order(["\xe9"])[1][0]) == _boost_hash_string("\xe9")
    assert buckets >= 5000                                      # (load factor <= 1)
    # a walk of the table visits a bucket's names together: the bucket of consecutive ids changes
    # exactly (number of occupied buckets - 1) times
    b = [int(hashes[j]) % buckets for j in order]
    changes = sum(1 for x, y in zip(b, b[1:]) if x != y)
    assert changes == len(set(b)) - 1
    # the same names, the same order; a repeated name keeps one id
    assert pipeline.reference_order(names)[0].tolist() == order.tolist()
    assert len(pipeline.reference_order(["a", "b", "a"])[0]) == 2


def test_store_opened_in_arb_order(tmp_path):
    from sina_amd import pipeline
    rng = np.random.default_rng(6)
    names = ["Seq%04d" % int(x) for x in rng.permutation(3000)[:400]]
    p = tmp_path / "db.fasta"
    with open(p, "w") as f:
        for i, nm in enumerate(names):
            f.write(">%s\n%s\n" % (nm, "-" * (i % 7) + "ACGU" * 5 + "-" * (7 - i % 7)))
    order, _, _ = pipeline.reference_order(names)
    st = pipeline.Store.open(str(p), id_order="arb")
    assert [st.name(i) for i in range(400)] == [names[j] for j in order]
    assert st.name(400) == ""
    st.close()
    st = pipeline.Store.open(str(p))
    assert [st.name(i) for i in range(400)] == names
    st.close()
    with pytest.raises(ValueError):
        pipeline.Store.open(str(p), id_order="sorted")
