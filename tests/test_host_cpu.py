"""CPU tests of the host side: the C ABI library loads and exports every symbol the
header declares, and the C++ stage mirror's host logic (cseq container, NAST fix-up,
family DAG) agrees with the oracle.  No GPU compute is invoked here."""
import ctypes as C
import json
import os
import re
import sys

import numpy as np
import pytest

from sina_amd import capi, pipeline, synth
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_abi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sina_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(sina_hip_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(capi.ABI_SYMBOLS)
    L = capi.load()
    for s in declared:
        assert hasattr(L, s), s
    assert L.sina_hip_abi_version() == 5


def test_abi_rejects_bad_arguments_without_gpu():
    L = capi.load()
    assert L.sina_hip_init(0, None) != 0
    assert b"null" in L.sina_hip_last_error()
    assert L.sina_hip_sync(None) != 0
    p = capi.AlignParams()
    L.sina_hip_align_params_default(C.byref(p))
    assert (p.match_score, p.mismatch_score, p.gap_penalty, p.gap_ext_penalty, p.fs_weight) == (2, -1, 5, 2, 1)


def _op(aligned, op=0, arg=0, what=0):
    H = pipeline.load_host()
    buf = C.create_string_buffer(4096)
    n, w = C.c_uint32(), C.c_uint32()
    rc = H.sina_host_cseq_op(aligned.encode(), op, arg, what, buf, len(buf), C.byref(n), C.byref(w))
    return rc, buf.value.decode(), n.value, w.value


def test_host_container_selftest():
    """Round 6's host machinery, checked inside the library (sina_host_selftest): a sequence kept as its mask bytes
    against the same sequence built base by base under every operation; an attribute kept as the list it is rendered
    from, through copies and the merged working copy; base lists out of the block pool handed between threads."""
    H = pipeline.load_host()
    err = C.create_string_buffer(256)
    H.sina_host_selftest.argtypes = [C.c_char_p, C.c_uint32]
    assert H.sina_host_selftest(err, 256) == 0, err.value.decode()


def test_host_cseq_matches_reference_kats():
    k = json.load(open(os.path.join(GOLD, "cseq_kat.json")))
    rc, s, n, w = _op(k["rna_aligned"])
    assert (rc, s, n, w) == (0, k["rna_aligned"], len(k["rna"]), len(k["rna_aligned"]))
    assert _op(k["rna_aligned"], what=1)[1] == k["rna_aligned_dots"]
    assert _op(k["rna_aligned"], what=3)[1] == k["rna"]
    for wdt, s in k["setwidth_chain"]:
        assert _op(k["rna_aligned"], 1, wdt)[:2] == (0, s)
    assert _op(k["rna_aligned"], 1, k["setwidth_throws"])[0] == 2      # runtime_error
    assert _op(k["rna_aligned"], 2)[1] == k["rna_aligned"][::-1]
    assert _op(k["rna"], 3, what=3)[1] == k["complement_bases"]
    low = k["rna_aligned"].lower()
    assert _op(low, what=2)[1] == low.replace("u", "t")
    assert _op(low, 4)[1] == k["rna_aligned"]
    assert _op("ACGX")[0] == 3                                          # bad_character_exception


def test_host_nast_fixup_equals_oracle(oracle):
    """50 random position vectors per setting: positions, case flags, log text and
    the throw/no-throw outcome must match the oracle."""
    H = pipeline.load_host()
    L = oracle.lib()
    rng = np.random.default_rng(3)
    n_shift = n_throw = 0
    for it in range(400):
        n = int(rng.integers(1, 60))
        width = int(rng.integers(n, n + 40)) if it % 5 else n   # tight widths force shifting
        if it % 7 == 0:
            width = max(1, n - int(rng.integers(1, 4)))        # more bases than columns: must throw
        pos = np.sort(rng.integers(0, width, size=n)).astype(np.uint32)
        if it % 3 == 0:
            pos[rng.integers(0, n):] = pos[-1]                  # pile-up at the end
        mask = rng.choice([1, 2, 4, 8], size=n).astype(np.uint32)
        ab = (pos | (mask << 24)).astype(np.uint32)
        for lowercase in (0, 1):
            mine = ab.copy()
            log = C.create_string_buffer(4096)
            rc = H.sina_host_fix_duplicates(mine.ctypes.data_as(capi.u32p), n, width, lowercase, 0, log, len(log))
            c = oracle.Cseq.from_packed("x", ab, width)
            lg = oracle.new_log()
            orc = L.so_cseq_fix_duplicate_positions(c.h, C.byref(lg), lowercase, 0)
            otxt = oracle.log_text(lg)
            L.so_log_free(C.byref(lg))
            assert (rc == 2) == (orc == -1)
            if rc == 0:
                assert (mine == c.packed()).all()
                assert log.value.decode() == otxt
                n_shift += "shifting" in otxt
            else:
                n_throw += 1
    assert n_shift > 20 and n_throw > 5     # the hard branches were exercised


def test_host_family_graph_equals_oracle(oracle):
    refs = synth.make_refs(80, length=300, width=3000, seed=41, amb_rate=0.02, lower_rate=0.03,
                           long_del_prob=0.3)
    cs = util.cseqs_from_refs(refs)
    st = pipeline.Store(":mem:hostgraph", refs)
    rng = np.random.default_rng(1)
    try:
        for fsw in (1.0, 0.0, 3.0):
            for _ in range(6):
                ids = rng.choice(refs.n, size=int(rng.integers(1, 41)), replace=False).astype(np.uint32)
                g = st.build_graph(ids, fsw)
                o = util.graph_dict([cs[i] for i in ids], fsw)
                assert g["n"] == o["n"]
                assert (g["pos"] == o["pos"]).all() and (g["mask"] == o["mask"]).all()
                assert (util.f32_bits(g["weight"]) == util.f32_bits(o["weight"])).all()
                assert (g["pred_off"] == o["pred_off"]).all() and (g["pred"] == o["pred"]).all()
                assert (g["succ_minpos"] == o["succ_minpos"]).all()
    finally:
        st.close()


def test_host_family_profile_equals_oracle(oracle):
    """--fs-no-graph on the host (build_family_profile, stages.cpp): the columns and the per-node match terms the
    aligner hands to the device against the oracle's pseq + base_profile::comp, bit for bit."""
    refs = synth.make_refs(80, length=300, width=3000, seed=43, amb_rate=0.03, lower_rate=0.03, long_del_prob=0.3)
    cs = util.cseqs_from_refs(refs)
    st = pipeline.Store(":mem:hostprofile", refs)
    rng = np.random.default_rng(2)
    try:
        for ms, mms, gp, gpe in [(-2.0, 1.0, 5.0, 2.0), (-3.0, 2.0, 4.0, 1.5), (-1.5, 0.5, 6.0, 0.5)]:
            for _ in range(5):
                ids = rng.choice(refs.n, size=int(rng.integers(1, 41)), replace=False).astype(np.uint32)
                pos, sc, own = st.build_profile(ids, ms, mms, gp, gpe)
                o = oracle.pseq_build([cs[i] for i in ids])
                assert len(pos) == o["n"] and (pos == o["pos"]).all()
                want = np.array([[oracle.profile_comp(o["prof"][m], code, ms, mms, gp, gpe) for code in range(1, 16)]
                                 for m in range(0, o["n"], max(1, o["n"] // 40))], np.float32)
                got = sc[::max(1, o["n"] // 40), 1:]
                assert (util.f32_bits(got) == util.f32_bits(want)).all()
                assert np.isinf(sc[:, 0]).all()          # (code 0: no query base has it)
                assert (util.f32_bits(own[1:]) == util.f32_bits(np.array(
                    [oracle.profile_comp(None, code, ms, mms, gp, gpe) for code in range(1, 16)], np.float32))).all()
    finally:
        st.close()


def test_option_names_and_validation():
    H = pipeline.load_host()
    H.sina_host_reset_options()
    assert H.sina_host_set_option(b"famfinder", b"fs-min", b"20") == 0
    assert H.sina_host_set_option(b"aligner", b"overhang", b"edge") == 0
    assert H.sina_host_set_option(b"aligner", b"overhang", b"sideways") != 0
    assert H.sina_host_set_option(b"famfinder", b"no-such-option", b"1") != 0
    assert H.sina_host_set_option(b"aligner", b"fs-no-graph", b"1") == 0      # the family as a profile: on this path
    assert H.sina_host_set_option(b"aligner", b"use-subst-matrix", b"1") != 0  # needs the ARB database: refused
    assert b"accelerated path" in H.sina_host_last_error()
    H.sina_host_reset_options()
    assert not H.sina_host_pipeline_create()          # no --db: "Must have reference database"
    assert b"reference database" in H.sina_host_last_error()


def test_affinity_plan_blocks():
    """sina_amd/affinity.py: ranks sharing a NUMA node split its cores into disjoint compact blocks."""
    from sina_amd import affinity
    node_cpus = {0: list(range(0, 64)), 1: list(range(64, 128))}
    allowed = set(range(256))
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    blocks = [affinity.plan(r, nodes, node_cpus, allowed) for r in range(8)]
    assert blocks[0] == list(range(0, 16)) and blocks[3] == list(range(48, 64)) and blocks[4] == list(range(64, 80))
    assert len(set(c for b in blocks for c in b)) == 8 * 16  # disjoint
    # one rank: the first sixteen cores of its node; a restricted mask is honoured
    assert affinity.plan(0, [1], node_cpus, allowed) == list(range(64, 80))
    assert affinity.plan(0, [0], node_cpus, set(range(8, 40))) == list(range(8, 24))
    # unknown node, or too few cores to be worth it: leave the affinity alone
    assert affinity.plan(0, [None], node_cpus, allowed) is None
    assert affinity.plan(0, [0, 0, 0, 0], {0: list(range(8))}, allowed) is None
    assert affinity._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_affinity_plan_without_numa_information():
    """No NUMA node known for the devices: the ranks split all allowed cores in order."""
    from sina_amd import affinity
    allowed = set(range(64))
    node_cpus = {None: list(range(64))}
    blocks = [affinity.plan(r, [None] * 4, node_cpus, allowed) for r in range(4)]
    assert blocks == [list(range(16 * r, 16 * r + 16)) for r in range(4)]


def test_dp_kernels_have_no_inflight_scalar_load_reads(tmp_path):
    """Guard for a bug this repository had: a hand-placed asynchronous scalar load in inline asm whose
    destination SGPRs the compiler spilled before the load had landed (B = 12 --insertion=forbid kernels:
    wild addresses on the GPU).  Compiles mesh_dp.hip to ISA exactly as the Makefile does (`make isa`) and
    lets tools/check_inflight_spills.py assert that no memory instruction sits inside inline asm in any DP
    kernel variant, that it saw the compiler's own scalar loads, and that the production kernels
    were among those looked at; tools/isa_mix.py must find their row loops."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "sina_amd", "csrc")
    out = str(tmp_path / "mesh_dp.s")
    subprocess.run(["make", "-C", src, "isa", "B=" + str(tmp_path)], check=True, capture_output=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_inflight_spills.py"), out,
                        "mesh_dp_simple_kernelILi8ELb0ELb0E", "mesh_dp_simple_kernelILi8ELb0ELb1E",
                        "mesh_dp_kernelILi8ELb1ELb1ELb0ELb0E",
                        "mesh_dp_kernelILi12ELb0ELb1ELb0ELb0E"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    m = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_mix.py"), out, "simple,8,0,0", "simple,8,0,1"],
                       capture_output=True, text=True)
    assert m.returncode == 0 and "mesh_dp_simple_kernel<8,0,0> row loop" in m.stdout, m.stdout + m.stderr
    assert "mesh_dp_simple_kernel<8,0,1> row loop" in m.stdout, m.stdout + m.stderr


def test_fasta_gzip_in_and_out(tmp_path):
    """Files ending in .gz are read and written through zlib, as the reference does with its
    boost::iostreams gzip filters (rw_fasta.cpp:200-202,358-360): same sequences as the plain files."""
    import gzip
    refs = synth.make_refs(40, length=200, width=1500, seed=91, amb_rate=0.02, lower_rate=0.05)
    text = "".join(">seq%d some description\n; key = value\n%s\n" % (i, synth.aligned_string(refs.seq(i), refs.width))
                   for i in range(refs.n))
    plain, packed = tmp_path / "in.fasta", tmp_path / "in.fasta.gz"
    plain.write_text(text)
    with gzip.open(packed, "wt") as f:
        f.write(text)
    out_plain, out_gz, out_gz2 = tmp_path / "o1.fasta", tmp_path / "o2.fasta", tmp_path / "o3.fasta.gz"
    assert pipeline.fasta_roundtrip(str(plain), str(out_plain)) == (refs.n, 0)
    assert pipeline.fasta_roundtrip(str(packed), str(out_gz)) == (refs.n, 0)
    assert pipeline.fasta_roundtrip(str(plain), str(out_gz2)) == (refs.n, 0)
    want = out_plain.read_text()
    assert want.count(">") == refs.n
    assert out_gz.read_text() == want
    with gzip.open(out_gz2, "rt") as f:
        assert f.read() == want
