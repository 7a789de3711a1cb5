"""ctypes bindings for the test oracle (oracle/liboracle.so) and, when built,
the reference-parts harness (oracle/_ref/libsina_refparts.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing in sina_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libsina_refparts.so")

u32p = C.POINTER(C.c_uint32)
f32p = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_uint8)
i16p = C.POINTER(C.c_int16)


def build():
    """(Re)build liboracle.so and, if /root/reference exists, _ref/."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


class Log(C.Structure):
    _fields_ = [("s", C.c_char_p), ("n", C.c_size_t), ("cap", C.c_size_t)]


class FFOpts(C.Structure):
    _fields_ = [("fs_min", C.c_uint32), ("fs_max", C.c_uint32), ("fs_msc", C.c_float),
                ("fs_msc_max", C.c_float), ("fs_leave_query_out", C.c_int), ("fs_req", C.c_uint32),
                ("fs_req_full", C.c_uint32), ("fs_full_len", C.c_uint32), ("fs_req_gaps", C.c_uint32),
                ("fs_min_len", C.c_uint32), ("fs_cover_gene", C.c_uint32)]


class AlignOpts(C.Structure):
    _fields_ = [("match_score", C.c_float), ("mismatch_score", C.c_float), ("gap_penalty", C.c_float),
                ("gap_ext_penalty", C.c_float), ("fs_weight", C.c_float), ("overhang", C.c_int),
                ("lowercase", C.c_int), ("insertion", C.c_int), ("realign", C.c_int),
                ("weights", f32p), ("n_weights", C.c_uint32), ("fs_no_graph", C.c_int)]


class AlignResult(C.Structure):
    _fields_ = [("status", C.c_int), ("head", C.c_int), ("tail", C.c_int), ("qual", C.c_int),
                ("score", C.c_float), ("cells", C.c_uint64), ("idty", C.c_float)]


class Graph(C.Structure):
    _fields_ = [("n", C.c_uint32), ("width", C.c_uint32), ("pos", u32p), ("mask", u8p), ("weight", f32p),
                ("pred_off", u32p), ("pred", u32p), ("succ_off", u32p), ("succ", u32p),
                ("n_src", C.c_uint32), ("src", u32p), ("n_snk", C.c_uint32), ("snk", u32p), ("prof", f32p)]


class MatchCounts(C.Structure):
    _fields_ = [("only_a_overhang", C.c_int), ("only_b_overhang", C.c_int), ("only_a", C.c_int),
                ("only_b", C.c_int), ("match", C.c_int), ("mismatch", C.c_int)]


class SearchOpts(C.Structure):
    _fields_ = [("kmer_candidates", C.c_uint32), ("max_result", C.c_uint32), ("min_sim", C.c_float),
                ("lca_quorum", C.c_float), ("ignore_super", C.c_int), ("search_all", C.c_int),
                ("iupac", C.c_int), ("dist", C.c_int), ("cover", C.c_int), ("filter_lc", C.c_int)]


IUPAC_RULES = {"optimistic": 0, "pessimistic": 1, "exact": 2}
DIST_RULES = {"none": 0, "jc": 1}
COVER_RULES = {"abs": 0, "query": 1, "target": 2, "overlap": 3, "all": 4, "average": 5, "min": 6, "max": 7,
               "nogap": 8}

CELL_DTYPE = np.dtype([("value_midx", "<u4"), ("value_sidx", "<u4"), ("gapm_idx", "<u4"),
                       ("gaps_idx", "<u4"), ("value", "<f4"), ("gapm_val", "<f4"), ("gaps_val", "<f4"),
                       ("gaps_max", "<u4")])

_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        vp = C.c_void_p
        L.so_cseq_new.restype = vp
        L.so_cseq_new.argtypes = [C.c_char_p]
        L.so_cseq_clone.restype = vp
        L.so_cseq_clone.argtypes = [vp]
        L.so_cseq_free.argtypes = [vp]
        L.so_cseq_clear.argtypes = [vp]
        L.so_cseq_append_str.argtypes = [vp, C.c_char_p]
        L.so_cseq_append_base.argtypes = [vp, C.c_uint32, C.POINTER(Log)]
        L.so_cseq_set_width.argtypes = [vp, C.c_uint32]
        L.so_cseq_reverse.argtypes = [vp]
        L.so_cseq_complement.argtypes = [vp]
        L.so_cseq_upper.argtypes = [vp]
        L.so_cseq_get_aligned.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
        L.so_cseq_get_bases.argtypes = [vp, C.c_char_p]
        L.so_cseq_fix_duplicate_positions.argtypes = [vp, C.POINTER(Log), C.c_int, C.c_int]
        L.so_cseq_size.restype = C.c_uint32
        L.so_cseq_size.argtypes = [vp]
        L.so_cseq_width.restype = C.c_uint32
        L.so_cseq_width.argtypes = [vp]
        L.so_cseq_data.restype = u32p
        L.so_cseq_data.argtypes = [vp]
        L.so_cseq_set_data.argtypes = [vp, u32p, C.c_uint32, C.c_uint32]
        L.so_kmer_trace.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_int, u8p, u32p]
        L.so_kmers.restype = C.c_uint32
        L.so_kmers.argtypes = [u32p, C.c_uint32, C.c_uint, C.c_uint, C.c_uint, C.c_int, u32p]
        L.so_vlimap_new.restype = vp
        L.so_vlimap_new.argtypes = [C.c_uint32]
        L.so_vlimap_free.argtypes = [vp]
        L.so_vlimap_push_back.argtypes = [vp, C.c_uint32]
        L.so_vlimap_increment.argtypes = [vp, i16p]
        L.so_vlimap_append.argtypes = [vp, vp]
        L.so_vlimap_invert.argtypes = [vp]
        L.so_vlimap_bytes.restype = C.c_size_t
        L.so_vlimap_bytes.argtypes = [vp, C.POINTER(u8p)]
        L.so_index_build.restype = vp
        L.so_index_build.argtypes = [C.POINTER(vp), C.c_uint32, C.c_uint, C.c_int]
        L.so_index_free.argtypes = [vp]
        L.so_index_size.restype = C.c_uint32
        L.so_index_size.argtypes = [vp]
        L.so_index_scores.argtypes = [vp, vp, i16p]
        L.so_index_find.restype = C.c_uint32
        L.so_index_find.argtypes = [vp, vp, C.c_uint32, u32p, f32p]
        L.so_turn_check.restype = C.c_int
        L.so_turn_check.argtypes = [vp, vp, C.c_int, f32p]
        L.so_index_csr.restype = C.c_uint64
        L.so_index_csr.argtypes = [vp, u32p, u32p]
        L.so_ff_opts_default.argtypes = [C.POINTER(FFOpts)]
        L.so_famfinder.restype = C.c_uint32
        L.so_famfinder.argtypes = [vp, C.POINTER(vp), vp, C.POINTER(FFOpts), u32p, f32p, C.c_uint32,
                                   C.POINTER(Log)]
        L.so_mseq_build.restype = C.POINTER(Graph)
        L.so_mseq_build.argtypes = [C.POINTER(vp), C.c_uint32, C.c_float]
        L.so_pseq_build.restype = C.POINTER(Graph)
        L.so_pseq_build.argtypes = [C.POINTER(vp), C.c_uint32]
        L.so_profile_comp.restype = C.c_float
        L.so_profile_comp.argtypes = [f32p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.so_graph_free.argtypes = [C.POINTER(Graph)]
        L.so_align_opts_default.argtypes = [C.POINTER(AlignOpts)]
        L.so_mesh_compute.argtypes = [C.POINTER(Graph), u32p, C.c_uint32, C.POINTER(AlignOpts), vp]
        L.so_backtrack.restype = C.c_float
        L.so_backtrack.argtypes = [C.POINTER(Graph), u32p, C.c_uint32, vp, C.POINTER(AlignOpts), vp,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(Log)]
        L.so_align.argtypes = [C.POINTER(vp), C.c_uint32, vp, C.POINTER(AlignOpts), vp,
                               C.POINTER(AlignResult), C.POINTER(Log)]
        L.so_score_op.restype = C.c_float
        L.so_score_op.argtypes = [C.c_int, C.c_float, C.c_uint32, C.c_int, C.c_float, C.c_int, C.c_int,
                                  C.c_float, C.c_float, C.c_float, C.c_float, f32p, C.c_uint32]
        L.so_bench_run.restype = C.c_double
        L.so_bench_run.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.c_uint32, C.POINTER(FFOpts),
                                   C.POINTER(AlignOpts), C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.so_log_init.argtypes = [C.POINTER(Log)]
        L.so_log_free.argtypes = [C.POINTER(Log)]
        L.so_index_write.argtypes = [vp, C.POINTER(C.c_char_p), C.c_char_p]
        L.so_index_read.restype = vp
        L.so_index_read.argtypes = [C.c_char_p, C.c_uint, C.c_int]
        L.so_compare_counts.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(MatchCounts)]
        L.so_compare_score.restype = C.c_float
        L.so_compare_score.argtypes = [C.POINTER(MatchCounts), C.c_int, C.c_int]
        L.so_compare.restype = C.c_float
        L.so_compare.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.so_search_opts_default.argtypes = [C.POINTER(SearchOpts)]
        L.so_search.restype = C.c_int
        L.so_search.argtypes = [vp, C.POINTER(vp), C.c_uint32, vp, C.POINTER(SearchOpts), u32p, f32p, C.c_uint32,
                                C.POINTER(Log)]
        cpp = C.POINTER(C.c_char_p)
        L.so_search_nearest.argtypes = [cpp, cpp, cpp, cpp, u32p, f32p, C.c_uint32, C.POINTER(Log)]
        L.so_search_lca.argtypes = [cpp, C.c_uint32, C.c_float, C.POINTER(Log)]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(_REF)
        vp = C.c_void_p
        R.ref_kmer_trace.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_int, u8p, u32p]
        R.ref_kmers.restype = C.c_uint32
        R.ref_kmers.argtypes = [u32p, C.c_uint32, C.c_uint, C.c_uint, C.c_uint, C.c_int, u32p]
        R.ref_pack.restype = C.c_uint32
        R.ref_pack.argtypes = [C.c_uint32, C.c_int]
        R.ref_vlimap_new.restype = vp
        R.ref_vlimap_new.argtypes = [C.c_uint32]
        R.ref_vlimap_free.argtypes = [vp]
        R.ref_vlimap_push_back.argtypes = [vp, C.c_uint32]
        R.ref_vlimap_increment.argtypes = [vp, i16p, C.c_uint32]
        R.ref_vlimap_append.argtypes = [vp, vp]
        R.ref_vlimap_invert.argtypes = [vp]
        R.ref_vlimap_size.restype = C.c_uint32
        R.ref_vlimap_size.argtypes = [vp]
        R.ref_vlimap_write.restype = C.c_uint32
        R.ref_vlimap_write.argtypes = [vp, u8p, C.c_uint32]
        R.ref_dag_build.restype = vp
        R.ref_dag_build.argtypes = [C.POINTER(u32p), u32p, C.c_uint32, C.c_uint32, C.c_float]
        R.ref_dag_free.argtypes = [vp]
        R.ref_dag_size.restype = C.c_uint32
        R.ref_dag_size.argtypes = [vp]
        R.ref_dag_dump.restype = C.c_uint32
        R.ref_dag_dump.argtypes = [vp, u32p, u32p, u8p, f32p, u32p, u32p, u32p, u32p, u32p, u32p]
        R.ref_score_op.restype = C.c_float
        R.ref_score_op.argtypes = [C.c_int, C.c_float, C.c_uint32, C.c_int, C.c_float, C.c_int, C.c_int,
                                   C.c_float, C.c_float, C.c_float, C.c_float, f32p, C.c_uint32]
        R.ref_mesh_compute_simple.argtypes = [vp, u32p, C.c_uint32, C.c_float, C.c_float, C.c_float,
                                              C.c_float, vp]
        _ref = R
    return _ref


def _p(arr, typ):
    return arr.ctypes.data_as(typ)


class Cseq:
    """Owning wrapper around so_cseq."""

    def __init__(self, name="", aligned=None, handle=None):
        L = lib()
        self.h = C.c_void_p(handle) if handle is not None else C.c_void_p(L.so_cseq_new(name.encode()))
        if aligned is not None:
            if L.so_cseq_append_str(self.h, aligned.encode()) != 0:
                raise ValueError("bad character")

    def __del__(self):
        try:
            lib().so_cseq_free(self.h)
        except Exception:
            pass

    @classmethod
    def from_packed(cls, name, ab, width):
        c = cls(name)
        ab = np.ascontiguousarray(ab, dtype=np.uint32)
        lib().so_cseq_set_data(c.h, _p(ab, u32p), len(ab), width)
        return c

    @property
    def size(self):
        return lib().so_cseq_size(self.h)

    @property
    def width(self):
        return lib().so_cseq_width(self.h)

    def packed(self):
        n = self.size
        if n == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(lib().so_cseq_data(self.h), shape=(n,)).copy()

    def aligned(self, nodots=False, dna=False):
        buf = C.create_string_buffer(self.width + 1)
        lib().so_cseq_get_aligned(self.h, int(nodots), int(dna), buf)
        return buf.value.decode()

    def bases(self):
        buf = C.create_string_buffer(self.size + 1)
        lib().so_cseq_get_bases(self.h, buf)
        return buf.value.decode()


def new_log():
    lg = Log()
    lib().so_log_init(C.byref(lg))
    return lg


def log_text(lg):
    s = lg.s.decode() if lg.s else ""
    return s


def handles(cseqs):
    arr = (C.c_void_p * len(cseqs))(*[c.h for c in cseqs])
    return arr


def ff_opts(**kw):
    o = FFOpts()
    lib().so_ff_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def align_opts(weights=None, **kw):
    o = AlignOpts()
    lib().so_align_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    if weights is not None:
        w = np.ascontiguousarray(weights, dtype=np.float32)
        o._keep = w
        o.weights = _p(w, f32p)
        o.n_weights = len(w)
    return o


def kmers(ab, k, p_len=0, p_val=0, unique=False):
    ab = np.ascontiguousarray(ab, dtype=np.uint32)
    out = np.zeros(len(ab) + 1, np.uint32)
    n = lib().so_kmers(_p(ab, u32p), len(ab), k, p_len, p_val, int(unique), _p(out, u32p))
    return out[:n].copy()


def kmer_trace(seq, k, p_len=0, p_val=0, unique=False):
    good = np.zeros(len(seq), np.uint8)
    val = np.zeros(len(seq), np.uint32)
    lib().so_kmer_trace(seq.encode(), k, p_len, p_val, int(unique), _p(good, u8p), _p(val, u32p))
    return good, val


class Index:
    def __init__(self, refs, k=10, nofast=False):
        self.refs = refs
        self._h = handles(refs)
        self.h = C.c_void_p(lib().so_index_build(self._h, len(refs), k, int(nofast)))
        self.k = k

    def __del__(self):
        try:
            lib().so_index_free(self.h)
        except Exception:
            pass

    def scores(self, q):
        s = np.zeros(len(self.refs), np.int16)
        lib().so_index_scores(self.h, q.h, _p(s, i16p))
        return s

    def find(self, q, mx):
        n = min(mx, len(self.refs))
        ids = np.zeros(max(n, 1), np.uint32)
        sc = np.zeros(max(n, 1), np.float32)
        r = lib().so_index_find(self.h, q.h, mx, _p(ids, u32p), _p(sc, f32p))
        return ids[:r].copy(), sc[:r].copy()

    def csr(self):
        nk = 1 << (2 * self.k)
        off = np.zeros(nk + 1, np.uint32)
        total = lib().so_index_csr(self.h, _p(off, u32p), None)
        ids = np.zeros(max(int(total), 1), np.uint32)
        lib().so_index_csr(self.h, _p(off, u32p), _p(ids, u32p))
        return off, ids[:total]

    def write_sidx(self, path, names=None):
        """The reference's on-disk index cache for this index (kmer_search::impl::store)."""
        names = names or ["ref%d" % i for i in range(len(self.refs))]
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        if lib().so_index_write(self.h, arr, path.encode()) != 0:
            raise IOError(path)

    @classmethod
    def read_sidx(cls, path, refs, k=10, nofast=False):
        """kmer_search::impl::try_load; None when the header does not fit."""
        h = lib().so_index_read(path.encode(), k, int(nofast))
        if not h:
            return None
        self = cls.__new__(cls)
        self.refs = refs
        self._h = handles(refs)
        self.h = C.c_void_p(h)
        self.k = k
        return self

    TURN_NAMES = ("none", "reversed", "complemented", "reversed and complemented")

    def turn_check(self, q, all_orientations):
        """famfinder::impl::turn_check: (orientation 0..3, the four top-1 scores)."""
        sc = np.zeros(4, np.float32)
        best = lib().so_turn_check(self.h, q.h, int(all_orientations), _p(sc, f32p))
        return best, sc

    def famfinder(self, q, opts=None):
        opts = opts or ff_opts()
        cap = len(self.refs)
        ids = np.zeros(max(cap, 1), np.uint32)
        sc = np.zeros(max(cap, 1), np.float32)
        lg = new_log()
        n = lib().so_famfinder(self.h, self._h, q.h, C.byref(opts), _p(ids, u32p), _p(sc, f32p), cap,
                               C.byref(lg))
        txt = log_text(lg)
        lib().so_log_free(C.byref(lg))
        return ids[:n].copy(), sc[:n].copy(), txt


def mseq_build(fam, weight=1.0):
    """Returns dict of numpy arrays describing the family DAG (or None if it would throw)."""
    h = handles(fam)
    g = lib().so_mseq_build(h, len(fam), weight)
    if not g:
        return None
    G = g.contents
    n = G.n

    def arr(p, cnt, dt):
        if cnt == 0:
            return np.zeros(0, dt)
        return np.ctypeslib.as_array(p, shape=(cnt,)).astype(dt).copy()
    pred_off = arr(G.pred_off, n + 1, np.uint32)
    succ_off = arr(G.succ_off, n + 1, np.uint32)
    d = dict(n=n, width=G.width, pos=arr(G.pos, n, np.uint32), mask=arr(G.mask, n, np.uint8),
             weight=arr(G.weight, n, np.float32), pred_off=pred_off,
             pred=arr(G.pred, int(pred_off[-1]) if n else 0, np.uint32), succ_off=succ_off,
             succ=arr(G.succ, int(succ_off[-1]) if n else 0, np.uint32),
             src=arr(G.src, G.n_src, np.uint32), snk=arr(G.snk, G.n_snk, np.uint32))
    lib().so_graph_free(g)
    return d


def pseq_build(fam):
    """The family as a profile (--fs-no-graph): dict(n, width, pos, prof[n, 6])."""
    g = lib().so_pseq_build(handles(fam), len(fam))
    G = g.contents
    n = G.n
    d = dict(n=n, width=G.width,
             pos=np.ctypeslib.as_array(G.pos, shape=(n,)).copy() if n else np.zeros(0, np.uint32),
             prof=np.ctypeslib.as_array(G.prof, shape=(n, 6)).copy() if n else np.zeros((0, 6), np.float32))
    lib().so_graph_free(g)
    return d


def profile_comp(prof, smask, match, mismatch, gap, gap_ext):
    """base_profile::comp of a column (6 floats; None: the base's own profile) with a query base."""
    p = None if prof is None else _p(np.ascontiguousarray(prof, dtype=np.float32), f32p)
    return np.float32(lib().so_profile_comp(p, smask, match, mismatch, gap, gap_ext))


def mesh_compute(fam, query, opts=None, weight=None):
    """Full cell plane (structured array [N, L]) for family x query."""
    opts = opts or align_opts()
    h = handles(fam)
    g = (lib().so_pseq_build(h, len(fam)) if opts.fs_no_graph else
         lib().so_mseq_build(h, len(fam), opts.fs_weight if weight is None else weight))
    n = g.contents.n
    q = query.packed()
    cells = np.zeros((n, len(q)), CELL_DTYPE)
    lib().so_mesh_compute(g, _p(q, u32p), len(q), C.byref(opts), cells.ctypes.data_as(C.c_void_p))
    lib().so_graph_free(g)
    return cells


def align(fam, query, opts=None):
    """Returns dict(status, head, tail, qual, score, cells, aligned, packed, width, log)."""
    opts = opts or align_opts()
    out = Cseq("out")
    res = AlignResult()
    lg = new_log()
    lib().so_align(handles(fam), len(fam), query.h, C.byref(opts), out.h, C.byref(res), C.byref(lg))
    txt = log_text(lg)
    lib().so_log_free(C.byref(lg))
    d = dict(status=res.status, head=res.head, tail=res.tail, qual=res.qual, score=np.float32(res.score),
             cells=res.cells, log=txt, packed=out.packed(), width=out.width, idty=np.float32(res.idty))
    d["aligned"] = out.aligned() if res.status in (0, 1) else None
    return d


def bench_run(index, queries, threads, ff=None, al=None, interleave=False):
    """Times the oracle's whole path over `queries` (list of Cseq) on `threads` host threads
    (interleave: new pages spread over all NUMA nodes, as `numactl --interleave=all`).
    Returns dict(seconds, cells, aligned, interleaved)."""
    ff = ff or ff_opts()
    al = al or align_opts()
    qh = handles(queries)
    cells, aligned = C.c_uint64(), C.c_uint32()
    L = lib()
    L.so_bench_mempolicy_interleave.argtypes = [C.c_int]
    il = bool(interleave) and L.so_bench_mempolicy_interleave(1) == 0
    try:
        sec = L.so_bench_run(index.h, index._h, qh, len(queries), C.byref(ff), C.byref(al), threads,
                             C.byref(cells), C.byref(aligned))
    finally:
        if il:
            L.so_bench_mempolicy_interleave(0)
    return dict(seconds=sec, cells=cells.value, aligned=aligned.value, interleaved=il)


# ---- section 8f-1: cseq_comparator + search_filter

def compare_counts(a, b, iupac="optimistic", filter_lc=False):
    """match_counter of traverse(a, b): (only_a_overhang, only_b_overhang, only_a, only_b, match, mismatch)."""
    m = MatchCounts()
    lib().so_compare_counts(a.h, b.h, IUPAC_RULES[iupac], int(filter_lc), C.byref(m))
    return (m.only_a_overhang, m.only_b_overhang, m.only_a, m.only_b, m.match, m.mismatch)


def compare(a, b, iupac="optimistic", dist="none", cover="query", filter_lc=False):
    return np.float32(lib().so_compare(a.h, b.h, IUPAC_RULES[iupac], DIST_RULES[dist], COVER_RULES[cover],
                                       int(filter_lc)))


def compare_score(counts, cover="query", dist="none"):
    m = MatchCounts(*[int(x) for x in counts])
    return np.float32(lib().so_compare_score(C.byref(m), COVER_RULES[cover], DIST_RULES[dist]))


def search_opts(**kw):
    o = SearchOpts()
    lib().so_search_opts_default(C.byref(o))
    for k, v in kw.items():
        if k == "iupac":
            v = IUPAC_RULES[v]
        elif k == "dist":
            v = DIST_RULES[v]
        elif k == "cover":
            v = COVER_RULES[v]
        setattr(o, k, v)
    return o


def search(index, query_aligned, opts=None):
    """search_filter::operator() up to the result vector: (ids, scores, log) best first; ids is None when
    the stage bails out (sequence too short)."""
    opts = opts or search_opts()
    cap = max(len(index.refs), 1)
    ids = np.zeros(cap, np.uint32)
    sc = np.zeros(cap, np.float32)
    lg = new_log()
    n = lib().so_search(index.h, index._h, len(index.refs), query_aligned.h, C.byref(opts), _p(ids, u32p),
                        _p(sc, f32p), cap, C.byref(lg))
    txt = log_text(lg)
    lib().so_log_free(C.byref(lg))
    if n < 0:
        return None, None, txt
    return ids[:n].copy(), sc[:n].copy(), txt


def _cstrs(strings):
    arr = (C.c_char_p * max(len(strings), 1))()
    for i, x in enumerate(strings):
        arr[i] = x.encode()
    return arr


def search_nearest(acc, version, start, stop, ids, scores):
    lg = new_log()
    ids = np.ascontiguousarray(ids, np.uint32)
    scores = np.ascontiguousarray(scores, np.float32)
    lib().so_search_nearest(_cstrs(acc), _cstrs(version), _cstrs(start), _cstrs(stop), _p(ids, u32p),
                            _p(scores, f32p), len(ids), C.byref(lg))
    txt = log_text(lg)
    lib().so_log_free(C.byref(lg))
    return txt


def search_lca(tax_of_results, quorum=0.7):
    lg = new_log()
    lib().so_search_lca(_cstrs(tax_of_results), len(tax_of_results), quorum, C.byref(lg))
    txt = log_text(lg)
    lib().so_log_free(C.byref(lg))
    return txt

