"""ctypes view of the C ABI in include/sina_hip.h (sina_amd/libsina_hip.so).

This is what a foreign host (SINA's C++ stages, or any other FFI) binds; the
Python layer adds nothing but argument marshalling.  The library is loaded from
the package directory (in-tree build) and loading fails loudly if it is missing:
there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SINA_HIP_LIB: the profiling build of the same library, tools only)
LIB_PATH = os.path.abspath(os.environ["SINA_HIP_LIB"]) if os.environ.get("SINA_HIP_LIB") else os.path.join(_HERE, "libsina_hip.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)
i16p = C.POINTER(C.c_int16)

# every extern "C" symbol include/sina_hip.h declares
ABI_SYMBOLS = [
    "sina_hip_abi_version", "sina_hip_last_error", "sina_hip_init", "sina_hip_fork", "sina_hip_prewarm", "sina_hip_destroy",
    "sina_hip_sync",
    "sina_hip_upload_refs", "sina_hip_build_index", "sina_hip_download_index", "sina_hip_upload_index", "sina_hip_store_view_get",
    "sina_hip_store_alloc_like", "sina_hip_kmer_topk", "sina_hip_kmer_scores", "sina_hip_compare",
    "sina_hip_align_params_default", "sina_hip_staged_out_pos", "sina_hip_align_graphs", "sina_hip_align_families",
    "sina_hip_debug_mesh", "sina_hip_debug_family_graph", "sina_hip_debug_dp_info", "sina_hip_debug_rgain", "sina_hip_get_stats",
]


class AlignParams(C.Structure):
    _fields_ = [("match_score", C.c_float), ("mismatch_score", C.c_float), ("gap_penalty", C.c_float),
                ("gap_ext_penalty", C.c_float), ("fs_weight", C.c_float), ("overhang", C.c_int32),
                ("lowercase", C.c_int32), ("insertion", C.c_int32), ("weights", f32p),
                ("n_weights", C.c_uint32), ("assemble", C.c_int32)]


class GraphBatch(C.Structure):
    _fields_ = [("nq", C.c_uint32), ("node_off", u64p), ("edge_off", u64p), ("node_pos", u32p),
                ("node_mask", u8p), ("node_weight", f32p), ("pred_off", u32p), ("pred", u32p),
                ("succ_minpos", u32p), ("width", C.c_uint32), ("node_score16", f32p), ("self_score16", f32p)]


class AlignOut(C.Structure):
    _fields_ = [("end_m", C.c_uint32), ("end_s", C.c_uint32), ("raw", C.c_float), ("sum_weight", C.c_float),
                ("aligned_bases", C.c_int32), ("cutoff_head", C.c_int32), ("cutoff_tail", C.c_int32),
                ("n_out", C.c_uint32), ("status", C.c_int32), ("assembled", C.c_uint32),
                ("nast_total", C.c_uint32), ("nast_longest", C.c_uint32), ("nast_last_run", C.c_uint32)]


ALIGN_OUT_DTYPE = np.dtype([("end_m", "<u4"), ("end_s", "<u4"), ("raw", "<f4"), ("sum_weight", "<f4"),
                            ("aligned_bases", "<i4"), ("cutoff_head", "<i4"), ("cutoff_tail", "<i4"),
                            ("n_out", "<u4"), ("status", "<i4"), ("assembled", "<u4"), ("nast_total", "<u4"),
                            ("nast_longest", "<u4"), ("nast_last_run", "<u4")])


class StoreView(C.Structure):
    _fields_ = [("ref_ab", C.c_void_p), ("ref_ab_bytes", C.c_uint64), ("ref_off", C.c_void_p),
                ("ref_off_bytes", C.c_uint64), ("idx_offsets", C.c_void_p), ("idx_offsets_bytes", C.c_uint64),
                ("idx_ids", C.c_void_p), ("idx_ids_bytes", C.c_uint64), ("n_refs", C.c_uint32),
                ("width", C.c_uint32), ("k", C.c_uint32), ("nofast", C.c_uint32), ("n_postings", C.c_uint64),
                ("total_bases", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("dp_ms", C.c_double), ("backtrack_ms", C.c_double), ("graph_ms", C.c_double),
                ("kmer_count_ms", C.c_double), ("kmer_select_ms", C.c_double), ("dp_cells", C.c_uint64),
                ("postings", C.c_uint64), ("dp_launches", C.c_uint32), ("kmer_launches", C.c_uint32),
                ("compare_ms", C.c_double), ("compare_bases", C.c_uint64), ("compare_launches", C.c_uint32),
                ("n_dense_lists", C.c_uint32), ("dp_busy_ms", C.c_double), ("dags_built", C.c_uint64),
                ("dags_used", C.c_uint64), ("dp_rows", C.c_uint64), ("dp_rows_swept", C.c_uint64),
                ("dp_cells_swept", C.c_uint64), ("dp_queries_pruned", C.c_uint64), ("dp_second_attempts", C.c_uint64),
                ("dp_full_sweeps", C.c_uint64), ("dp_prune_rho", C.c_double), ("graph_bytes", C.c_uint64),
                ("graph_launches", C.c_uint32), ("kmer_queries", C.c_uint32), ("scout_ms", C.c_double),
                ("scout_launches", C.c_uint32), ("pad_", C.c_uint32)]


class DpInfo(C.Structure):
    _fields_ = [("end_m", C.c_uint32), ("end_s", C.c_uint32), ("raw", C.c_float), ("status", C.c_int32),
                ("rows_swept", C.c_uint32), ("cells_swept", C.c_uint32), ("attempts", C.c_uint32),
                ("gain0", C.c_float), ("ubound", C.c_float), ("prune_step", C.c_uint32), ("prune_gmin", C.c_uint32),
                ("scout", C.c_float)]


_lib = None


def load():
    """Loads libsina_hip.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("sina_amd/libsina_hip.so is missing: run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.sina_hip_abi_version.restype = C.c_int
    L.sina_hip_last_error.restype = C.c_char_p
    L.sina_hip_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.sina_hip_fork.argtypes = [vp, C.POINTER(vp)]
    L.sina_hip_prewarm.argtypes = [vp, C.c_int]
    L.sina_hip_destroy.argtypes = [vp]
    L.sina_hip_destroy.restype = None
    L.sina_hip_sync.argtypes = [vp]
    L.sina_hip_staged_out_pos.restype = C.POINTER(C.c_uint32)
    L.sina_hip_staged_out_pos.argtypes = [vp]
    L.sina_hip_upload_refs.argtypes = [vp, u32p, u64p, C.c_uint32, C.c_uint32]
    L.sina_hip_build_index.argtypes = [vp, C.c_uint, C.c_int]
    L.sina_hip_upload_index.argtypes = [vp, C.c_uint, C.c_int, u32p, u32p, C.c_uint64]
    L.sina_hip_download_index.argtypes = [vp, u32p, u32p]
    L.sina_hip_store_view_get.argtypes = [vp, C.POINTER(StoreView)]
    L.sina_hip_store_alloc_like.argtypes = [vp, C.POINTER(StoreView)]
    L.sina_hip_kmer_topk.argtypes = [vp, u8p, u64p, C.c_uint32, C.c_uint32, u32p, f32p, u32p]
    L.sina_hip_kmer_scores.argtypes = [vp, u8p, C.c_uint32, i16p]
    L.sina_hip_compare.argtypes = [vp, u32p, u64p, C.c_uint32, u32p, u64p, C.c_int, C.c_int, C.c_void_p]
    L.sina_hip_align_params_default.argtypes = [C.POINTER(AlignParams)]
    L.sina_hip_align_params_default.restype = None
    L.sina_hip_align_graphs.argtypes = [vp, C.POINTER(GraphBatch), u8p, u64p, C.POINTER(AlignParams),
                                        C.POINTER(AlignOut), u32p]
    L.sina_hip_align_families.argtypes = [vp, u32p, u64p, C.c_uint32, u8p, u64p, C.POINTER(AlignParams),
                                          C.POINTER(AlignOut), u32p]
    L.sina_hip_debug_mesh.argtypes = [vp, C.POINTER(GraphBatch), u8p, C.c_uint32, C.POINTER(AlignParams),
                                      u32p, u32p, f32p, C.c_int]
    L.sina_hip_debug_family_graph.argtypes = [vp, u32p, C.c_uint32, C.c_float, C.c_uint32, u32p, u32p, u32p, u8p,
                                              f32p, u32p, u32p, u32p, u8p, u32p, C.c_uint32, C.c_uint32]
    L.sina_hip_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.sina_hip_debug_dp_info.argtypes = [vp, C.c_uint32, C.POINTER(DpInfo)]
    L.sina_hip_debug_rgain.argtypes = [vp, C.c_uint32, u32p, u32p]
    _lib = L
    return L


class SinaHipError(RuntimeError):
    pass


def _ptr(a, t):
    return a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Context:
    """Owns one sina_hip_ctx (one GPU, one stream)."""

    def __init__(self, device=0, _parent=None):
        self.L = load()
        self.h = C.c_void_p()
        if _parent is None:
            self._check(self.L.sina_hip_init(device, C.byref(self.h)))
        else:
            self._check(self.L.sina_hip_fork(_parent.h, C.byref(self.h)))
        self._parent = _parent  # keeps the owner of the store alive
        self.device = device
        self.n_refs = 0 if _parent is None else _parent.n_refs

    def fork(self):
        """A context with its own stream and scratch that shares this one's store and index."""
        return Context(self.device, _parent=self)

    def _check(self, rc):
        if rc != 0:
            raise SinaHipError(self.L.sina_hip_last_error().decode())

    def close(self):
        if self.h:
            self.L.sina_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- store
    def upload_refs(self, ab, off, width):
        ab = _c(ab, np.uint32)
        off = _c(off, np.uint64)
        self._keep = (ab, off)
        self.n_refs = len(off) - 1
        self._check(self.L.sina_hip_upload_refs(self.h, _ptr(ab, u32p), _ptr(off, u64p), self.n_refs, width))

    def build_index(self, k=10, nofast=False):
        self._check(self.L.sina_hip_build_index(self.h, k, int(nofast)))

    def upload_index(self, k, nofast, offsets, ids):
        offsets = _c(offsets, np.uint32)
        ids = _c(ids, np.uint32)
        self._check(self.L.sina_hip_upload_index(self.h, k, int(nofast), _ptr(offsets, u32p), _ptr(ids, u32p),
                                                 len(ids)))

    def download_index(self):
        """The device index as CSR: (offsets u32[4^k+1], ids u32[n_postings])."""
        v = self.store_view()
        off = np.zeros((1 << (2 * v.k)) + 1, np.uint32)
        ids = np.zeros(max(int(v.n_postings), 1), np.uint32)
        self._check(self.L.sina_hip_download_index(self.h, _ptr(off, u32p), _ptr(ids, u32p)))
        return off, ids[:int(v.n_postings)]

    def store_view(self):
        v = StoreView()
        self._check(self.L.sina_hip_store_view_get(self.h, C.byref(v)))
        return v

    def store_alloc_like(self, view):
        self._check(self.L.sina_hip_store_alloc_like(self.h, C.byref(view)))
        self.n_refs = view.n_refs
        return view

    # ---- k-mer search
    def kmer_topk(self, qmask, qoff, mx):
        qmask = _c(qmask, np.uint8)
        qoff = _c(qoff, np.uint64)
        nq = len(qoff) - 1
        mx_eff = max(1, min(mx, self.n_refs))
        ids = np.zeros((nq, mx_eff), np.uint32)
        sc = np.zeros((nq, mx_eff), np.float32)
        n = np.zeros(nq, np.uint32)
        self._check(self.L.sina_hip_kmer_topk(self.h, _ptr(qmask, u8p), _ptr(qoff, u64p), nq, mx, _ptr(ids, u32p),
                                              _ptr(sc, f32p), _ptr(n, u32p)))
        return ids, sc, n

    def kmer_scores(self, qmask):
        qmask = _c(qmask, np.uint8)
        s = np.zeros(self.n_refs, np.int16)
        self._check(self.L.sina_hip_kmer_scores(self.h, _ptr(qmask, u8p), len(qmask), _ptr(s, i16p)))
        return s

    # ---- alignment
    @staticmethod
    def params(weights=None, **kw):
        p = AlignParams()
        load().sina_hip_align_params_default(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        if weights is not None:
            w = _c(weights, np.float32)
            p._keep = w
            p.weights = _ptr(w, f32p)
            p.n_weights = len(w)
        return p

    @staticmethod
    def graph_batch(graphs, width):
        """graphs: list of dicts with pos, mask, weight, pred_off, pred[, succ_minpos]."""
        nq = len(graphs)
        node_off = np.zeros(nq + 1, np.uint64)
        edge_off = np.zeros(nq + 1, np.uint64)
        for i, g in enumerate(graphs):
            node_off[i + 1] = node_off[i] + len(g["pos"])
            edge_off[i + 1] = edge_off[i] + len(g["pred"])
        cat = lambda key, dt: _c(np.concatenate([np.asarray(g[key]) for g in graphs]) if nq else [], dt)
        arrs = dict(node_off=node_off, edge_off=edge_off, node_pos=cat("pos", np.uint32),
                    node_mask=cat("mask", np.uint8), node_weight=cat("weight", np.float32),
                    pred_off=cat("pred_off", np.uint32), pred=cat("pred", np.uint32))
        if nq and all("succ_minpos" in g for g in graphs):
            arrs["succ_minpos"] = cat("succ_minpos", np.uint32)
        gb = GraphBatch()
        gb.nq = nq
        gb.node_off = _ptr(arrs["node_off"], u64p)
        gb.edge_off = _ptr(arrs["edge_off"], u64p)
        gb.node_pos = _ptr(arrs["node_pos"], u32p)
        gb.node_mask = _ptr(arrs["node_mask"], u8p)
        gb.node_weight = _ptr(arrs["node_weight"], f32p)
        gb.pred_off = _ptr(arrs["pred_off"], u32p)
        gb.pred = _ptr(arrs["pred"], u32p)
        gb.succ_minpos = _ptr(arrs["succ_minpos"], u32p) if "succ_minpos" in arrs else None
        gb.width = width
        gb._keep = arrs
        return gb

    def align_graphs(self, gb, qmask, qoff, params=None):
        params = params or self.params()
        qmask = _c(qmask, np.uint8)
        qoff = _c(qoff, np.uint64)
        out = np.zeros(gb.nq, ALIGN_OUT_DTYPE)
        pos = np.zeros(max(len(qmask), 1), np.uint32)
        self._check(self.L.sina_hip_align_graphs(self.h, C.byref(gb), _ptr(qmask, u8p), _ptr(qoff, u64p),
                                                 C.byref(params), out.ctypes.data_as(C.POINTER(AlignOut)),
                                                 _ptr(pos, u32p)))
        return out, pos

    def compare(self, q_ab, q_off, cand_ids, cand_off, iupac=0, filter_lc=False):
        """Search-stage comparison: int32 [n candidates][6] = only_a_overhang, only_b_overhang, only_a,
        only_b, match, mismatch of every (query, candidate) pair."""
        q_ab = _c(q_ab, np.uint32)
        q_off = _c(q_off, np.uint64)
        cand_ids = _c(cand_ids, np.uint32)
        cand_off = _c(cand_off, np.uint64)
        out = np.zeros((max(len(cand_ids), 1), 6), np.int32)
        self._check(self.L.sina_hip_compare(self.h, _ptr(q_ab, u32p), _ptr(q_off, u64p), len(q_off) - 1,
                                            _ptr(cand_ids, u32p), _ptr(cand_off, u64p), int(iupac),
                                            int(filter_lc), out.ctypes.data_as(C.c_void_p)))
        return out[:len(cand_ids)]

    def align_families(self, fam_ids, fam_off, qmask, qoff, params=None):
        params = params or self.params()
        fam_ids = _c(fam_ids, np.uint32)
        fam_off = _c(fam_off, np.uint64)
        qmask = _c(qmask, np.uint8)
        qoff = _c(qoff, np.uint64)
        nq = len(qoff) - 1
        out = np.zeros(nq, ALIGN_OUT_DTYPE)
        pos = np.zeros(max(len(qmask), 1), np.uint32)
        self._check(self.L.sina_hip_align_families(self.h, _ptr(fam_ids, u32p), _ptr(fam_off, u64p), nq,
                                                   _ptr(qmask, u8p), _ptr(qoff, u64p), C.byref(params),
                                                   out.ctypes.data_as(C.POINTER(AlignOut)), _ptr(pos, u32p)))
        return out, pos

    def debug_mesh(self, gb, qmask, params=None, want_value=True, prune=False):
        """DP planes of ONE query.  prune=False (default): every row of every strip is swept, the planes are
        the reference's cell for cell; prune=True: the production launch with its certified row skip -- cells the
        kernel proved irrelevant hold whatever it left there (see dp_info / rgain for the bound it used)."""
        params = params or self.params()
        qmask = _c(qmask, np.uint8)
        n = int(gb._keep["node_off"][1])
        L = len(qmask)
        vm = np.zeros((n, L), np.uint32)
        vs = np.zeros((n, L), np.uint32)
        val = np.zeros((n, L), np.float32) if want_value else None
        self._check(self.L.sina_hip_debug_mesh(self.h, C.byref(gb), _ptr(qmask, u8p), L, C.byref(params),
                                               _ptr(vm, u32p), _ptr(vs, u32p),
                                               _ptr(val, f32p) if want_value else None, 1 if prune else 0))
        return vm, vs, val

    def debug_family_graph(self, fam_ids, fs_weight=1.0, ring_depth=4):
        fam_ids = _c(fam_ids, np.uint32)
        cap_n, cap_e = 65536, 400000
        nn, ne = C.c_uint32(), C.c_uint32()
        pos = np.zeros(cap_n, np.uint32)
        mask = np.zeros(cap_n, np.uint8)
        w = np.zeros(cap_n, np.float32)
        poff = np.zeros(cap_n + 1, np.uint32)
        pred = np.zeros(cap_e, np.uint32)
        smin = np.zeros(cap_n, np.uint32)
        sink = np.zeros(cap_n, np.uint8)
        spill = np.zeros(cap_n, np.uint32)
        self._check(self.L.sina_hip_debug_family_graph(self.h, _ptr(fam_ids, u32p), len(fam_ids), fs_weight,
                                                       ring_depth, C.byref(nn), C.byref(ne), _ptr(pos, u32p),
                                                       _ptr(mask, u8p), _ptr(w, f32p), _ptr(poff, u32p),
                                                       _ptr(pred, u32p), _ptr(smin, u32p), _ptr(sink, u8p),
                                                       _ptr(spill, u32p), cap_n, cap_e))
        n, e = nn.value, ne.value
        return dict(n=n, pos=pos[:n].copy(), mask=mask[:n].copy(), weight=w[:n].copy(), pred_off=poff[:n + 1].copy(),
                    pred=pred[:e].copy(), succ_minpos=smin[:n].copy(), sink=sink[:n].copy(), spill=spill[:n].copy())

    def dp_info(self, q=0):
        """What the DP kernel reported for query q of this context's last launch (row-skip test hook)."""
        d = DpInfo()
        self._check(self.L.sina_hip_debug_dp_info(self.h, q, C.byref(d)))
        return {f[0]: getattr(d, f[0]) for f in DpInfo._fields_}

    def rgain(self, n):
        """First n entries of the per-node row-skip bound the last launch / debug_family_graph left on the device:
        (R(m) in units of 1/64, C(m) = occupied columns right of the node's)."""
        out = np.zeros(max(n, 1), np.uint32)
        cols = np.zeros(max(n, 1), np.uint32)
        self._check(self.L.sina_hip_debug_rgain(self.h, n, _ptr(out, u32p), _ptr(cols, u32p)))
        return out[:n], cols[:n]

    def stats(self):
        s = Stats()
        self._check(self.L.sina_hip_get_stats(self.h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in Stats._fields_}
