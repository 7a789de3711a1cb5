"""Python driver over the C++ stage mirror (sina_amd/libsina_host.so).

famfinder -> aligner exactly as SINA wires them (tray in, tray out), with the
k-mer search and the mesh DP on the GPU through the C ABI.  Used by the tests and
bench.py; fails loudly when the native libraries are missing.
"""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libsina_host.so")
_host = None


def load_host():
    global _host
    if _host is not None:
        return _host
    capi.load()  # libsina_hip.so first (RPATH $ORIGIN resolves it as well)
    if not os.path.exists(HOST_LIB_PATH):
        raise RuntimeError("sina_amd/libsina_host.so is missing: build it with __graft_entry__.build()")
    H = C.CDLL(HOST_LIB_PATH)
    vp = C.c_void_p
    H.sina_host_last_error.restype = C.c_char_p
    H.sina_host_store_from_packed.argtypes = [C.c_char_p, capi.u32p, capi.u64p, C.c_uint32, C.c_uint32, C.c_int]
    H.sina_host_store_close.argtypes = [C.c_char_p]
    H.sina_host_add_filter.argtypes = [C.c_char_p, C.c_char_p, capi.f32p, C.c_uint32]
    H.sina_host_set_option.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    H.sina_host_pipeline_create.restype = vp
    H.sina_host_pipeline_destroy.argtypes = [vp]
    H.sina_host_pipeline_destroy.restype = None
    H.sina_host_pipeline_run.argtypes = [vp, capi.u8p, capi.u64p, C.c_uint32, C.c_uint32, C.c_uint32]
    H.sina_host_pipeline_run_single_trays.argtypes = [vp, capi.u8p, capi.u64p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                      C.c_uint32, C.c_int32, capi.u8p, C.c_char_p, C.c_uint32]
    H.sina_host_result.argtypes = [vp, C.c_uint32] + [C.POINTER(C.c_int)] * 4 + [capi.u32p, capi.u32p]
    H.sina_host_result_bases.restype = capi.u32p
    H.sina_host_result_bases.argtypes = [vp, C.c_uint32]
    H.sina_host_result_log.restype = C.c_char_p
    H.sina_host_result_log.argtypes = [vp, C.c_uint32]
    H.sina_host_result_family.restype = C.c_char_p
    H.sina_host_result_family.argtypes = [vp, C.c_uint32]
    H.sina_host_timings.argtypes = [vp] + [C.POINTER(C.c_double)] * 3
    H.sina_host_timings.restype = None
    H.sina_host_profile.restype = C.c_char_p
    H.sina_host_profile.argtypes = [C.c_int]
    H.sina_host_build_graph.argtypes = [C.c_char_p, capi.u32p, C.c_uint32, C.c_float, capi.u32p, capi.u32p,
                                        capi.u32p, capi.u8p, capi.f32p, capi.u32p, capi.u32p, capi.u32p,
                                        C.c_uint32, C.c_uint32]
    H.sina_host_cseq_op.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_int, C.c_char_p, C.c_uint32, capi.u32p,
                                    capi.u32p]
    H.sina_host_fix_duplicates.argtypes = [capi.u32p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_char_p,
                                           C.c_uint32]
    H.sina_host_store_ctx.restype = vp
    H.sina_host_store_ctx.argtypes = [C.c_char_p]
    H.sina_host_store_build_index.argtypes = [C.c_char_p, C.c_uint, C.c_int]
    H.sina_host_store_index_ready.argtypes = [C.c_char_p, C.c_uint, C.c_int]
    H.sina_host_store_set_attr.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p]
    H.sina_host_sidx_load.argtypes = [C.c_char_p, C.c_uint, C.c_int, capi.u32p, capi.u32p, capi.u32p, C.c_uint64,
                                      C.POINTER(C.c_uint64)]
    H.sina_host_sidx_store.argtypes = [C.c_char_p, C.c_uint, C.c_int, C.c_uint32, capi.u32p, capi.u32p, C.c_uint64]
    H.sina_host_store_open.argtypes = [C.c_char_p, C.c_int]
    H.sina_host_build_profile.argtypes = [C.c_char_p, capi.u32p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float,
                                          capi.u32p, capi.u32p, capi.f32p, capi.f32p, C.c_uint32]
    H.sina_host_store_open_arb_order.argtypes = [C.c_char_p, C.c_int]
    H.sina_host_reference_order.restype = C.c_uint32
    H.sina_host_reference_order.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, capi.u32p, capi.u64p, capi.u64p]
    H.sina_host_store_name.restype = C.c_char_p
    H.sina_host_store_name.argtypes = [C.c_char_p, C.c_uint32]
    H.sina_host_store_expect_broadcast.argtypes = [C.c_char_p]
    H.sina_host_run_fasta.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint32,
                                      C.POINTER(C.c_double)]
    H.sina_host_run_fasta_serial.argtypes = H.sina_host_run_fasta.argtypes
    H.sina_host_fasta_roundtrip.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    H.sina_host_store_index_origin.restype = C.c_char_p
    H.sina_host_store_index_origin.argtypes = [C.c_char_p]
    H.sina_host_compare.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    H.sina_host_pipeline_create_search.restype = vp
    H.sina_host_result_search.argtypes = [vp, C.c_uint32, capi.u32p, capi.f32p, C.c_uint32]
    H.sina_host_result_attr.restype = C.c_char_p
    H.sina_host_result_attr.argtypes = [vp, C.c_uint32, C.c_char_p]
    H.sina_host_search_seconds.restype = C.c_double
    H.sina_host_result_idty.restype = C.c_float
    H.sina_host_result_idty.argtypes = [vp, C.c_uint32]
    H.sina_host_search_seconds.argtypes = [vp]
    _host = H
    return H


class HostError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise HostError(load_host().sina_host_last_error().decode())


def reference_order(names):
    """(order, hashes, bucket_count): order[i] = position in `names` (database order) of the sequence the
    reference gives id i -- the walk of its unordered_map<string, ..., boost::hash<string>>."""
    H = load_host()
    n = len(names)
    arr = (C.c_char_p * max(n, 1))(*[s.encode() for s in names])
    order = np.zeros(max(n, 1), np.uint32)
    hashes = np.zeros(max(n, 1), np.uint64)
    buckets = np.zeros(1, np.uint64)
    m = H.sina_host_reference_order(arr, n, order.ctypes.data_as(capi.u32p), hashes.ctypes.data_as(capi.u64p),
                                    buckets.ctypes.data_as(capi.u64p))
    return order[:m].copy(), hashes[:n].copy(), int(buckets[0])


class Store:
    """A reference store registered under `key` (what SINA calls --db)."""

    def __init__(self, key, refs, device=0, upload=True):
        """upload=False: a rank other than 0 of a multi-GPU run -- the host side (the cseq objects the
        results point into) is built here, the device copy arrives by broadcast (dist.broadcast_device_index)."""
        self.H = load_host()
        self.key = key
        ab = np.ascontiguousarray(refs.ab, np.uint32)
        off = np.ascontiguousarray(refs.off, np.uint64)
        _chk(self.H.sina_host_store_from_packed(key.encode(), ab.ctypes.data_as(capi.u32p),
                                                off.ctypes.data_as(capi.u64p), refs.n, refs.width, device))
        if not upload:
            _chk(self.H.sina_host_store_expect_broadcast(key.encode()))

    @classmethod
    def open(cls, path, device=0, id_order="file"):
        """An aligned-FASTA database file; its k-mer index is cached beside it as <name>.sidx in the
        reference's own format (kmer_search.cpp:213-243).  id_order="arb": number the sequences as the
        reference numbers an ARB database's (host/id_order.h; query_arb.cpp:160,732-739)."""
        self = cls.__new__(cls)
        self.H = load_host()
        self.key = path
        if id_order not in ("file", "arb"):
            raise ValueError("id_order: 'file' or 'arb'")
        _chk((self.H.sina_host_store_open_arb_order if id_order == "arb" else self.H.sina_host_store_open)(
            path.encode(), device))
        return self

    def name(self, i):
        return self.H.sina_host_store_name(self.key.encode(), i).decode()

    def index_origin(self):
        return self.H.sina_host_store_index_origin(self.key.encode()).decode()

    def add_filter(self, name, weights):
        w = np.ascontiguousarray(weights, np.float32)
        _chk(self.H.sina_host_add_filter(self.key.encode(), name.encode(), w.ctypes.data_as(capi.f32p), len(w)))

    def build_profile(self, ids, match, mismatch, gap, gap_ext, cap=200000):
        """--fs-no-graph: (columns, match-term table [n, 16], self table [16]) of a family as the aligner builds them."""
        ids = np.ascontiguousarray(ids, np.uint32)
        nn = C.c_uint32()
        pos = np.zeros(cap, np.uint32)
        sc = np.zeros(16 * cap, np.float32)
        own = np.zeros(16, np.float32)
        _chk(self.H.sina_host_build_profile(self.key.encode(), ids.ctypes.data_as(capi.u32p), len(ids), match, mismatch,
                                            gap, gap_ext, C.byref(nn), pos.ctypes.data_as(capi.u32p),
                                            sc.ctypes.data_as(capi.f32p), own.ctypes.data_as(capi.f32p), cap))
        n = nn.value
        return pos[:n].copy(), sc[:16 * n].reshape(n, 16).copy(), own

    def build_graph(self, ids, fs_weight=1.0):
        ids = np.ascontiguousarray(ids, np.uint32)
        cap_n, cap_e = 200000, 400000
        nn, ne = C.c_uint32(), C.c_uint32()
        pos = np.zeros(cap_n, np.uint32)
        mask = np.zeros(cap_n, np.uint8)
        w = np.zeros(cap_n, np.float32)
        poff = np.zeros(cap_n + 1, np.uint32)
        pred = np.zeros(cap_e, np.uint32)
        smin = np.zeros(cap_n, np.uint32)
        _chk(self.H.sina_host_build_graph(self.key.encode(), ids.ctypes.data_as(capi.u32p), len(ids), fs_weight,
                                          C.byref(nn), C.byref(ne), pos.ctypes.data_as(capi.u32p),
                                          mask.ctypes.data_as(capi.u8p), w.ctypes.data_as(capi.f32p),
                                          poff.ctypes.data_as(capi.u32p), pred.ctypes.data_as(capi.u32p),
                                          smin.ctypes.data_as(capi.u32p), cap_n, cap_e))
        n, e = nn.value, ne.value
        return dict(n=n, pos=pos[:n].copy(), mask=mask[:n].copy(), weight=w[:n].copy(),
                    pred_off=poff[:n + 1].copy(), pred=pred[:e].copy(), succ_minpos=smin[:n].copy())

    def ctx_handle(self):
        """Raw sina_hip_ctx* of this store (device context with the references uploaded)."""
        h = self.H.sina_host_store_ctx(self.key.encode())
        if not h:
            raise HostError(self.H.sina_host_last_error().decode())
        return C.c_void_p(h)

    def set_attr(self, ref_id, field, value):
        """A database field of one reference (version, start, stop, a taxonomy path ...)."""
        _chk(self.H.sina_host_store_set_attr(self.key.encode(), ref_id, field.encode(), value.encode()))

    def build_index(self, k=10, nofast=False):
        _chk(self.H.sina_host_store_build_index(self.key.encode(), k, int(nofast)))

    def index_ready(self, k=10, nofast=False):
        _chk(self.H.sina_host_store_index_ready(self.key.encode(), k, int(nofast)))

    def stats(self):
        s = capi.Stats()
        if capi.load().sina_hip_get_stats(self.ctx_handle(), C.byref(s)) != 0:
            raise HostError("get_stats failed")
        return {f[0]: getattr(s, f[0]) for f in capi.Stats._fields_}

    def close(self):
        self.H.sina_host_store_close(self.key.encode())


class Pipeline:
    """famfinder + aligner (+ search_filter when `search` is given, as `sina --search`) over one Store.
    Options use SINA's command-line names."""

    def __init__(self, store, famfinder=None, aligner=None, host_threads=None, search=None, dedup=True):
        self.H = load_host()
        self.store = store
        self.H.sina_host_reset_options()
        self._set("host", "dedup", bool(dedup))  # repeated queries of a batch go to the device once
        self._set("famfinder", "db", store.key)
        self._set("aligner", "db", store.key)
        for k, v in (famfinder or {}).items():
            self._set("famfinder", k, v)
        for k, v in (aligner or {}).items():
            self._set("aligner", k, v)
        if host_threads:
            self._set("host", "threads", host_threads)
        self.has_search = search is not None
        if self.has_search:
            self._set("search", "search-db", store.key)
            for k, v in search.items():
                self._set("search", k, v)
            self.h = self.H.sina_host_pipeline_create_search()
        else:
            self.h = self.H.sina_host_pipeline_create()
        if not self.h:
            raise HostError(self.H.sina_host_last_error().decode())

    def _set(self, stage, name, value):
        if isinstance(value, bool):
            value = "1" if value else "0"
        _chk(self.H.sina_host_set_option(stage.encode(), name.encode(), str(value).encode()))

    def run(self, qmask, qoff, batch=1024, inflight=2):
        qmask = np.ascontiguousarray(qmask, np.uint8)
        qoff = np.ascontiguousarray(qoff, np.uint64)
        self.nq = len(qoff) - 1
        _chk(self.H.sina_host_pipeline_run(self.h, qmask.ctypes.data_as(capi.u8p), qoff.ctypes.data_as(capi.u64p),
                                           self.nq, batch, inflight))
        return self.timings()

    def run_single_trays(self, qmask, qoff, threads=32, max_batch=1024, linger_us=300, poison=-1):
        """The boundary as INTEGRATION.md binds it: `threads` concurrent callers push ONE tray at a time through
        sina::batched<famfinder> -> batched<aligner> (-> batched<search_filter>), as SINA's TBB nodes would
        (src/sina.cpp:497-519).  poison: index of a query whose family gets a sequence that is not of the store
        before the aligner sees it (the stage throws for the batch it travels in).
        Returns (failed flags per query, first exception text)."""
        qmask = np.ascontiguousarray(qmask, np.uint8)
        qoff = np.ascontiguousarray(qoff, np.uint64)
        self.nq = len(qoff) - 1
        failed = np.zeros(max(self.nq, 1), np.uint8)
        err = C.create_string_buffer(512)
        _chk(self.H.sina_host_pipeline_run_single_trays(self.h, qmask.ctypes.data_as(capi.u8p),
                                                        qoff.ctypes.data_as(capi.u64p), self.nq, threads, max_batch,
                                                        linger_us, poison, failed.ctypes.data_as(capi.u8p), err, 512))
        return failed[:self.nq].astype(bool), err.value.decode()

    def profile(self, reset=True):
        """Per-phase host wall times (needs SINA_HOST_PROFILE=1 in the environment)."""
        return self.H.sina_host_profile(int(reset)).decode()

    def timings(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.H.sina_host_timings(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(wall_s=a.value, famfinder_s=b.value, aligner_s=c.value)

    def result(self, q):
        st, hd, tl, ql = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        w, n = C.c_uint32(), C.c_uint32()
        _chk(self.H.sina_host_result(self.h, q, C.byref(st), C.byref(hd), C.byref(tl), C.byref(ql), C.byref(w),
                                     C.byref(n)))
        ab = np.ctypeslib.as_array(self.H.sina_host_result_bases(self.h, q), shape=(n.value,)).copy() \
            if n.value else np.zeros(0, np.uint32)
        d = dict(status=st.value, head=hd.value, tail=tl.value, qual=ql.value, width=w.value, packed=ab,
                 log=self.H.sina_host_result_log(self.h, q).decode(),
                 family=self.H.sina_host_result_family(self.h, q).decode(),
                 idty=np.float32(self.H.sina_host_result_idty(self.h, q)))
        if self.has_search:
            ids = np.zeros(4096, np.uint32)
            sc = np.zeros(4096, np.float32)
            n = self.H.sina_host_result_search(self.h, q, ids.ctypes.data_as(capi.u32p),
                                               sc.ctypes.data_as(capi.f32p), len(ids))
            d["search_ids"] = ids[:n].copy() if n >= 0 else None
            d["search_scores"] = sc[:n].copy() if n >= 0 else None
        return d

    def attr(self, q, name):
        """String attribute the search stage set on query q's sequence (nearest_slv, lca_<field>, ...)."""
        return self.H.sina_host_result_attr(self.h, q, name.encode()).decode()

    def search_seconds(self):
        return self.H.sina_host_search_seconds(self.h)

    def close(self):
        if self.h:
            self.H.sina_host_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_compare(a_aligned, b_aligned, iupac=0, dist=0, cover=1, filter_lc=False):
    """The host stage's cseq_comparator on two aligned strings: (score, six counters)."""
    H = load_host()
    sc = C.c_float()
    cnt = (C.c_int32 * 6)()
    _chk(H.sina_host_compare(a_aligned.encode(), b_aligned.encode(), iupac, dist, cover, int(filter_lc),
                             C.byref(sc), cnt))
    return np.float32(sc.value), tuple(cnt)


def sidx_load(path, k=10, nofast=False, ids_cap=1 << 26):
    """The reference's .sidx index cache as CSR: (n_sequences, offsets[4^k+1], ids)."""
    H = load_host()
    off = np.zeros((1 << (2 * k)) + 1, np.uint32)
    ids = np.zeros(ids_cap, np.uint32)
    n, nid = C.c_uint32(), C.c_uint64()
    _chk(H.sina_host_sidx_load(path.encode(), k, int(nofast), C.byref(n), off.ctypes.data_as(capi.u32p),
                               ids.ctypes.data_as(capi.u32p), ids_cap, C.byref(nid)))
    return n.value, off, ids[:nid.value].copy()


def sidx_store(path, n_sequences, offsets, ids, k=10, nofast=False):
    """Writes a CSR index as the reference's .sidx file (names ref0, ref1, ...)."""
    H = load_host()
    offsets = np.ascontiguousarray(offsets, np.uint32)
    ids = np.ascontiguousarray(ids, np.uint32)
    _chk(H.sina_host_sidx_store(path.encode(), k, int(nofast), n_sequences, offsets.ctypes.data_as(capi.u32p),
                                ids.ctypes.data_as(capi.u32p), len(ids)))


def _set_options(H, stage, opts):
    for k, v in (opts or {}).items():
        if isinstance(v, bool):
            v = "1" if v else "0"
        _chk(H.sina_host_set_option(stage.encode(), k.encode(), str(v).encode()))


def run_fasta(store, in_path, out_path, famfinder=None, aligner=None, search=None, fasta=None, show_dist=False,
              log_path="", batch=1024, serial=False):
    """`sina -i in_path -o out_path --db <store> [--search] [--show-dist]`: FASTA in, aligned FASTA out, through
    the stage mirror.  Returns dict(read, aligned, written, skipped, avg_sps, avg_cpm, avg_idty)."""
    H = load_host()
    H.sina_host_reset_options()
    _set_options(H, "famfinder", dict({"db": store.key}, **(famfinder or {})))
    _set_options(H, "aligner", dict({"db": store.key}, **(aligner or {})))
    if search is not None:
        _set_options(H, "search", dict({"search-db": store.key}, **search))
    _set_options(H, "fasta", fasta)
    out = (C.c_double * 7)()
    # (serial=True: one batch at a time, stage after stage -- the parity driver; default: the stages as concurrent
    # nodes with an ordered, parallel-composing sink; same output byte for byte)
    run = H.sina_host_run_fasta_serial if serial else H.sina_host_run_fasta
    _chk(run(in_path.encode(), out_path.encode(), log_path.encode(), int(search is not None), int(show_dist), batch, out))
    return dict(read=int(out[0]), aligned=int(out[1]), written=int(out[2]), skipped=int(out[3]), avg_sps=out[4],
                avg_cpm=out[5], avg_idty=out[6])


def fasta_roundtrip(in_path, out_path, fasta=None):
    """FASTA reader -> writer only (no GPU): (sequences read, sequences skipped for bad characters)."""
    H = load_host()
    H.sina_host_reset_options()
    _set_options(H, "fasta", fasta)
    n, sk = C.c_int(), C.c_int()
    _chk(H.sina_host_fasta_roundtrip(in_path.encode(), out_path.encode(), C.byref(n), C.byref(sk)))
    return n.value, sk.value

