/*
 * sina_hip.h -- C ABI of the MI355X (gfx950) implementation of SINA's per-query
 * hot path: k-mer reference search and partial-order ("mesh") DP alignment.
 *
 * The reference (epruesse/SINA) has no FFI; this is the boundary its stage
 * functors would bind.  Each entry point names the reference code it replaces
 * (paths under the SINA source tree).  Conventions:
 *   - plain pointers + sizes, caller-owned host buffers unless stated;
 *   - every function returns 0 on success, non-zero on error, and
 *     sina_hip_last_error() (thread-local) describes the failure -- the C++
 *     stage functors turn that into SINA's exception / tray.log conventions;
 *   - no C++ types, exceptions or torch types cross this line;
 *   - a context may be used from several host threads; calls on one context
 *     serialise on an internal mutex (one HIP stream per context).
 *
 * Sequences use SINA's own packed form (src/aligned_base.h:287-319):
 *   aligned base = uint32: (column & 0xFFFFFF) | iupac_mask << 24
 *   iupac mask   = A 1, G 2, C 4, T/U 8, lower-case bit 16 (src/aligned_base.h:47-52)
 */
#ifndef SINA_HIP_H
#define SINA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SINA_HIP_ABI_VERSION 5  /* 5: sina_hip_stats grew scout_ms, scout_launches; 4: the row-skip counters, sina_hip_debug_dp_info, _rgain */

typedef struct sina_hip_ctx sina_hip_ctx;

/* ------------------------------------------------------------------ lifecycle */

int sina_hip_abi_version(void);
const char *sina_hip_last_error(void);

/* Creates a context on HIP device `device` (own stream). */
int sina_hip_init(int device, sina_hip_ctx **ctx);
/* A second context on the same device that shares `parent`'s reference store, k-mer index and
 * statistics but has its own stream and scratch buffers: calls on different contexts run
 * concurrently on the GPU (the reference gets the same effect from its TBB pipeline running
 * several famfinder/aligner filters at once, src/sina.cpp:430-470).  A fork cannot change the
 * store (upload_refs / build_index / upload_index fail on it); destroy forks before the parent,
 * and do not change the parent's store while forks are in use. */
int sina_hip_fork(sina_hip_ctx *parent, sina_hip_ctx **ctx);
/* Brings the scratch buffers ONE kind of call uses (0: k-mer search, 1: alignment, 2: search-stage comparison) to the
 * largest sizes any context of the store has needed for them so far: device allocations stall every stream of the
 * device, so a host that runs a pipeline makes its worker contexts, and warms them, before the run -- not in it
 * (the reference sizes its per-thread state the same way: one famfinder / aligner copy per worker, made before the
 * flow graph starts, src/sina.cpp:497-519).  No-op where nothing is known yet or the buffers are big enough. */
int sina_hip_prewarm(sina_hip_ctx *ctx, int kind);
void sina_hip_destroy(sina_hip_ctx *ctx);
/* Blocks until all work queued on the context's stream has finished. */
int sina_hip_sync(sina_hip_ctx *ctx);

/* ------------------------------------------------------- reference store + index
 * Replaces: the in-memory sequence cache famfinder/kmer_search read through
 * query_arb::getCseq (src/kmer_search.cpp:417, src/query_arb.cpp:742-755) and
 * kmer_search::impl::build / try_load (src/kmer_search.cpp:245-351).
 */

/* Copies n_refs packed aligned references (concatenated `ab`, offsets `off`
 * [n_refs+1]) of common alignment width `width` into HBM. */
int sina_hip_upload_refs(sina_hip_ctx *ctx, const uint32_t *ab, const uint64_t *off,
                         uint32_t n_refs, uint32_t width);

/* Builds the device k-mer index from the uploaded references, on the GPU:
 * posting list of k-mer v = ascending ids of references containing v at least
 * once (unique_prefix_kmers / unique_kmers + IndexBuilder,
 * src/kmer_search.cpp:152-211,245-276).  Stored as plain
 * CSR (u32 offsets [4^k+1], u32 ids); SINA's list inversion (src/idset.h:367-384)
 * is a storage optimisation with identical scores and is not reproduced. */
int sina_hip_build_index(sina_hip_ctx *ctx, unsigned k, int nofast);

/* Copies the device index back to the host (offsets [4^k+1], ids [n_postings as reported by
 * sina_hip_store_view_get]); used to write the reference's .sidx cache file after a device build
 * (kmer_search::impl::store, src/kmer_search.cpp:279-304). */
int sina_hip_download_index(sina_hip_ctx *ctx, uint32_t *offsets, uint32_t *ids);

/* Alternative to build_index: adopt a host-built CSR index. */
int sina_hip_upload_index(sina_hip_ctx *ctx, unsigned k, int nofast, const uint32_t *offsets,
                          const uint32_t *ids, uint64_t n_postings);

/* Device-pointer view of the index and reference store, for the one-off
 * start-up broadcast over RCCL (torch.distributed) described in INTEGRATION.md:
 * rank 0 builds, every rank allocates with sina_hip_alloc_like, then the
 * buffers are broadcast in place. */
typedef struct sina_hip_store_view {
    void *ref_ab;       uint64_t ref_ab_bytes;    /* u32[total bases]           */
    void *ref_off;      uint64_t ref_off_bytes;   /* u64[n_refs+1]              */
    void *idx_offsets;  uint64_t idx_offsets_bytes; /* u32[4^k+1]               */
    void *idx_ids;      uint64_t idx_ids_bytes;   /* u32[n_postings]            */
    uint32_t n_refs, width, k, nofast;
    uint64_t n_postings, total_bases;
} sina_hip_store_view;
int sina_hip_store_view_get(sina_hip_ctx *ctx, sina_hip_store_view *view);
/* Allocates (uninitialised) device buffers of the sizes in `shape` so that a
 * broadcast can fill them; afterwards the context behaves as if it had built
 * the store itself. Pointer members of `shape` are ignored on input and
 * overwritten with the new device pointers. */
int sina_hip_store_alloc_like(sina_hip_ctx *ctx, sina_hip_store_view *shape);

/* ------------------------------------------------------------- k-mer search
 * Replaces kmer_search::impl::find (src/kmer_search.cpp:366-420) for a batch:
 * for every query, count shared k-mers against every reference
 * (prefix_kmers/all_kmers, duplicates counted, final k-mer dropped:
 * src/kmer.h:188-201) and return the top-`max` by (score desc, id desc)
 * (std::greater<std::pair<int16_t,int>>, src/kmer_search.cpp:412).
 *
 *   qmask/qoff : concatenated iupac masks of the queries' bases, offsets [nq+1]
 *   max        : results wanted per query (clamped to n_refs)
 *   out_ids    : [nq * max] reference ids, out_scores: [nq * max] as float
 *                (result_item.score is float, src/search.h:57-60)
 *   out_n      : [nq] number of valid results per query
 */
int sina_hip_kmer_topk(sina_hip_ctx *ctx, const uint8_t *qmask, const uint64_t *qoff, uint32_t nq,
                       uint32_t max, uint32_t *out_ids, float *out_scores, uint32_t *out_n);

/* Full score vector (score+offset of src/kmer_search.cpp:405-409) of one query
 * against every reference; for tests and the search_filter stage. */
int sina_hip_kmer_scores(sina_hip_ctx *ctx, const uint8_t *qmask, uint32_t qlen, int16_t *scores);

/* ------------------------------------------------------------- search stage
 * Replaces the per-candidate cseq_comparator::operator() calls of search_filter::operator()
 * (src/search_filter.cpp:311-313 and :274-276; traverse + match_counter,
 * src/cseq_comparator.cpp:56-240): for every query (an ALIGNED sequence: packed aligned bases
 * u32 = column | mask << 24, bit 4 of the mask = lower case) and every candidate reference of
 * the store, the six counters the comparator derives its score from.  The float score itself
 * ((float)match / base, Jukes-Cantor) stays with the caller, as in the reference.
 *
 *   q_ab/q_off     : concatenated packed aligned bases of the queries, offsets [nq+1]
 *   cand_ids/off   : concatenated candidate reference ids per query, offsets [nq+1]
 *   iupac_rule     : SINA_CMP_IUPAC_* (base_comp_optimistic / pessimistic / exact, query first)
 *   filter_lowercase : filter_lowercase instead of filter_none
 *   out            : [cand_off[nq]] counters
 * Columns must ascend strictly within a sequence; a side without any unfiltered base gives
 * all-zero counters (the reference dereferences end() there). */
#define SINA_CMP_IUPAC_OPTIMISTIC 0
#define SINA_CMP_IUPAC_PESSIMISTIC 1
#define SINA_CMP_IUPAC_EXACT 2
typedef struct sina_hip_match_counts {
    int32_t only_a_overhang, only_b_overhang, only_a, only_b, match, mismatch;
} sina_hip_match_counts;
int sina_hip_compare(sina_hip_ctx *ctx, const uint32_t *q_ab, const uint64_t *q_off, uint32_t nq,
                     const uint32_t *cand_ids, const uint64_t *cand_off, int iupac_rule, int filter_lowercase,
                     sina_hip_match_counts *out);

/* ------------------------------------------------------------- alignment
 * Replaces, for a batch of queries: mseq::mseq + sort + reduce_edges
 * (src/mseq.cpp:47-118, src/graph.h:451-488) when given family ids, compute()
 * with compute_node_simple<transition_simple|transition_aspace_aware> over
 * scoring_scheme_simple|weighted (src/mesh.h:263-528, src/scoring_schemes.h:
 * 102-241) and the walk of backtrack() (src/mesh.h:535-721).
 */
enum { SINA_OVERHANG_ATTACH = 0, SINA_OVERHANG_REMOVE = 1, SINA_OVERHANG_EDGE = 2 };
enum { SINA_LOWERCASE_NONE = 0, SINA_LOWERCASE_ORIGINAL = 1, SINA_LOWERCASE_UNALIGNED = 2 };
enum { SINA_INSERTION_SHIFT = 0, SINA_INSERTION_FORBID = 1, SINA_INSERTION_REMOVE = 2 };

typedef struct sina_hip_align_params {
    float match_score;      /* --match-score    (2)  src/align.cpp:250-252 */
    float mismatch_score;   /* --mismatch-score (-1) */
    float gap_penalty;      /* --pen-gap        (5)  */
    float gap_ext_penalty;  /* --pen-gapext     (2)  */
    float fs_weight;        /* --fs-weight      (1)  src/align.cpp:247-249 */
    int32_t overhang;       /* SINA_OVERHANG_*  */
    int32_t lowercase;      /* SINA_LOWERCASE_* */
    int32_t insertion;      /* SINA_INSERTION_* */
    const float *weights;   /* posvar weights => scoring_scheme_weighted, or NULL */
    uint32_t n_weights;
    int32_t assemble;       /* 1: also do the cseq container steps of backtrack() on the device where they
                             * are plain (see sina_hip_align_out::assembled); 0 (default): columns only */
} sina_hip_align_params;
void sina_hip_align_params_default(sina_hip_align_params *p);
/* The aligned columns of the context's last sina_hip_align_graphs / sina_hip_align_families call, laid out like
 * that call's out_pos with qoff[0] taken as 0 (query q's entries start at qoff[q] - qoff[0]).  Host memory owned
 * by the context; valid until the next align call on this context (or its destruction). */
const uint32_t *sina_hip_staged_out_pos(sina_hip_ctx *ctx);

/* Longest query (bases) any entry point takes: k-mer search, family alignment, graph alignment.
 * (The k-mer count kernel keeps a query's k-mer cursors in LDS beside its 64 KiB score tile: 9 bytes
 * per base; the DP itself takes any number of 512-column strips.)  The reference aligns any length
 * (src/mesh.h:76,119-121); the stage functors fail a longer query softly, alone. */
#define SINA_HIP_MAX_QUERY_LEN 10240u

/* A batch of family DAGs in CSR form (node id == topological rank == mesh row).
 * All arrays are concatenated over the nq queries of the batch.  Limits (the call fails, nothing is
 * truncated): at most 65535 nodes per DAG, at most 255 predecessors per node, every predecessor id
 * smaller than its node's id, queries of 1..SINA_HIP_MAX_QUERY_LEN bases. */
typedef struct sina_hip_graph_batch {
    uint32_t nq;
    const uint64_t *node_off;  /* [nq+1] into the node arrays                      */
    const uint64_t *edge_off;  /* [nq+1] into pred                                 */
    const uint32_t *node_pos;  /* alignment column of each node                    */
    const uint8_t *node_mask;  /* iupac mask of each node                          */
    const float *node_weight;  /* mseq_node::weight (src/mseq.cpp:113)             */
    const uint32_t *pred_off;  /* per query N+1 entries, relative to edge_off[q];
                                  stored at node_off[q] + q                        */
    const uint32_t *pred;      /* predecessor node ids, ascending per node         */
    const uint32_t *succ_minpos; /* min column over successors, 1000000 if none
                                  (src/mesh.h:480-484); may be NULL unless FORBID  */
    uint32_t width;            /* alignment width (mseq::getWidth)                 */
    /* --fs-no-graph (src/pseq.{h,cpp}, scoring_scheme_profile src/scoring_schemes.h:37-100): the family
     * as a profile -- one node per column, every node's one predecessor the node before it -- whose
     * match term is base_profile::comp(node, query base) instead of (mask & base) ? match : mismatch.
     * node_score16[16 * node + m]: that term for a query base with iupac mask m (1..15; entry 0 unused),
     * self_score16[m]: comp of a base's own profile with itself (the sum_weight term of backtrack(),
     * src/mesh.h:631-638).  Both NULL (the default) for DAG batches; node_mask / node_weight are
     * ignored when they are set. */
    const float *node_score16;
    const float *self_score16;
} sina_hip_graph_batch;

/* Per-query result of DP + backtrack walk, before the cseq container steps
 * (append rule, setWidth, reverse, fix_duplicate_positions) that the host-side
 * aligner applies (src/mesh.h:723-726). */
typedef struct sina_hip_align_out {
    uint32_t end_m, end_s;   /* start cell of the backtrack (src/mesh.h:567-592)   */
    float raw;               /* value at the end cell  (rval, :618)                */
    float sum_weight;        /* (:621,637,683)                                     */
    int32_t aligned_bases;   /* (:622)                                             */
    int32_t cutoff_head, cutoff_tail;
    uint32_t n_out;          /* number of appended bases written to pos[]          */
    int32_t status;          /* 0 ok; <0 device-side failure                       */
    /* With sina_hip_align_params::assemble: 1 if out_pos holds this query's FINISHED alignment -- n_out
     * packed aligned_base words (column | iupac mask << 24, case bit included) in sequence order,
     * after the append rule, setWidth, reverse and the NAST fix-up (src/mesh.h:603-726,
     * src/cseq.cpp:456-594) -- and the fix-up's log facts below; 0 if out_pos holds the appended
     * columns as without the switch (an insertion that does not fit its gap and makes neighbours
     * move, a column beyond the alignment, more than 4096 bases: the host finishes those). */
    uint32_t assembled;
    uint32_t nast_total, nast_longest, nast_last_run; /* "total inserted bases", "longest insertion",
                                                        * "total inserted bases before shifting"     */
} sina_hip_align_out;

/* Aligns nq queries (upper-cased or not is the caller's business: masks are
 * used as given) against their DAGs.
 *   qmask/qoff : concatenated query iupac masks, offsets [nq+1]
 *   out        : [nq]
 *   out_pos    : concatenated like qmask; for query q, out_pos[qoff[q] + i] is
 *                the column handed to the i-th cseq::append() call of
 *                backtrack() (i.e. in reverse query order, tail overhang first) --
 *                or, with p->assemble and out[q].assembled, the finished alignment.
 *                May be NULL: the columns then stay where the device copied them, in the context's
 *                pinned staging buffer -- see sina_hip_staged_out_pos (spares a 6 KB copy per 16S query).
 */
int sina_hip_align_graphs(sina_hip_ctx *ctx, const sina_hip_graph_batch *g, const uint8_t *qmask,
                          const uint64_t *qoff, const sina_hip_align_params *p,
                          sina_hip_align_out *out, uint32_t *out_pos);

/* Same, but the DAGs are built on the GPU from family member ids into the
 * uploaded reference store (order of ids = family order = mseq input order).
 *   fam_ids/fam_off : concatenated reference ids, offsets [nq+1]
 */
int sina_hip_align_families(sina_hip_ctx *ctx, const uint32_t *fam_ids, const uint64_t *fam_off,
                            uint32_t nq, const uint8_t *qmask, const uint64_t *qoff,
                            const sina_hip_align_params *p, sina_hip_align_out *out,
                            uint32_t *out_pos);

/* Test hook: DP planes of ONE query for bit-exact comparison with the oracle.
 * tb_vm/tb_vs: [N*L] value_midx / value_sidx; value: [N*L] float (may be NULL).
 * prune: 0 = every row of every strip is swept (the planes are the reference's cell for cell), 1 = the launch as
 * production makes it, certified row skip included (cells it proved irrelevant hold what it left there). */
int sina_hip_debug_mesh(sina_hip_ctx *ctx, const sina_hip_graph_batch *g, const uint8_t *qmask,
                        uint32_t qlen, const sina_hip_align_params *p, uint32_t *tb_vm,
                        uint32_t *tb_vs, float *value, int prune);

/* Test hooks for the certified row skip: what the DP kernel reported for query q of the context's LAST launch
 * (attempts 0: that launch swept everything), and the first n entries of the per-node bound R(m) the last launch /
 * the last sina_hip_debug_family_graph left on the device (units of 1/64; entry i belongs to node i of the launch's
 * first DAG) with C(m), the number of occupied columns right of the node's.  prune_step = the largest gain of one
 * match step the launch assumed, prune_gmin = the smallest column maximum of query q's DAG, same units. */
typedef struct sina_hip_dp_info {
    uint32_t end_m, end_s;
    float raw;
    int32_t status;
    uint32_t rows_swept, cells_swept, attempts;
    float gain0, ubound;
    uint32_t prune_step, prune_gmin;
    float scout;  /* what the scout pass found for this query (the first attempt's bound U); NaN: the launch had none */
} sina_hip_dp_info;
int sina_hip_debug_dp_info(sina_hip_ctx *ctx, uint32_t q, sina_hip_dp_info *out);
int sina_hip_debug_rgain(sina_hip_ctx *ctx, uint32_t n, uint32_t *out, uint32_t *cols_right /* C(m), may be NULL */);

/* Test hook: the DAG the GPU builds for ONE family (ids into the uploaded store, in family
 * order), in compact CSR form, for comparison with mseq (src/mseq.cpp:47-118).
 * ring_depth is the number of LDS row slots the DP kernel will have; spill_idx reports where each
 * finished DP row is kept: 0xFFFFFFFF nowhere, slot number, or 0x80000000 | spill row. */
int sina_hip_debug_family_graph(sina_hip_ctx *ctx, const uint32_t *fam_ids, uint32_t F, float fs_weight,
                                uint32_t ring_depth, uint32_t *n_nodes, uint32_t *n_edges, uint32_t *pos,
                                uint8_t *mask, float *weight, uint32_t *pred_off, uint32_t *pred,
                                uint32_t *succ_minpos, uint8_t *sink, uint32_t *spill_idx, uint32_t cap_nodes,
                                uint32_t cap_edges);

/* Cumulative statistics of this context and its forks since sina_hip_init (callers take
 * differences); kernel times come from HIP events on the context stream. */
typedef struct sina_hip_stats {
    double dp_ms;          /* mesh DP fill kernel(s)                  */
    double backtrack_ms;   /* backtrack walk kernel                   */
    double graph_ms;       /* device DAG build                        */
    double kmer_count_ms;  /* k-mer count kernel                      */
    double kmer_select_ms; /* top-k select kernel                     */
    uint64_t dp_cells;     /* sum of N*L over the batch               */
    uint64_t postings;     /* postings visited by the count kernel    */
    uint32_t dp_launches, kmer_launches;
    double compare_ms;       /* search-stage comparison kernel            */
    uint64_t compare_bases;  /* candidate bases streamed by it            */
    uint32_t compare_launches;
    uint32_t n_dense_lists;  /* gauge, not cumulative: posting lists the index currently also holds as
                                reference bitmaps (0 until the first search after an index change) */
    double dp_busy_ms;       /* time during which a DP kernel was resident: dp_ms minus the time a launch shared
                                the device with the launch before it (a DP launch starts when its predecessor
                                has dispatched its last workgroup, not when it has ended)                     */
    uint64_t dags_built;     /* family DAGs built on the device ...                                             */
    uint64_t dags_used;      /* ... and queries aligned against them (queries with the same ordered family share one) */
    /* Certified row skip of the DP kernel (the reference fills every cell, src/mesh.h:512-528; here a row of a
     * 512-column strip whose every input provably exceeds what any path ending at the optimum can hold is not
     * swept -- the end cell, its value and the whole trace-back path are certified identical, or the query is
     * swept again): dp_cells above stays the NOMINAL N*L. */
    uint64_t dp_rows;            /* (row, strip) pairs of the launches, nominal                                  */
    uint64_t dp_rows_swept;      /* ... actually swept, second and third attempts included                       */
    uint64_t dp_cells_swept;     /* cells actually computed (= dp_cells where nothing was skipped)               */
    uint64_t dp_queries_pruned;  /* queries that went through the skipping kernel ...                            */
    uint64_t dp_second_attempts; /* ... whose first bound failed its certificate (swept again under the bound the
                                    first attempt found) ...                                                      */
    uint64_t dp_full_sweeps;     /* ... and whose second did too (swept in full)                                 */
    double dp_prune_rho;         /* gauge: the guess (optimum / bound on the whole gain) the next launch starts with */
    uint64_t graph_bytes;        /* device DAG build, algorithmic bytes: the families' packed bases read once, the DAGs
                                    (row records, columns, row-skip bounds, predecessor lists) written once          */
    uint32_t graph_launches, kmer_queries; /* DAG-build launches; queries searched by the k-mer count kernel           */
    double scout_ms;             /* the scout pass (a bound on the optimum per query from a banded sweep, one lane per
                                    query): launch to end, on the context's own stream beside other batches' kernels    */
    uint32_t scout_launches, pad_;
} sina_hip_stats;
int sina_hip_get_stats(sina_hip_ctx *ctx, sina_hip_stats *s);

#ifdef __cplusplus
}
#endif
#endif /* SINA_HIP_H */
