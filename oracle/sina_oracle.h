/*
 * sina_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded, literal CPU restatement of the SINA per-query hot
 * path (k-mer reference search -> family selection -> family DAG -> mesh DP ->
 * backtrack -> NAST fix-up).  Every function cites the reference file:line it
 * follows (paths relative to /root/reference).
 *
 * This library is the *checker*.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (sina_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - k-mer generator, posting lists: pinned against the reference's own KAT
 *     tables (src/unit_tests/kmer_test.cpp:51-163) and against the real
 *     reference headers compiled in oracle/_ref (kmer.h, idset.h,
 *     aligned_base.cpp).
 *   - cseq container ops (append/setWidth/reverse/complement/getAligned):
 *     pinned against src/unit_tests/cseq_test.cpp KATs.
 *   - family DAG, mesh DP, backtrack, NAST fix-up: the reference has no unit
 *     test or fixture for these (Makefile.am:265-274) and mesh.h/mseq.cpp/
 *     cseq.cpp cannot be compiled here (they include Boost headers that are
 *     absent from the image).  The scoring arithmetic and the DAG container
 *     are cross-checked against the real scoring_schemes.h / graph.h /
 *     aligned_base.cpp in oracle/_ref; the recurrence, backtrack and NAST are
 *     restated from the source.  PARITY UNPINNED for those three rows beyond
 *     the probe vector recorded in SURVEY.md section 8c.
 */
#ifndef SINA_ORACLE_H
#define SINA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- bases: src/aligned_base.h:38-52,287-319; src/aligned_base.cpp:70-114 */
/* aligned base packed as in aligned_compact: (pos & 0xFFFFFF) | mask << 24   */
#define SO_AB(pos, mask) ((uint32_t)(((uint32_t)(pos) & 0xFFFFFFu) | ((uint32_t)(mask) << 24)))
#define SO_POS(ab) ((ab) & 0xFFFFFFu)
#define SO_MASK(ab) ((uint8_t)((ab) >> 24))

int so_char_to_mask(int c);  /* -1: bad character (would throw); 0 for '-' '.' */
int so_mask_to_rna(int mask);
int so_mask_to_dna(int mask);

/* ---- growable log text (stands in for std::ostream& log / tray.log) */
typedef struct so_log {
    char *s;
    size_t n, cap;
} so_log;
void so_log_init(so_log *l);
void so_log_free(so_log *l);
const char *so_log_str(so_log *l);

/* ---- cseq_base: src/cseq.h:47-165, src/cseq.cpp */
typedef struct so_cseq {
    uint32_t *ab;
    uint32_t n, cap;
    uint32_t width;
    char name[64];
} so_cseq;

so_cseq *so_cseq_new(const char *name);
so_cseq *so_cseq_clone(const so_cseq *c);
void so_cseq_free(so_cseq *c);
void so_cseq_clear(so_cseq *c);
int so_cseq_append_str(so_cseq *c, const char *str);        /* -1: bad character */
void so_cseq_append_base(so_cseq *c, uint32_t ab, so_log *errlog);
int so_cseq_set_width(so_cseq *c, uint32_t w);              /* -1: would throw   */
void so_cseq_reverse(so_cseq *c);
void so_cseq_complement(so_cseq *c);
void so_cseq_upper(so_cseq *c);
/* out must hold width+1 bytes */
void so_cseq_get_aligned(const so_cseq *c, int nodots, int dna, char *out);
/* out must hold n+1 bytes */
void so_cseq_get_bases(const so_cseq *c, char *out);
int so_cseq_fix_duplicate_positions(so_cseq *c, so_log *log, int lowercase, int remove); /* -1: throw */

/* raw-array helpers for ctypes */
uint32_t so_cseq_size(const so_cseq *c);
uint32_t so_cseq_width(const so_cseq *c);
const uint32_t *so_cseq_data(const so_cseq *c);
void so_cseq_set_data(so_cseq *c, const uint32_t *ab, uint32_t n, uint32_t width);

/* ---- k-mers: src/kmer.h */
/* Per-push trace of the generator stack (for the kmer_test.cpp tables).
 * p_len == 0 disables the prefix filter; unique != 0 enables unique_filter. */
void so_kmer_trace(const char *seq, unsigned k, unsigned p_len, unsigned p_val, int unique,
                   uint8_t *good_out, uint32_t *val_out);
/* The iterable (range-for) semantics incl. the dropped final k-mer
 * (src/kmer.h:188-201).  Returns the number of k-mers written. */
uint32_t so_kmers(const uint32_t *ab, uint32_t n, unsigned k, unsigned p_len, unsigned p_val,
                  int unique, uint32_t *out);

/* ---- posting lists: src/idset.h:279-410 (vlimap) */
typedef struct so_vlimap {
    uint8_t *data;
    size_t nbytes, cap;
    size_t size;
    int16_t inc;
    uint32_t last, maxsize;
} so_vlimap;
so_vlimap *so_vlimap_new(uint32_t maxsize);
void so_vlimap_free(so_vlimap *v);
void so_vlimap_push_back(so_vlimap *v, uint32_t n);
int so_vlimap_increment(const so_vlimap *v, int16_t *t);
void so_vlimap_append(so_vlimap *v, const so_vlimap *o);
void so_vlimap_invert(so_vlimap *v);
size_t so_vlimap_bytes(const so_vlimap *v, const uint8_t **data);

/* ---- k-mer search engine: src/kmer_search.cpp:152-276,366-420 */
typedef struct so_index so_index;
so_index *so_index_build(const so_cseq *const *refs, uint32_t n_refs, unsigned k, int nofast);
void so_index_free(so_index *idx);
uint32_t so_index_size(const so_index *idx);
/* scores[i] = score + offset for every reference (the `ranks` vector before sorting) */
void so_index_scores(const so_index *idx, const so_cseq *query, int16_t *scores);
/* find(): top-`max` by (score desc, id desc). out arrays must hold min(max,n) entries */
uint32_t so_index_find(const so_index *idx, const so_cseq *query, uint32_t max,
                       uint32_t *out_ids, float *out_scores);
/* the reference's on-disk index cache (.sidx): kmer_search::impl::store / try_load,
 * src/kmer_search.cpp:66-88,279-351 + vlimap::write/read, src/idset.h:386-410.
 * write: 0 ok.  read: NULL on wrong magic / version / k / nofast. */
int so_index_write(const so_index *idx, const char *const *names, const char *path);
so_index *so_index_read(const char *path, unsigned k, int nofast);
/* CSR export of the (un-inverted) index for loading the device index in tests */
uint64_t so_index_csr(const so_index *idx, uint32_t *offsets /* 4^k+1 or NULL */,
                      uint32_t *ids /* or NULL */);

/* ---- famfinder: src/famfinder.cpp:439-494,497-612 */
typedef struct so_ff_opts {
    uint32_t fs_min, fs_max;
    float fs_msc, fs_msc_max;
    int fs_leave_query_out;
    uint32_t fs_req, fs_req_full, fs_full_len, fs_req_gaps, fs_min_len, fs_cover_gene;
} so_ff_opts;
void so_ff_opts_default(so_ff_opts *o);
/* famfinder::impl::turn_check, src/famfinder.cpp:344-378: 0 none, 1 reversed, 2 complemented,
 * 3 reversed and complemented; scores4 (may be NULL) receives the four top-1 scores */
int so_turn_check(const so_index *idx, const so_cseq *query, int all, float *scores4);
/* returns number of family members (0 + log text "unable to align" if < fs_req) */
uint32_t so_famfinder(const so_index *idx, const so_cseq *const *refs, const so_cseq *query,
                      const so_ff_opts *o, uint32_t *out_ids, float *out_scores, uint32_t cap,
                      so_log *log);

/* ---- family DAG: src/mseq.cpp:47-118, src/graph.h:332-357,451-488 */
typedef struct so_graph {
    uint32_t n, width;
    uint32_t *pos;
    uint8_t *mask;
    float *weight;
    uint32_t *pred_off, *pred;
    uint32_t *succ_off, *succ;
    uint32_t n_src, *src;
    uint32_t n_snk, *snk;
    /* --fs-no-graph (pseq, src/pseq.{h,cpp}): the family as a PROFILE -- one node per column, a linear
     * chain -- instead of a DAG.  prof[6 * m] = {A, G, C, T/U share, gap-open share, gap-extend share}
     * of node m (base_profile, pseq.h:54-63); NULL for an mseq graph (mask / weight are unused then). */
    float *prof;
} so_graph;
so_graph *so_mseq_build(const so_cseq *const *fam, uint32_t F, float weight); /* NULL: throw */
/* pseq::pseq, src/pseq.cpp:41-112.  PARITY UNPINNED for --fs-no-graph: the reference holds no test or vector
 * for it and pseq.h does not compile here (Boost); checked against hand-worked values only */
so_graph *so_pseq_build(const so_cseq *const *fam, uint32_t F);
/* base_profile::comp (pseq.h:100-117) of node profile `prof` (6 floats) against the base with iupac mask
 * `smask`, or -- prof NULL -- of that base's own profile against itself (the "had there been a match"
 * term of backtrack(), mesh.h:631-638) */
float so_profile_comp(const float *prof, int smask, float match, float mismatch, float gap, float gap_ext);
void so_graph_free(so_graph *g);

/* ---- mesh DP: src/mesh.h:263-528, src/scoring_schemes.h:102-241 */
typedef struct so_cell {
    uint32_t value_midx, value_sidx, gapm_idx, gaps_idx;
    float value, gapm_val, gaps_val;
    uint32_t gaps_max; /* transition_aspace_aware only (src/mesh.h:390-401) */
} so_cell;

enum { SO_OVERHANG_ATTACH = 0, SO_OVERHANG_REMOVE = 1, SO_OVERHANG_EDGE = 2 };
enum { SO_LOWERCASE_NONE = 0, SO_LOWERCASE_ORIGINAL = 1, SO_LOWERCASE_UNALIGNED = 2 };
enum { SO_INSERTION_SHIFT = 0, SO_INSERTION_FORBID = 1, SO_INSERTION_REMOVE = 2 };

typedef struct so_align_opts {
    float match_score, mismatch_score, gap_penalty, gap_ext_penalty; /* 2,-1,5,2 */
    float fs_weight;                                                  /* 1 */
    int overhang, lowercase, insertion, realign;
    const float *weights; /* posvar weights (scoring_scheme_weighted) or NULL */
    uint32_t n_weights;
    int fs_no_graph;      /* --fs-no-graph: pseq + scoring_scheme_profile (align.cpp:428-433) */
} so_align_opts;
void so_align_opts_default(so_align_opts *o);

float so_score_op(int op, float prev, uint32_t mpos, int mmask, float mweight, int smask, int offset,
                  float ms, float mms, float gp, float gpe, const float *weights, uint32_t nw);

/* cells: caller-allocated g->n * L array */
void so_mesh_compute(const so_graph *g, const uint32_t *query_ab, uint32_t L,
                     const so_align_opts *o, so_cell *cells);
/* returns score; out receives the aligned result. -1e30f if fix-up would throw */
float so_backtrack(const so_graph *g, const uint32_t *query_ab, uint32_t L, const so_cell *cells,
                   const so_align_opts *o, so_cseq *out, int *cutoff_head, int *cutoff_tail,
                   so_log *log);

/* ---- aligner glue: src/align.cpp:307-460,462-521 */
typedef struct so_align_result {
    int status;   /* 0 aligned by DP, 1 copied alignment, 2 skipped (all removed), -1 error */
    int head, tail, qual;
    float score;
    uint64_t cells; /* N*L of the mesh that was filled (0 if none) */
    float idty;     /* --calc-idty (align.cpp:380-382,443-453): 100 * best overlap identity with a family
                       member; always computed here (40 merge walks, nothing beside the mesh) */
} so_align_result;
void so_align(const so_cseq *const *family, uint32_t F, const so_cseq *query,
              const so_align_opts *o, so_cseq *out, so_align_result *res, so_log *log);

/* ---- cseq_comparator: src/cseq_comparator.cpp:56-296 */
enum { SO_CMP_IUPAC_OPTIMISTIC = 0, SO_CMP_IUPAC_PESSIMISTIC = 1, SO_CMP_IUPAC_EXACT = 2 };
enum { SO_CMP_DIST_NONE = 0, SO_CMP_DIST_JC = 1 };
enum {
    SO_CMP_COVER_ABS = 0, SO_CMP_COVER_QUERY, SO_CMP_COVER_TARGET, SO_CMP_COVER_OVERLAP, SO_CMP_COVER_ALL,
    SO_CMP_COVER_AVERAGE, SO_CMP_COVER_MIN, SO_CMP_COVER_MAX, SO_CMP_COVER_NOGAP
};
typedef struct so_match_counts {  /* match_counter, cseq_comparator.cpp:146-163 */
    int only_a_overhang, only_b_overhang, only_a, only_b, match, mismatch;
} so_match_counts;
/* traverse() with the counter functor; a = query, b = target.  Both must keep at least one
 * unfiltered base (the reference dereferences end() otherwise). */
void so_compare_counts(const so_cseq *a, const so_cseq *b, int iupac, int filter_lc, so_match_counts *m);
float so_compare_score(const so_match_counts *m, int cover, int dist);
float so_compare(const so_cseq *a, const so_cseq *b, int iupac, int dist, int cover, int filter_lc);

/* ---- search_filter::operator(): src/search_filter.cpp:244-412 */
typedef struct so_search_opts {
    uint32_t kmer_candidates, max_result;
    float min_sim, lca_quorum;
    int ignore_super, search_all;
    int iupac, dist, cover, filter_lc;
} so_search_opts;
void so_search_opts_default(so_search_opts *o);
/* the result_vector after scoring, sorting and the min_sim cut (ids into refs, best first).
 * returns -1 (and the reference's log text) when the query is shorter than 20 bases. */
int so_search(const so_index *idx, const so_cseq *const *refs, uint32_t n_refs, const so_cseq *query_aligned,
              const so_search_opts *o, uint32_t *out_ids, float *out_scores, uint32_t cap, so_log *log);
/* "acc.version.start.stop~score " per result (search_filter.cpp:357-363) */
void so_search_nearest(const char *const *acc, const char *const *version, const char *const *start,
                       const char *const *stop, const uint32_t *ids, const float *scores, uint32_t n, so_log *out);
/* the lowest-common-ancestor vote over one taxonomy field of the results (:374-409);
 * tax[i] = that field of result i */
void so_search_lca(const char *const *tax, uint32_t n_results, float quorum, so_log *out);

#ifdef __cplusplus
}
#endif
#endif
