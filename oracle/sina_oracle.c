/*
 * sina_oracle.c -- TEST INFRASTRUCTURE ONLY (see sina_oracle.h).
 *
 * Literal, single-threaded C restatement of the SINA hot path.  Written for
 * clarity and for operation-order fidelity, not for speed: float expressions
 * keep the reference's association and are compiled with -ffp-contract=off.
 * All citations are file:line under /root/reference.
 */
#include "sina_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ log */

void so_log_init(so_log *l) {
    l->cap = 256;
    l->s = (char *)malloc(l->cap);
    l->s[0] = 0;
    l->n = 0;
}
void so_log_free(so_log *l) {
    free(l->s);
    l->s = NULL;
    l->n = l->cap = 0;
}
const char *so_log_str(so_log *l) { return l->s ? l->s : ""; }

static void so_logf(so_log *l, const char *fmt, ...) {
    if (!l || !l->s) return;
    va_list ap;
    for (;;) {
        va_start(ap, fmt);
        int w = vsnprintf(l->s + l->n, l->cap - l->n, fmt, ap);
        va_end(ap);
        if (w < 0) return;
        if ((size_t)w < l->cap - l->n) {
            l->n += (size_t)w;
            return;
        }
        l->cap = (l->cap + (size_t)w) * 2;
        l->s = (char *)realloc(l->s, l->cap);
    }
}

/* ---------------------------------------------------------------- bases */

/* src/aligned_base.cpp:70-114 (A=1 G=2 C=4 T/U=8, lower case bit 16) */
int so_char_to_mask(int c) {
    int lc = 0, m;
    if (c == '-' || c == '.') return 0;
    if (c >= 'a' && c <= 'z') {
        lc = 16;
        c -= 'a' - 'A';
    }
    switch (c) {
    case 'A': m = 1; break;
    case 'G': m = 2; break;
    case 'C': m = 4; break;
    case 'T': case 'U': m = 8; break;
    case 'R': m = 2 | 1; break;
    case 'Y': m = 8 | 4; break;
    case 'K': m = 2 | 8; break;
    case 'M': m = 1 | 4; break;
    case 'S': m = 2 | 4; break;
    case 'W': m = 1 | 8; break;
    case 'B': m = 2 | 8 | 4; break;
    case 'D': m = 2 | 1 | 8; break;
    case 'H': m = 1 | 4 | 8; break;
    case 'V': m = 2 | 4 | 1; break;
    case 'N': m = 15; break;
    default: return -1; /* base_iupac::bad_character_exception, aligned_base.h:77-83 */
    }
    return m | lc;
}

static const char rna_chars[33] = ".AGRCMSVUWKDYHBN.agrcmsvuwkdyhbn";
static const char dna_chars[33] = ".AGRCMSVTWKDYHBN.agrcmsvtwkdyhbn";
int so_mask_to_rna(int mask) { return rna_chars[mask & 31]; }
int so_mask_to_dna(int mask) { return dna_chars[mask & 31]; }

static int is_ambig(uint8_t mask) { return __builtin_popcount(mask & 0xf) > 1; } /* aligned_base.h:140-146 */
static unsigned base_type(uint8_t mask) { return (unsigned)__builtin_ctz(mask & 0xf); } /* :113-115 */
static uint8_t complement_mask(uint8_t d) { /* aligned_base.h:117-124 */
    return (uint8_t)(((d & 2) << 1) | ((d & 4) >> 1) | ((d & 1) << 3) | ((d & 8) >> 3) | (d & 16));
}

/* ----------------------------------------------------------------- cseq */

so_cseq *so_cseq_new(const char *name) {
    so_cseq *c = (so_cseq *)calloc(1, sizeof(so_cseq));
    if (name) {
        strncpy(c->name, name, sizeof(c->name) - 1);
    }
    return c;
}
so_cseq *so_cseq_clone(const so_cseq *o) {
    so_cseq *c = so_cseq_new(o->name);
    so_cseq_set_data(c, o->ab, o->n, o->width);
    return c;
}
void so_cseq_free(so_cseq *c) {
    if (!c) return;
    free(c->ab);
    free(c);
}
void so_cseq_clear(so_cseq *c) { /* cseq.cpp:56-60 */
    c->n = 0;
    c->width = 0;
}
static void cseq_push(so_cseq *c, uint32_t ab) {
    if (c->n == c->cap) {
        c->cap = c->cap ? c->cap * 2 : 64;
        c->ab = (uint32_t *)realloc(c->ab, c->cap * sizeof(uint32_t));
    }
    c->ab[c->n++] = ab;
}
uint32_t so_cseq_size(const so_cseq *c) { return c->n; }
uint32_t so_cseq_width(const so_cseq *c) { return c->width; }
const uint32_t *so_cseq_data(const so_cseq *c) { return c->ab; }
void so_cseq_set_data(so_cseq *c, const uint32_t *ab, uint32_t n, uint32_t width) {
    c->n = 0;
    for (uint32_t i = 0; i < n; i++) cseq_push(c, ab[i]);
    c->width = width;
}

/* cseq.cpp:62-77 */
int so_cseq_append_str(so_cseq *c, const char *str) {
    while (*str != 0) {
        if (*str != ' ' && *str != '\t' && *str != '\n' && *str != '\r') {
            if (*str != '-' && *str != '.') {
                int m = so_char_to_mask((unsigned char)*str);
                if (m <= 0) return -1;
                cseq_push(c, SO_AB(c->width, m));
            }
            c->width++;
        }
        str++;
    }
    return 0;
}

/* cseq.cpp:79-95 */
void so_cseq_append_base(so_cseq *c, uint32_t ab, so_log *errlog) {
    if (SO_POS(ab) >= c->width) {
        cseq_push(c, ab);
        c->width = SO_POS(ab);
    } else {
        so_logf(errlog, "$ cseq::append(): wrong order! %c(%u<%u)", so_mask_to_rna(SO_MASK(ab)),
                SO_POS(ab), c->width);
        cseq_push(c, SO_AB(c->width, SO_MASK(ab)));
    }
}

/* cseq.cpp:98-132 */
int so_cseq_set_width(so_cseq *c, uint32_t w) {
    if (c->n == 0 || w >= SO_POS(c->ab[c->n - 1]) + 1) {
        c->width = w;
        return 0;
    }
    if (w < c->n) return -1;
    uint32_t skip;
    for (skip = 0; skip < c->n; skip++) {
        if (SO_POS(c->ab[c->n - skip - 1]) + skip < w) break;
    }
    for (uint32_t i = skip; i > 0; --i) {
        uint32_t *b = &c->ab[c->n - i];
        *b = SO_AB(w - i, SO_MASK(*b));
    }
    c->width = w;
    return 0;
}

/* cseq.cpp:283-289 */
void so_cseq_reverse(so_cseq *c) {
    for (uint32_t i = 0, j = c->n; i + 1 < j; i++) {
        --j;
        uint32_t t = c->ab[i];
        c->ab[i] = c->ab[j];
        c->ab[j] = t;
    }
    for (uint32_t i = 0; i < c->n; i++) {
        c->ab[i] = SO_AB(c->width - 1 - SO_POS(c->ab[i]), SO_MASK(c->ab[i]));
    }
}
/* cseq.cpp:291-296 */
void so_cseq_complement(so_cseq *c) {
    for (uint32_t i = 0; i < c->n; i++)
        c->ab[i] = SO_AB(SO_POS(c->ab[i]), complement_mask(SO_MASK(c->ab[i])));
}
/* cseq.cpp:298-303 */
void so_cseq_upper(so_cseq *c) {
    for (uint32_t i = 0; i < c->n; i++) c->ab[i] &= ~((uint32_t)16 << 24);
}

/* cseq.cpp:135-174 */
void so_cseq_get_aligned(const so_cseq *c, int nodots, int dna, char *out) {
    char dot = nodots ? '-' : '.';
    uint32_t cursor = 0, o = 0;
    for (uint32_t i = 0; i < c->n; i++) {
        uint32_t pos = SO_POS(c->ab[i]);
        for (; cursor < pos; cursor++) out[o++] = dot;
        dot = '-';
        cursor = pos;
        out[o++] = (char)(dna ? so_mask_to_dna(SO_MASK(c->ab[i])) : so_mask_to_rna(SO_MASK(c->ab[i])));
        cursor++;
    }
    if (cursor < c->width) {
        if (!nodots) dot = '.';
        for (; cursor < c->width; cursor++) out[o++] = dot;
    }
    out[o] = 0;
}
/* cseq.cpp:176-186 */
void so_cseq_get_bases(const so_cseq *c, char *out) {
    for (uint32_t i = 0; i < c->n; i++) out[i] = (char)so_mask_to_rna(SO_MASK(c->ab[i]));
    out[c->n] = 0;
}

/* cseq.cpp:456-594 -- NAST insertion fix-up, restated with indices for iterators */
int so_cseq_fix_duplicate_positions(so_cseq *c, so_log *log, int lowercase, int remove) {
    uint32_t total_inserts = 0, longest_insert = 0, orig_inserts = 0;
    uint32_t *b = c->ab;
#define POS(i) SO_POS(b[(i)])
    if (remove) so_logf(log, "insertion=remove not implemented, using shift; ");

    long last_it = 0;
    const long bases_end = (long)c->n;
    for (long curr_it = 0; curr_it < bases_end; ++curr_it) {
        if (POS(last_it) == POS(curr_it)) {
            if (curr_it + 1 != bases_end) continue;
            ++curr_it;
        }
        uint32_t num_inserts = (uint32_t)(curr_it - last_it - 1);
        if (num_inserts == 0) {
            last_it = curr_it;
            continue;
        }
        uint32_t range_begin = POS(last_it) + 1;
        uint32_t range_end = (curr_it == bases_end) ? c->width : POS(curr_it);
        ++last_it;
        --curr_it;

        orig_inserts = num_inserts;
        if (range_end - range_begin < num_inserts) {
            so_logf(log, "shifting bases to fit in %u bases at pos %u to %u;", num_inserts, range_begin,
                    range_end);
            while (range_end - range_begin < num_inserts) {
                int next_left_gap, next_right_gap;
                long left = last_it, right = curr_it;
                if (left == 0) {
                    next_left_gap = (range_begin > 0) ? (int)(range_begin - 1) : -1;
                } else {
                    if (POS(left - 1) + 1 < range_begin) {
                        next_left_gap = (int)(range_begin - 1);
                    } else {
                        --left;
                        while (left != 0 && POS(left - 1) + 1 >= POS(left)) --left;
                        next_left_gap = (int)(POS(left) - 1u);
                    }
                }
                if (right + 1 == bases_end) {
                    next_right_gap = (range_end < c->width) ? (int)range_end : -1;
                } else {
                    if (POS(right + 1) > range_end) {
                        next_right_gap = (int)range_end;
                    } else {
                        ++right;
                        while (right + 1 != bases_end && POS(right) + 1 >= POS(right + 1)) ++right;
                        next_right_gap = (int)(POS(right) + 1);
                    }
                }
                if (next_right_gap == -1 ||
                    (next_left_gap != -1 && (uint32_t)(range_begin - (uint32_t)next_left_gap) <=
                                                (uint32_t)((uint32_t)next_right_gap - (range_end - 1)))) {
                    if (next_left_gap == -1) return -1; /* runtime_error "no space to left and right" */
                    num_inserts += (uint32_t)(last_it - left);
                    range_begin = (uint32_t)next_left_gap;
                    last_it = left;
                } else {
                    num_inserts += (uint32_t)(right - curr_it);
                    range_end = (uint32_t)next_right_gap + 1;
                    curr_it = right;
                }
            }
        } else {
            range_begin = range_end - num_inserts;
        }
        ++curr_it;
        for (; last_it != curr_it; ++last_it) {
            uint8_t m = SO_MASK(b[last_it]);
            if (lowercase) m |= 16;
            b[last_it] = SO_AB(range_begin++, m);
        }
        total_inserts += num_inserts;
        if (num_inserts > longest_insert) longest_insert = num_inserts;
        last_it = curr_it;
    }
#undef POS
    if (total_inserts > 0) {
        so_logf(log, "total inserted bases=%u;longest insertion=%u;total inserted bases before shifting=%u;",
                total_inserts, longest_insert, orig_inserts);
    }
    return 0;
}

/* ---------------------------------------------------------------- k-mers */

/* kmer.h:44-106 generator + :109-125 prefix_filter + :128-151 unique_filter.
 * `seen` is a direct-address byte table (k <= 12) standing in for the
 * unordered_set; semantics (first occurrence wins) are identical. */
typedef struct {
    unsigned k, mask, val, good_count;
    unsigned p_len, p_mask, p_val;
    int unique, is_good;
    uint8_t *seen;
} kgen;

static void kgen_init(kgen *g, unsigned k, unsigned p_len, unsigned p_val, int unique) {
    memset(g, 0, sizeof(*g));
    g->k = k;
    g->mask = (unsigned)((1UL << (2 * k)) - 1);
    g->p_len = p_len;
    if (p_len) {
        g->p_mask = ((1u << (p_len * 2)) - 1) << ((k - p_len) * 2);
        g->p_val = p_val << ((k - p_len) * 2);
    }
    g->unique = unique;
    if (unique) g->seen = (uint8_t *)calloc((size_t)1 << (2 * k), 1);
}
static void kgen_free(kgen *g) { free(g->seen); }
static int kgen_base_good(const kgen *g) { /* generator::good && prefix_filter::good */
    if (g->good_count < g->k) return 0;
    if (g->p_len && (g->val & g->p_mask) != g->p_val) return 0;
    return 1;
}
static void kgen_push(kgen *g, uint8_t mask) {
    if (is_ambig(mask)) {
        g->good_count = 0;
    } else {
        g->good_count++;
        g->val <<= 2;
        g->val &= g->mask;
        g->val += base_type(mask);
    }
    if (g->unique) { /* kmer.h:143-146 */
        g->is_good = 0;
        if (kgen_base_good(g)) {
            if (!g->seen[g->val]) {
                g->seen[g->val] = 1;
                g->is_good = 1;
            }
        }
    }
}
static int kgen_good(const kgen *g) { return g->unique ? g->is_good : kgen_base_good(g); }

void so_kmer_trace(const char *seq, unsigned k, unsigned p_len, unsigned p_val, int unique,
                   uint8_t *good_out, uint32_t *val_out) {
    kgen g;
    kgen_init(&g, k, p_len, p_val, unique);
    for (size_t i = 0; seq[i]; i++) {
        int m = so_char_to_mask((unsigned char)seq[i]);
        kgen_push(&g, (uint8_t)(m < 0 ? 0 : m));
        good_out[i] = (uint8_t)kgen_good(&g);
        val_out[i] = g.val;
    }
    kgen_free(&g);
}

/* kmer.h:155-203 iterable::iterator: the ctor and operator++ push bases until
 * good() or the input is exhausted; operator!= compares the input cursor with
 * end, so a k-mer completed by the LAST base is never yielded. */
uint32_t so_kmers(const uint32_t *ab, uint32_t n, unsigned k, unsigned p_len, unsigned p_val,
                  int unique, uint32_t *out) {
    kgen g;
    uint32_t cnt = 0, begin = 0;
    kgen_init(&g, k, p_len, p_val, unique);
    if (begin != n) {
        do {
            kgen_push(&g, SO_MASK(ab[begin++]));
        } while (!kgen_good(&g) && begin != n);
    }
    while (begin != n) { /* it != end() */
        out[cnt++] = g.val;
        do {
            kgen_push(&g, SO_MASK(ab[begin++]));
        } while (!kgen_good(&g) && begin != n);
    }
    kgen_free(&g);
    return cnt;
}

/* ----------------------------------------------------------- vlimap */

so_vlimap *so_vlimap_new(uint32_t maxsize) {
    so_vlimap *v = (so_vlimap *)calloc(1, sizeof(so_vlimap));
    v->inc = 1;
    v->maxsize = maxsize;
    return v;
}
void so_vlimap_free(so_vlimap *v) {
    if (!v) return;
    free(v->data);
    free(v);
}
static void vl_byte(so_vlimap *v, uint8_t b) {
    if (v->nbytes == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 16;
        v->data = (uint8_t *)realloc(v->data, v->cap);
    }
    v->data[v->nbytes++] = b;
}
/* idset.h:279-286 */
static void vl_abs_push(so_vlimap *v, uint32_t n) {
    while (n > 127) {
        vl_byte(v, (uint8_t)(n | 0x80));
        n >>= 7;
    }
    vl_byte(v, (uint8_t)n);
    ++v->size;
}
/* idset.h:310-313 */
void so_vlimap_push_back(so_vlimap *v, uint32_t n) {
    vl_abs_push(v, n - v->last);
    v->last = n;
}
/* idset.h:315-337 */
int so_vlimap_increment(const so_vlimap *v, int16_t *t) {
    size_t it = 0, end = v->nbytes;
    uint32_t last = 0;
    while (it != end) {
        uint8_t byte = v->data[it];
        if (byte < 128) {
            last += byte;
        } else {
            uint32_t val = byte & 0x7f;
            unsigned shift = 7;
            do {
                byte = v->data[++it];
                val |= (uint32_t)(byte & 0x7f) << shift;
                shift += 7;
            } while (byte >= 128);
            last += val;
        }
        t[last] = (int16_t)(t[last] + v->inc);
        ++it;
    }
    return 1 - (v->inc + 1) / 2;
}
/* decode one varint at *it (vlimap_abs::const_iterator::operator*, idset.h:206-259) */
static uint32_t vl_decode(const so_vlimap *v, size_t *it) {
    uint8_t byte = v->data[*it];
    uint32_t val;
    if (byte < 128) {
        val = byte;
    } else {
        val = byte & 0x7f;
        unsigned shift = 7;
        do {
            byte = v->data[++*it];
            val |= (uint32_t)(byte & 0x7f) << shift;
            shift += 7;
        } while (byte >= 128);
    }
    ++*it;
    return val;
}
/* idset.h:343-361 */
void so_vlimap_append(so_vlimap *v, const so_vlimap *o) {
    if (o->nbytes == 0) return;
    if (v->nbytes == 0) {
        for (size_t i = 0; i < o->nbytes; i++) vl_byte(v, o->data[i]);
        v->last = o->last;
        return; /* NB: the reference does not copy _size here */
    }
    size_t it = 0;
    uint32_t val = vl_decode(o, &it);
    so_vlimap_push_back(v, val);
    for (; it < o->nbytes; it++) vl_byte(v, o->data[it]);
    v->last = o->last;
}
/* idset.h:367-384 */
void so_vlimap_invert(so_vlimap *v) {
    so_vlimap *res = so_vlimap_new(0);
    res->inc = -1;
    uint32_t next = 0, last = 0;
    size_t it = 0;
    while (it < v->nbytes) {
        uint32_t inc = vl_decode(v, &it);
        next += inc;
        while (last < next) so_vlimap_push_back(res, last++);
        ++last;
    }
    while (last < v->maxsize) so_vlimap_push_back(res, last++);
    /* std::swap(data,_inc,_last,_maxsize); _size is NOT swapped */
    free(v->data);
    v->data = res->data;
    v->nbytes = res->nbytes;
    v->cap = res->cap;
    res->data = NULL;
    {
        int16_t ti = v->inc;
        v->inc = res->inc;
        res->inc = ti;
        uint32_t tl = v->last;
        v->last = res->last;
        res->last = tl;
        uint32_t tm = v->maxsize;
        v->maxsize = res->maxsize;
        res->maxsize = tm;
    }
    so_vlimap_free(res);
}
size_t so_vlimap_bytes(const so_vlimap *v, const uint8_t **data) {
    if (data) *data = v->data;
    return v->nbytes;
}

/* ------------------------------------------------------------- index */

struct so_index {
    unsigned k, n_kmers;
    uint32_t n_sequences;
    int nofast;
    so_vlimap **kmer_idx;
};

/* kmer_search.cpp:152-181 (IndexBuilder, single range) + :245-276 (build) */
so_index *so_index_build(const so_cseq *const *refs, uint32_t n_refs, unsigned k, int nofast) {
    so_index *idx = (so_index *)calloc(1, sizeof(so_index));
    idx->k = k;
    idx->n_kmers = 1u << (k * 2);
    idx->n_sequences = n_refs;
    idx->nofast = nofast;
    idx->kmer_idx = (so_vlimap **)calloc(idx->n_kmers, sizeof(so_vlimap *));
    uint32_t maxlen = 0;
    for (uint32_t i = 0; i < n_refs; i++)
        if (refs[i]->n > maxlen) maxlen = refs[i]->n;
    uint32_t *buf = (uint32_t *)malloc(sizeof(uint32_t) * (maxlen + 1));
    for (uint32_t i = 0; i < n_refs; i++) {
        uint32_t nk = nofast ? so_kmers(refs[i]->ab, refs[i]->n, k, 0, 0, 1, buf)
                             : so_kmers(refs[i]->ab, refs[i]->n, k, 1, 0 /*BASE_A*/, 1, buf);
        for (uint32_t j = 0; j < nk; j++) {
            uint32_t km = buf[j];
            if (!idx->kmer_idx[km]) idx->kmer_idx[km] = so_vlimap_new(n_refs);
            so_vlimap_push_back(idx->kmer_idx[km], i);
        }
    }
    free(buf);
    for (uint32_t i = 0; i < idx->n_kmers; i++) { /* :260-271 */
        if (idx->kmer_idx[i] && idx->kmer_idx[i]->size > n_refs / 2) so_vlimap_invert(idx->kmer_idx[i]);
    }
    return idx;
}
void so_index_free(so_index *idx) {
    if (!idx) return;
    for (uint32_t i = 0; i < idx->n_kmers; i++) so_vlimap_free(idx->kmer_idx[i]);
    free(idx->kmer_idx);
    free(idx);
}
uint32_t so_index_size(const so_index *idx) { return idx->n_sequences; }

/* kmer_search.cpp:373-409 */
void so_index_scores(const so_index *idx, const so_cseq *query, int16_t *scores) {
    memset(scores, 0, sizeof(int16_t) * idx->n_sequences);
    uint32_t *buf = (uint32_t *)malloc(sizeof(uint32_t) * (query->n + 1));
    uint32_t nk = idx->nofast ? so_kmers(query->ab, query->n, idx->k, 0, 0, 0, buf)
                              : so_kmers(query->ab, query->n, idx->k, 1, 0, 0, buf);
    int offset = 0;
    for (uint32_t j = 0; j < nk; j++) {
        const so_vlimap *v = idx->kmer_idx[buf[j]];
        if (v) offset += so_vlimap_increment(v, scores);
    }
    free(buf);
    for (uint32_t i = 0; i < idx->n_sequences; i++) scores[i] = (int16_t)(scores[i] + offset);
}

typedef struct {
    int16_t first;
    int second;
} rank_pair;
/* std::greater<std::pair<int16_t,int>> */
static int rank_cmp(const void *a, const void *b) {
    const rank_pair *x = (const rank_pair *)a, *y = (const rank_pair *)b;
    if (x->first != y->first) return x->first > y->first ? -1 : 1;
    if (x->second != y->second) return x->second > y->second ? -1 : 1;
    return 0;
}
/* kmer_search.cpp:366-420; partial_sort's first `max` elements equal the first
 * `max` of a full sort because the order is total (ids are distinct). */
uint32_t so_index_find(const so_index *idx, const so_cseq *query, uint32_t max, uint32_t *out_ids,
                       float *out_scores) {
    uint32_t n = idx->n_sequences;
    if (max > n) max = n;
    if (max == 0) return 0;
    int16_t *scores = (int16_t *)malloc(sizeof(int16_t) * n);
    rank_pair *ranks = (rank_pair *)malloc(sizeof(rank_pair) * n);
    so_index_scores(idx, query, scores);
    for (uint32_t i = 0; i < n; i++) {
        ranks[i].first = scores[i];
        ranks[i].second = (int)i;
    }
    qsort(ranks, n, sizeof(rank_pair), rank_cmp);
    for (uint32_t i = 0; i < max; i++) {
        out_ids[i] = (uint32_t)ranks[i].second;
        out_scores[i] = (float)ranks[i].first;
    }
    free(scores);
    free(ranks);
    return max;
}

/* ---- .sidx (kmer_search.cpp:66-88: idx_header is {u64 magic; u16 vers; [2 pad]; u32 n_sequences;
 * u16 flags = k | nofast << 8; [6 pad]} = 24 bytes, the padding is whatever the stack held) */
static void vlimap_write(const so_vlimap *v, FILE *f) { /* idset.h:390-398 */
    uint32_t head[4] = {(uint32_t)(int32_t)v->inc, v->last, (uint32_t)v->nbytes, (uint32_t)v->size};
    fwrite(head, 4, 4, f);
    fwrite(v->data, 1, v->nbytes, f);
}
static so_vlimap *vlimap_read(uint32_t maxsize, FILE *f) { /* :400-409 */
    uint32_t head[4];
    if (fread(head, 4, 4, f) != 4) return NULL;
    so_vlimap *v = so_vlimap_new(maxsize);
    v->inc = (int16_t)head[0];
    v->last = head[1];
    v->size = head[3];
    v->nbytes = head[2];
    v->cap = head[2] ? head[2] : 1;
    free(v->data);
    v->data = (uint8_t *)malloc(v->cap);
    if (fread(v->data, 1, v->nbytes, f) != v->nbytes) {
        so_vlimap_free(v);
        return NULL;
    }
    return v;
}
int so_index_write(const so_index *idx, const char *const *names, const char *path) { /* :279-304 */
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    unsigned char header[24];
    memset(header, 0, sizeof header);
    const uint64_t magic = 0x5844494b414e4953ull; /* SINAKIDX */
    const uint16_t vers = 0, flags = (uint16_t)((idx->k & 0xff) | ((idx->nofast ? 1 : 0) << 8));
    memcpy(header, &magic, 8);
    memcpy(header + 8, &vers, 2);
    memcpy(header + 12, &idx->n_sequences, 4);
    memcpy(header + 16, &flags, 2);
    fwrite(header, 1, 24, f);
    for (uint32_t i = 0; i < idx->n_sequences; i++) fprintf(f, "%s\n", names[i]);
    so_vlimap *emptymap = so_vlimap_new(idx->n_sequences);
    for (uint32_t i = 0; i < idx->n_kmers; i++)
        if (idx->kmer_idx[i] && idx->kmer_idx[i]->size > 0) so_vlimap_push_back(emptymap, i);
    vlimap_write(emptymap, f);
    size_t it = 0, idxno = 0;
    while (it < emptymap->nbytes) {
        idxno += vl_decode(emptymap, &it);
        vlimap_write(idx->kmer_idx[idxno], f);
    }
    so_vlimap_free(emptymap);
    fclose(f);
    return 0;
}
so_index *so_index_read(const char *path, unsigned k, int nofast) { /* :306-351 */
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    unsigned char header[24];
    uint64_t magic;
    uint16_t vers, flags;
    uint32_t nseq;
    if (fread(header, 1, 24, f) != 24) goto bad;
    memcpy(&magic, header, 8);
    memcpy(&vers, header + 8, 2);
    memcpy(&nseq, header + 12, 4);
    memcpy(&flags, header + 16, 2);
    if (magic != 0x5844494b414e4953ull || vers != 0 || (flags & 0xff) != k || ((flags >> 8) & 1) != (nofast ? 1 : 0))
        goto bad;
    {
        so_index *idx = (so_index *)calloc(1, sizeof(so_index));
        idx->k = k;
        idx->n_kmers = 1u << (k * 2);
        idx->n_sequences = nseq;
        idx->nofast = nofast;
        idx->kmer_idx = (so_vlimap **)calloc(idx->n_kmers, sizeof(so_vlimap *));
        for (uint32_t i = 0; i < nseq; i++) { /* getline per name */
            int ch;
            while ((ch = fgetc(f)) != EOF && ch != '\n') {}
        }
        so_vlimap *emptymap = vlimap_read(nseq, f);
        if (!emptymap) {
            so_index_free(idx);
            goto bad;
        }
        size_t it = 0, idxno = 0;
        while (it < emptymap->nbytes) {
            idxno += vl_decode(emptymap, &it);
            idx->kmer_idx[idxno] = vlimap_read(nseq, f);
        }
        so_vlimap_free(emptymap);
        fclose(f);
        return idx;
    }
bad:
    fclose(f);
    return NULL;
}

/* plain CSR of "ref r contains k-mer v" (SURVEY Appendix A.1): inverted lists
 * are expanded back.  offsets has n_kmers+1 entries. */
uint64_t so_index_csr(const so_index *idx, uint32_t *offsets, uint32_t *ids) {
    uint64_t total = 0;
    uint8_t *present = (uint8_t *)malloc(idx->n_sequences ? idx->n_sequences : 1);
    for (uint32_t km = 0; km < idx->n_kmers; km++) {
        if (offsets) offsets[km] = (uint32_t)total;
        const so_vlimap *v = idx->kmer_idx[km];
        if (!v) continue;
        memset(present, v->inc < 0 ? 1 : 0, idx->n_sequences);
        size_t it = 0;
        uint32_t last = 0;
        while (it < v->nbytes) {
            last += vl_decode(v, &it);
            present[last] = v->inc < 0 ? 0 : 1;
        }
        for (uint32_t r = 0; r < idx->n_sequences; r++) {
            if (present[r]) {
                if (ids) ids[total] = r;
                total++;
            }
        }
    }
    if (offsets) offsets[idx->n_kmers] = (uint32_t)total;
    free(present);
    return total;
}

/* ----------------------------------------------------------- famfinder */

void so_ff_opts_default(so_ff_opts *o) { /* famfinder.cpp:156-203 */
    o->fs_min = 40;
    o->fs_max = 40;
    o->fs_msc = .7f;
    o->fs_msc_max = 2;
    o->fs_leave_query_out = 0;
    o->fs_req = 1;
    o->fs_req_full = 1;
    o->fs_full_len = 1400;
    o->fs_req_gaps = 10;
    o->fs_min_len = 150;
    o->fs_cover_gene = 0;
}

/* famfinder.cpp:344-378 (turn_check): top-1 k-mer score of the query as it is [0], reversed [1],
 * complemented [2], reversed and complemented [3]; [1] and [2] are only searched with --turn=all
 * (else 0).  The orientation with the strictly largest score wins, the first one on ties, "as it
 * is" when every search comes back empty.  cseq::reverse() also mirrors the columns; the k-mer
 * search only reads the base order. */
int so_turn_check(const so_index *idx, const so_cseq *query, int all, float *scores4) {
    double score[4];
    uint32_t id;
    float sc;
    score[0] = so_index_find(idx, query, 1, &id, &sc) ? sc : 0;
    so_cseq *turn = so_cseq_clone(query);
    so_cseq_reverse(turn);
    if (all) {
        score[1] = so_index_find(idx, turn, 1, &id, &sc) ? sc : 0;
        so_cseq *comp = so_cseq_clone(query);
        so_cseq_complement(comp);
        score[2] = so_index_find(idx, comp, 1, &id, &sc) ? sc : 0;
        so_cseq_free(comp);
    } else {
        score[1] = score[2] = 0;
    }
    so_cseq_complement(turn);
    score[3] = so_index_find(idx, turn, 1, &id, &sc) ? sc : 0;
    so_cseq_free(turn);
    double max = 0;
    int best = 0;
    for (int i = 0; i < 4; i++) {
        if (scores4) scores4[i] = (float)score[i];
        if (max < score[i]) {
            max = score[i];
            best = i;
        }
    }
    return best;
}

/* famfinder.cpp:497-612 (match) then :472-491 (gap filter, fs_req).
 * remove_similar (:545-548) needs cseq_comparator and is a provable no-op for
 * fs_msc_max >= 1 (identity <= 1, cseq_comparator.cpp:280); smaller values are
 * outside this oracle. remove_superstring is dead (noid=false, :503). */
uint32_t so_famfinder(const so_index *idx, const so_cseq *const *refs, const so_cseq *query,
                      const so_ff_opts *o, uint32_t *out_ids, float *out_scores, uint32_t cap,
                      so_log *log) {
    const uint32_t min_match = o->fs_min, max_match = o->fs_max;
    const float min_score = o->fs_msc;
    const uint32_t min_len = o->fs_min_len, num_full = o->fs_req_full;
    const uint32_t full_min_len = o->fs_full_len, range_cover = o->fs_cover_gene;
    const size_t range_begin = 0, range_end = 0;
    size_t have = 0, have_full = 0, have_cover_left = 0, have_cover_right = 0;
    uint32_t n = idx->n_sequences;
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    float *scores = (float *)malloc(sizeof(float) * (n ? n : 1));
    uint32_t nres = 0, from = 0;

    size_t max_results = (size_t)max_match + 1;
    while (have < max_match || have_full < num_full || have_cover_left < range_cover ||
           have_cover_right < range_cover) {
        nres = so_index_find(idx, query, (uint32_t)(max_results > n ? n : max_results), ids, scores);
        if (nres == 0) {
            free(ids);
            free(scores);
            goto tail_empty;
        }
        have = have_full = have_cover_left = have_cover_right = 0;
        from = 0; /* std::remove_if: stable compaction, predicate once per element in order */
        for (uint32_t i = 0; i < nres; i++) {
            const so_cseq *seq = refs[ids[i]];
            int is_full = seq->n >= full_min_len;
            int is_range_left = seq->n && SO_POS(seq->ab[0]) <= range_begin;
            int is_range_right = seq->n && SO_POS(seq->ab[seq->n - 1]) >= range_end;
            int rm = 0;
            if (seq->n < min_len) rm = 1;                                         /* remove_short */
            else if (o->fs_leave_query_out && strcmp(query->name, seq->name) == 0) rm = 1;
            else {
                int min_reached = have >= min_match;
                int max_reached = have >= max_match;
                int score_good = scores[i] < min_score; /* sic, :565-567 */
                int adds_to_full = num_full && have_full < num_full && is_full;
                int adds_to_range = (range_cover && have_cover_right < range_cover && is_range_right) ||
                                    (range_cover && have_cover_left < range_cover && is_range_left);
                if (min_reached && (max_reached || !score_good) && !adds_to_full && !adds_to_range) rm = 1;
            }
            if (!rm) { /* count_good */
                ++have;
                if (num_full && is_full) ++have_full;
                if (range_cover && is_range_right) ++have_cover_right;
                if (range_cover && is_range_left) ++have_cover_left;
                ids[from] = ids[i];
                scores[from] = scores[i];
                from++;
            }
        }
        if (max_results >= n) break;
        max_results *= 10;
    }
    nres = from;

    /* famfinder.cpp:472-480 too_few_gaps */
    if (o->fs_req_gaps != 0) {
        uint32_t w = 0;
        for (uint32_t i = 0; i < nres; i++) {
            const so_cseq *seq = refs[ids[i]];
            int drop = seq->n == 0 ||
                       (uint32_t)(SO_POS(seq->ab[seq->n - 1]) - seq->n + 1) < o->fs_req_gaps;
            if (!drop) {
                ids[w] = ids[i];
                scores[w] = scores[i];
                w++;
            }
        }
        nres = w;
    }
    if (nres < o->fs_req) { /* :486-491 */
        so_logf(log, "unable to align: too few relatives (%u);", nres);
        nres = 0;
    }
    if (nres > cap) nres = cap;
    memcpy(out_ids, ids, sizeof(uint32_t) * nres);
    memcpy(out_scores, scores, sizeof(float) * nres);
    free(ids);
    free(scores);
    return nres;
tail_empty:
    if (0 < o->fs_req) so_logf(log, "unable to align: too few relatives (%u);", 0u);
    return 0;
}

/* ---------------------------------------------------------------- mseq */

typedef struct {
    uint32_t a, b;
} edge_t;
static int edge_cmp_ba(const void *x, const void *y) { /* by (b, a) */
    const edge_t *e = (const edge_t *)x, *f = (const edge_t *)y;
    if (e->b != f->b) return e->b < f->b ? -1 : 1;
    if (e->a != f->a) return e->a < f->a ? -1 : 1;
    return 0;
}
static int edge_cmp_ab(const void *x, const void *y) {
    const edge_t *e = (const edge_t *)x, *f = (const edge_t *)y;
    if (e->a != f->a) return e->a < f->a ? -1 : 1;
    if (e->b != f->b) return e->b < f->b ? -1 : 1;
    return 0;
}

/* mseq.cpp:47-118 + dag::insert/link (graph.h:332-357) + reduce_edges (:466-488).
 * Node ids are insertion order; list::sort by position is a stable no-op because
 * nodes are inserted in column order. */
so_graph *so_mseq_build(const so_cseq *const *fam, uint32_t F, float weight) {
    uint32_t bases_width = 0;
    if (F) bases_width = fam[0]->width;
    for (uint32_t j = 0; j < F; j++)
        if (fam[j]->width != bases_width) return NULL; /* runtime_error, mseq.cpp:56-65 */

    size_t total = 0;
    for (uint32_t j = 0; j < F; j++) total += fam[j]->n;
    so_graph *g = (so_graph *)calloc(1, sizeof(so_graph));
    g->width = bases_width;
    g->pos = (uint32_t *)malloc(sizeof(uint32_t) * (total + 1));
    g->mask = (uint8_t *)malloc(total + 1);
    g->weight = (float *)malloc(sizeof(float) * (total + 1));
    edge_t *edges = (edge_t *)malloc(sizeof(edge_t) * (total + 1));
    size_t n_edges = 0;

    uint32_t *cit = (uint32_t *)calloc(F ? F : 1, sizeof(uint32_t));
    long *last = (long *)malloc(sizeof(long) * (F ? F : 1));
    for (uint32_t j = 0; j < F; j++) last[j] = -1;
    long nodes[256];
    uint32_t min_next = 0;

    for (uint32_t i = 0; i < bases_width; i++) {
        if (min_next > i) continue;
        min_next = 0x7fffffff; /* numeric_limits<int>::max() */
        for (int c = 0; c < 256; c++) nodes[c] = -1;
        for (uint32_t j = 0; j < F; j++) {
            if (cit[j] != fam[j]->n && SO_POS(fam[j]->ab[cit[j]]) == i) {
                long newnode;
                unsigned char base = (unsigned char)so_mask_to_rna(SO_MASK(fam[j]->ab[cit[j]]));
                if (nodes[base] < 0) {
                    newnode = (long)g->n;
                    g->pos[g->n] = i;
                    g->mask[g->n] = SO_MASK(fam[j]->ab[cit[j]]);
                    g->weight[g->n] = 1.f;
                    g->n++;
                    nodes[base] = newnode;
                } else {
                    newnode = nodes[base];
                    g->weight[newnode] += 1.f;
                }
                if (last[j] >= 0) {
                    edges[n_edges].a = (uint32_t)last[j];
                    edges[n_edges].b = (uint32_t)newnode;
                    n_edges++;
                }
                last[j] = newnode;
                ++cit[j];
            }
            if (cit[j] != fam[j]->n) {
                uint32_t p = SO_POS(fam[j]->ab[cit[j]]);
                if (p < min_next) min_next = p;
            }
        }
        for (int c = 0; c < 256; c++) {
            if (nodes[c] >= 0) {
                float nw = g->weight[nodes[c]];
                /* mseq.cpp:113: double reciprocal + float product, rounded to float */
                g->weight[nodes[c]] = (float)(1.0 / (double)(weight + 1) + (double)(weight * (nw / (float)F)));
            }
        }
    }
    free(cit);
    free(last);

    /* reduce_edges: per-node sorted unique pred / succ lists */
    uint32_t N = g->n;
    g->pred_off = (uint32_t *)calloc(N + 1, sizeof(uint32_t));
    g->succ_off = (uint32_t *)calloc(N + 1, sizeof(uint32_t));
    g->pred = (uint32_t *)malloc(sizeof(uint32_t) * (n_edges + 1));
    g->succ = (uint32_t *)malloc(sizeof(uint32_t) * (n_edges + 1));
    qsort(edges, n_edges, sizeof(edge_t), edge_cmp_ba);
    {
        size_t w = 0;
        for (size_t e = 0; e < n_edges; e++) {
            if (e && edges[e].a == edges[e - 1].a && edges[e].b == edges[e - 1].b) continue;
            g->pred[w++] = edges[e].a;
            g->pred_off[edges[e].b + 1]++;
        }
    }
    qsort(edges, n_edges, sizeof(edge_t), edge_cmp_ab);
    {
        size_t w = 0;
        for (size_t e = 0; e < n_edges; e++) {
            if (e && edges[e].a == edges[e - 1].a && edges[e].b == edges[e - 1].b) continue;
            g->succ[w++] = edges[e].b;
            g->succ_off[edges[e].a + 1]++;
        }
    }
    for (uint32_t m = 0; m < N; m++) {
        g->pred_off[m + 1] += g->pred_off[m];
        g->succ_off[m + 1] += g->succ_off[m];
    }
    free(edges);
    /* sentinel lists (graph.h:301-327): sources = no preds, sinks = no succs */
    g->src = (uint32_t *)malloc(sizeof(uint32_t) * (N + 1));
    g->snk = (uint32_t *)malloc(sizeof(uint32_t) * (N + 1));
    for (uint32_t m = 0; m < N; m++) {
        if (g->pred_off[m + 1] == g->pred_off[m]) g->src[g->n_src++] = m;
        if (g->succ_off[m + 1] == g->succ_off[m]) g->snk[g->n_snk++] = m;
    }
    return g;
}

void so_graph_free(so_graph *g) {
    if (!g) return;
    free(g->pos);
    free(g->mask);
    free(g->weight);
    free(g->pred_off);
    free(g->pred);
    free(g->succ_off);
    free(g->succ);
    free(g->src);
    free(g->snk);
    free(g->prof);
    free(g);
}

/* ------------------------------------------------------------- pseq (--fs-no-graph) */

/* base_profile(const base_iupac&), pseq.h:65-86: the share 1 / ambig_order for every base the code stands for */
static void profile_of_base(int mask, float out[6]) {
    const int order = __builtin_popcount(mask & 0xf);
    for (int i = 0; i < 6; i++) out[i] = 0.f;
    if (order > 0) {
        const float val = 1.f / (float)order;
        if (mask & 1) out[0] = val; /* A */
        if (mask & 2) out[1] = val; /* G */
        if (mask & 4) out[2] = val; /* C */
        if (mask & 8) out[3] = val; /* T/U */
    }
}
/* base_profile::comp(const base_profile&, ...), pseq.h:100-113: sixteen products added up in i-outer, j-inner
 * order (float, no contraction), then the two gap terms */
static float profile_comp2(const float a[6], const float b[6], float match, float mismatch, float gap, float gap_ext) {
    float res = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            if (i == j) res += match * a[i] * b[j];
            else res += mismatch * a[i] * b[j];
        }
    return res + gap * a[4] + gap_ext * a[5];
}
float so_profile_comp(const float *prof, int smask, float match, float mismatch, float gap, float gap_ext) {
    float b[6];
    profile_of_base(smask, b);
    return profile_comp2(prof ? prof : b, b, match, mismatch, gap, gap_ext);
}

/* pseq::pseq, pseq.cpp:41-112: walk the alignment's occupied columns (column 0 first, occupied or not);
 * per column every family member either shows a base -- 12 / ambig_order points for each base it may
 * be -- or is in a gap, which counts as opened (its previous column had a base) or extended */
so_graph *so_pseq_build(const so_cseq *const *fam, uint32_t F) {
    so_graph *g = (so_graph *)calloc(1, sizeof(so_graph));
    const uint32_t width = F ? fam[0]->width : 0;
    g->width = width;
    uint32_t *it = (uint32_t *)calloc(F ? F : 1, sizeof(uint32_t));
    uint8_t *gap = (uint8_t *)malloc(F ? F : 1);
    for (uint32_t i = 0; i < F; i++) gap[i] = 1;
    uint32_t cap = 1024, n = 0;
    g->pos = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    g->prof = (float *)malloc(sizeof(float) * 6 * cap);
    uint32_t current = 0;
    while (current < width) {
        uint32_t next = width;
        int A = 0, G = 0, Cc = 0, T = 0, gap_open = 0, gap_extend = 0;
        for (uint32_t row = 0; row < F; row++) {
            const so_cseq *s = fam[row];
            if (it[row] < s->n && SO_POS(s->ab[it[row]]) == current) {
                const int m = SO_MASK(s->ab[it[row]]);
                const int order = __builtin_popcount(m & 0xf);
                if (order > 0) {
                    const int points = 12 / order;
                    if (m & 1) A += points;
                    if (m & 2) G += points;
                    if (m & 4) Cc += points;
                    if (m & 8) T += points;
                    gap[row] = 0;
                }
                ++it[row];
            } else {
                if (gap[row]) ++gap_extend;
                else {
                    gap[row] = 1;
                    ++gap_open;
                }
            }
            if (it[row] < s->n && SO_POS(s->ab[it[row]]) < next) next = SO_POS(s->ab[it[row]]);
        }
        if (n == cap) {
            cap *= 2;
            g->pos = (uint32_t *)realloc(g->pos, sizeof(uint32_t) * cap);
            g->prof = (float *)realloc(g->prof, sizeof(float) * 6 * cap);
        }
        { /* base_profile(a, g, c, t, open * 12, extend * 12), pseq.h:54-63 */
            const int open = gap_open * 12, extend = gap_extend * 12;
            const int sum = A + G + Cc + T + open + extend;
            float *p = g->prof + 6 * (size_t)n;
            p[0] = (float)A / sum;
            p[1] = (float)G / sum;
            p[2] = (float)Cc / sum;
            p[3] = (float)T / sum;
            p[4] = (float)open / sum;
            p[5] = (float)extend / sum;
        }
        g->pos[n++] = current;
        current = next;
    }
    free(it);
    free(gap);
    g->n = n;
    /* a chain: node m follows node m - 1 (pseq.h prev_begin / next_begin); first and last are the only
     * source and sink (pn_first_* / pn_last_*) */
    g->mask = (uint8_t *)calloc(n ? n : 1, 1);
    g->weight = (float *)calloc(n ? n : 1, sizeof(float));
    g->pred_off = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    g->succ_off = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    g->pred = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    g->succ = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    g->pred_off[0] = g->succ_off[0] = 0;
    for (uint32_t m = 0; m < n; m++) {
        g->pred_off[m + 1] = g->pred_off[m] + (m > 0 ? 1u : 0u);
        if (m > 0) g->pred[g->pred_off[m]] = m - 1;
        g->succ_off[m + 1] = g->succ_off[m] + (m + 1 < n ? 1u : 0u);
        if (m + 1 < n) g->succ[g->succ_off[m]] = m + 1;
    }
    g->n_src = g->n_snk = n ? 1 : 0;
    g->src = (uint32_t *)malloc(sizeof(uint32_t));
    g->snk = (uint32_t *)malloc(sizeof(uint32_t));
    g->src[0] = 0;
    g->snk[0] = n ? n - 1 : 0;
    return g;
}

/* ------------------------------------------------------------- mesh DP */

void so_align_opts_default(so_align_opts *o) { /* align.cpp:231-274 */
    memset(o, 0, sizeof(*o));
    o->match_score = 2;
    o->mismatch_score = -1;
    o->gap_penalty = 5.0f;
    o->gap_ext_penalty = 2.0f;
    o->fs_weight = 1;
    o->overhang = SO_OVERHANG_ATTACH;
    o->lowercase = SO_LOWERCASE_NONE;
    o->insertion = SO_INSERTION_SHIFT;
    o->realign = 0;
}

/* scoring_scheme_simple (scoring_schemes.h:102-164) and _weighted (:166-241).
 * ms/mms/gp/gpe are the already negated/pos. values handed to the scheme ctor
 * at align.cpp:406-414. Out-of-range posvar indices are UB in the reference;
 * here they clamp to the last weight (documented guard). */
typedef struct {
    float ms, mms, gp, gpe;
    const float *w;
    uint32_t nw;
} scheme;
/* scoring_scheme_profile::match (scoring_schemes.h:84-92): prev + comp(); its gap costs are the
 * constant ones of the simple scheme (:47-82) */
static float s_match_profile(const scheme *s, float prev, const float *prof, uint8_t smask) {
    return prev + so_profile_comp(prof, smask, s->ms, s->mms, s->gp, s->gpe);
}
static float sw(const scheme *s, uint32_t i) { return s->w[i < s->nw ? i : s->nw - 1]; }
static float s_insertion(const scheme *s, float prev, uint32_t mpos) {
    if (!s->w) return prev + s->gp;
    return prev + s->gp * sw(s, mpos + 1);
}
static float s_insertion_ext(const scheme *s, float prev, uint32_t mpos, int offset) {
    if (!s->w) return prev + s->gpe;
    return prev + s->gpe * sw(s, (uint32_t)((int)mpos + 1 + offset));
}
static float s_deletion(const scheme *s, float prev, uint32_t mpos) {
    if (!s->w) return prev + s->gp;
    return prev + s->gp * sw(s, mpos);
}
static float s_deletion_ext(const scheme *s, float prev, uint32_t mpos) {
    if (!s->w) return prev + s->gpe;
    return prev + s->gpe * sw(s, mpos);
}
static float s_match(const scheme *s, float prev, uint8_t mmask, uint8_t smask, uint32_t mpos, float mweight) {
    int comp = (0xf & mmask & smask) != 0; /* aligned_base.h:153-156 */
    if (!s->w) return prev + (comp ? s->ms : s->mms) * mweight;
    return prev + (comp ? s->ms : s->mms) * sw(s, mpos) * mweight;
}
static void scheme_from_opts(scheme *s, const so_align_opts *o) {
    s->ms = -o->match_score;
    s->mms = -o->mismatch_score;
    s->gp = o->gap_penalty;
    s->gpe = o->gap_ext_penalty;
    s->w = (o->weights && o->n_weights && !o->fs_no_graph) ? o->weights : NULL;
    s->nw = o->n_weights;
}

/* single scoring op, for cross-checks with the real scoring_schemes.h (oracle/_ref):
 * op 0 insertion, 1 insertion_ext, 2 deletion, 3 deletion_ext, 4 match */
float so_score_op(int op, float prev, uint32_t mpos, int mmask, float mweight, int smask, int offset,
                  float ms, float mms, float gp, float gpe, const float *weights, uint32_t nw) {
    scheme s;
    s.ms = ms; s.mms = mms; s.gp = gp; s.gpe = gpe;
    s.w = (weights && nw) ? weights : NULL;
    s.nw = nw;
    switch (op) {
    case 0: return s_insertion(&s, prev, mpos);
    case 1: return s_insertion_ext(&s, prev, mpos, offset);
    case 2: return s_deletion(&s, prev, mpos);
    case 3: return s_deletion_ext(&s, prev, mpos);
    default: return s_match(&s, prev, (uint8_t)mmask, (uint8_t)smask, mpos, mweight);
    }
}

/* mesh.h:455-502 compute_node_simple::calc with transition_simple (:307-374) or
 * transition_aspace_aware::insertion (:403-437); mesh.h:512-528 compute(). */
void so_mesh_compute(const so_graph *g, const uint32_t *q, uint32_t L, const so_align_opts *o,
                     so_cell *cells) {
    scheme s;
    scheme_from_opts(&s, o);
    const int forbid = (o->insertion == SO_INSERTION_FORBID);
    for (uint32_t m = 0; m < g->n; m++) {
        const uint32_t pb = g->pred_off[m], pe = g->pred_off[m + 1];
        const uint32_t mpos = g->pos[m];
        unsigned int min_mpos = 1000000;
        for (uint32_t e = g->succ_off[m]; e < g->succ_off[m + 1]; e++)
            if (g->pos[g->succ[e]] < min_mpos) min_mpos = g->pos[g->succ[e]];
        const int max_insert = (int)(min_mpos - mpos - 1);
        for (uint32_t si = 0; si < L; si++) {
            so_cell d;
            if (pb == pe || si == 0) { /* init_edge */
                d.value = d.gapm_val = d.gaps_val = 1;
            } else { /* init */
                d.value = d.gapm_val = d.gaps_val = 1000000;
            }
            d.value_midx = d.value_sidx = d.gapm_idx = d.gaps_idx = 0;
            d.gaps_max = 0;

            for (uint32_t e = pb; e < pe; e++) { /* deletion, mesh.h:307-330 */
                uint32_t mi = g->pred[e];
                const so_cell *src = &cells[(size_t)mi * L + si];
                float value = s_deletion(&s, src->value, mpos);
                float gap_val = s_deletion_ext(&s, src->gapm_val, mpos);
                uint32_t midx = mi;
                if (value < gap_val) {
                    d.gapm_val = value;
                    d.gapm_idx = midx;
                } else {
                    d.gapm_val = gap_val;
                    d.gapm_idx = src->gapm_idx;
                    value = gap_val;
                    midx = src->gapm_idx;
                }
                if (value < d.value) {
                    d.value = value;
                    d.value_midx = midx;
                    d.value_sidx = si;
                }
            }
            if (si > 0) { /* insertion from (m, s-1) */
                const uint32_t sidx = si - 1;
                const so_cell *src = &cells[(size_t)m * L + sidx];
                int done = 1;
                if (!forbid) { /* mesh.h:332-358 */
                    if (src->gaps_val != src->value) {
                        d.gaps_val = s_insertion(&s, src->value, mpos);
                        d.gaps_idx = sidx;
                    } else {
                        d.gaps_val = s_insertion_ext(&s, src->gaps_val, mpos, (int)(sidx - src->gaps_idx));
                        d.gaps_idx = src->gaps_idx;
                    }
                } else { /* mesh.h:403-437; smax is unsigned (slave idx_type) */
                    unsigned int smax = (unsigned int)max_insert;
                    if (smax < 1) {
                        done = 0;
                    } else if (src->gaps_val != src->value) {
                        d.gaps_val = s_insertion(&s, src->value, mpos);
                        d.gaps_idx = sidx;
                        d.gaps_max = smax - 1;
                    } else if (src->gaps_max > 0) {
                        d.gaps_val = s_insertion_ext(&s, src->gaps_val, mpos, (int)(sidx - src->gaps_idx));
                        d.gaps_idx = src->gaps_idx;
                        d.gaps_max = src->gaps_max - 1;
                    } else {
                        done = 0;
                    }
                }
                if (done && d.gaps_val <= d.value) {
                    d.value = d.gaps_val;
                    d.value_sidx = d.gaps_idx;
                    d.value_midx = m;
                }
                for (uint32_t e = pb; e < pe; e++) { /* match, mesh.h:360-374 */
                    uint32_t mi = g->pred[e];
                    const so_cell *msrc = &cells[(size_t)mi * L + sidx];
                    float value = g->prof ? s_match_profile(&s, msrc->value, g->prof + 6 * (size_t)m, SO_MASK(q[si]))
                                          : s_match(&s, msrc->value, g->mask[m], SO_MASK(q[si]), mpos, g->weight[m]);
                    if (value < d.value) {
                        d.value = value;
                        d.value_midx = mi;
                        d.value_sidx = sidx;
                    }
                }
            }
            cells[(size_t)m * L + si] = d;
        }
    }
}

/* --------------------------------------------------------------- backtrack */

static int in_set(const uint32_t *set, uint32_t n, uint32_t v) {
    for (uint32_t i = 0; i < n; i++)
        if (set[i] == v) return 1;
    return 0;
}

/* mesh.h:535-739 */
float so_backtrack(const so_graph *g, const uint32_t *q, uint32_t L, const so_cell *cells,
                   const so_align_opts *o, so_cseq *out, int *cutoff_head, int *cutoff_tail,
                   so_log *log) {
#define MESH(mi, si) cells[(size_t)(mi) * L + (si)]
    scheme sch;
    scheme_from_opts(&sch, o);
    const uint32_t alig_width = g->width;
    const uint32_t sbegin = 0, send = L - 1;

    /* starting point, :567-592 */
    uint32_t m = g->snk[0];
    for (uint32_t tmp = 0; tmp < g->n; tmp++) {
        if (MESH(tmp, send).value < MESH(m, send).value) m = tmp;
    }
    uint32_t s = send;
    for (uint32_t k = 0; k < g->n_snk; k++) {
        uint32_t mtmp = g->snk[k];
        for (uint32_t stmp = 0; stmp < L; stmp++) {
            if (MESH(mtmp, stmp).value < MESH(m, s).value) {
                m = mtmp;
                s = stmp;
            }
        }
    }

    /* right hand overhang, :594-615 */
    *cutoff_tail = (int)(send - s);
    if (*cutoff_tail && o->overhang != SO_OVERHANG_REMOVE) {
        int pos;
        if (o->overhang == SO_OVERHANG_ATTACH) {
            pos = (int)(alig_width - 1 - g->pos[m] - (uint32_t)*cutoff_tail);
        } else {
            pos = 0;
        }
        for (int i = 0; i < *cutoff_tail; i++) { /* slave.rbegin() .. +cutoff_tail */
            uint8_t b = SO_MASK(q[L - 1 - (uint32_t)i]);
            if (o->lowercase == SO_LOWERCASE_UNALIGNED) b |= 16;
            int p = pos++;
            so_cseq_append_base(out, SO_AB(p > 0 ? p : 0, b), NULL);
        }
    }

    float rval = MESH(m, s).value; /* :618 */
    unsigned int pos = alig_width - 1 - g->pos[m];
    float sum_weight = 0;
    int aligned_bases = 0;

    so_cseq_append_base(out, SO_AB(pos, SO_MASK(q[s])), NULL); /* :626-628 */
    aligned_bases++;
    /* :631-638: master node copy with the slave's base => comp() is true */
    sum_weight = g->prof ? s_match_profile(&sch, sum_weight, NULL, SO_MASK(q[s]))
                         : s_match(&sch, sum_weight, SO_MASK(q[s]), SO_MASK(q[s]), g->pos[m], g->weight[m]);

    while (s != sbegin && !in_set(g->src, g->n_src, m)) { /* :642-685 */
        uint32_t snew = MESH(m, s).value_sidx;
        m = MESH(m, s).value_midx;
        if (snew == MESH(m, snew).value_sidx && snew != 0) {
            m = MESH(m, snew).value_midx;
        }
        pos = alig_width - 1 - g->pos[m];
        while (s != snew) {
            --s;
            so_cseq_append_base(out, SO_AB(pos, SO_MASK(q[s])), NULL);
            aligned_bases++;
            sum_weight = g->prof ? s_match_profile(&sch, sum_weight, NULL, SO_MASK(q[s]))
                                 : s_match(&sch, sum_weight, SO_MASK(q[s]), SO_MASK(q[s]), g->pos[m], g->weight[m]);
        }
    }

    if (s != sbegin) { /* :690-721 */
        *cutoff_head = (int)(s - sbegin);
        switch (o->overhang) {
        case SO_OVERHANG_ATTACH:
            while (s-- != sbegin) {
                uint8_t b = SO_MASK(q[s]);
                ++pos;
                uint32_t p = (alig_width - 1 < pos) ? alig_width - 1 : pos;
                if (o->lowercase == SO_LOWERCASE_UNALIGNED) b |= 16;
                so_cseq_append_base(out, SO_AB(p, b), NULL);
            }
            break;
        case SO_OVERHANG_REMOVE:
            break;
        case SO_OVERHANG_EDGE: {
            int n = (int)(s - sbegin);
            while (n--) {
                uint8_t b = SO_MASK(q[n]);
                if (o->lowercase == SO_LOWERCASE_UNALIGNED) b |= 16;
                so_cseq_append_base(out, SO_AB(alig_width - (uint32_t)n - 1, b), NULL);
            }
            break;
        }
        }
    } else {
        *cutoff_head = 0;
    }

    if (so_cseq_set_width(out, alig_width) < 0) return -1e30f; /* :723 */
    so_cseq_reverse(out);
    if (so_cseq_fix_duplicate_positions(out, log, o->lowercase == SO_LOWERCASE_UNALIGNED,
                                        o->insertion == SO_INSERTION_REMOVE) < 0)
        return -1e30f;
    if (out->width > alig_width) so_logf(log, "warning: result sequence too wide!");

    /* :733-736 -- default ostream float formatting == %g with 6 significant digits */
    so_logf(log, "scoring: raw=%g, weight=%g, query-len=%u, aligned-bases=%d, score=%g; ", (double)rval,
            (double)sum_weight, L, aligned_bases, (double)(rval / sum_weight));
    return rval / sum_weight;
#undef MESH
}

/* ------------------------------------------------------------ align glue */

/* boost::algorithm::icontains / ifind_first on getBases() strings */
static long ifind(const char *hay, size_t hn, const char *needle, size_t nn) {
    if (nn == 0) return 0;
    if (nn > hn) return -1;
    for (size_t i = 0; i + nn <= hn; i++) {
        size_t j = 0;
        while (j < nn && toupper((unsigned char)hay[i + j]) == toupper((unsigned char)needle[j])) j++;
        if (j == nn) return (long)i;
    }
    return -1;
}

/* align.cpp:307-460 (operator()) + :462-521 (choose_transition, do_align) */
void so_align(const so_cseq *const *family, uint32_t F, const so_cseq *query, const so_align_opts *o,
              so_cseq *out, so_align_result *res, so_log *log) {
    memset(res, 0, sizeof(*res));
    so_cseq *c = so_cseq_clone(query);
    char *bases = (char *)malloc(query->n + 1);
    so_cseq_get_bases(c, bases);
    if (o->lowercase != SO_LOWERCASE_ORIGINAL) so_cseq_upper(c);

    const so_cseq **vc = (const so_cseq **)malloc(sizeof(so_cseq *) * (F ? F : 1));
    uint8_t *contains = (uint8_t *)malloc(F ? F : 1);
    char **rb = (char **)malloc(sizeof(char *) * (F ? F : 1));
    for (uint32_t i = 0; i < F; i++) {
        vc[i] = family[i];
        rb[i] = (char *)malloc(family[i]->n + 1);
        so_cseq_get_bases(family[i], rb[i]);
        contains[i] = ifind(rb[i], family[i]->n, bases, query->n) >= 0;
    }
    /* std::partition, libstdc++ bidirectional version (stl_algo.h __partition):
     * predicate = not_contains_query (align.cpp:329-333). Not stable. */
    uint32_t first = 0, last = F;
    for (;;) {
        for (;;) {
            if (first == last) goto part_done;
            else if (!contains[first]) ++first;
            else break;
        }
        --last;
        for (;;) {
            if (first == last) goto part_done;
            else if (contains[last]) --last;
            else break;
        }
        {
            const so_cseq *t = vc[first]; vc[first] = vc[last]; vc[last] = t;
            uint8_t tc = contains[first]; contains[first] = contains[last]; contains[last] = tc;
            char *tb = rb[first]; rb[first] = rb[last]; rb[last] = tb;
        }
        ++first;
    }
part_done:;
    uint32_t begin_containing = first;
    uint32_t nfam = F;

    if (begin_containing != F) {
        if (o->realign) { /* :337-348 */
            so_logf(log, "sequences ");
            for (uint32_t i = begin_containing; i < F; i++) so_logf(log, "%s ", vc[i]->name);
            so_logf(log, "containing exact candidate removed from family;");
            nfam = begin_containing;
            if (nfam == 0) {
                so_logf(log, "that's ALL of them. skipping sequence;");
                res->status = 2;
                goto done;
            }
        } else { /* :349-388 */
            long exact = -1;
            for (uint32_t i = begin_containing; i < F; i++) {
                if (vc[i]->n == query->n && ifind(rb[i], vc[i]->n, bases, query->n) == 0) {
                    exact = i;
                    break;
                }
            }
            if (exact >= 0) {
                so_cseq_set_data(c, vc[exact]->ab, vc[exact]->n, c->width);
                so_logf(log, "copied alignment from identical template sequence %s:%s; ", vc[exact]->name, "0");
            } else {
                const so_cseq *r = vc[begin_containing];
                long off = ifind(rb[begin_containing], r->n, bases, query->n);
                so_cseq_set_data(c, r->ab + off, query->n, c->width);
                so_logf(log, "copied alignment from (longer) template sequence %s:%s; ", r->name, "0");
            }
            so_cseq_set_width(c, vc[begin_containing]->width);
            so_cseq_set_data(out, c->ab, c->n, c->width);
            res->status = 1;
            res->idty = 100.f; /* :380-382 */
            res->qual = 100;
            res->head = res->tail = 0;
            res->score = 1.0f;
            goto done;
        }
    }

    {
        so_graph *g = o->fs_no_graph ? so_pseq_build(vc, nfam)                 /* :429 */
                                     : so_mseq_build(vc, nfam, o->fs_weight); /* :399-402 */
        if (!g) {
            res->status = -1;
            goto done;
        }
        uint32_t L = c->n;
        /* per-thread scratch, grown on demand and kept: the reference's mesh comes from
         * tbb's caching allocator (align.cpp:484-488), so a fresh 100+ MB mmap per query
         * would understate the CPU baseline */
        static __thread so_cell *tl_cells = NULL;
        static __thread size_t tl_cap = 0;
        if ((size_t)g->n * L > tl_cap) {
            free(tl_cells);
            tl_cap = (size_t)g->n * L + ((size_t)g->n * L >> 3);
            tl_cells = (so_cell *)malloc(sizeof(so_cell) * tl_cap);
        }
        so_cell *cells = tl_cells;
        so_mesh_compute(g, c->ab, L, o, cells); /* do_align :495 */
        so_cseq_clear(out);                     /* c.clearSequence() :498 */
        memcpy(out->name, c->name, sizeof(out->name));
        float score = so_backtrack(g, c->ab, L, cells, o, out, &res->head, &res->tail, log);
        res->cells = (uint64_t)g->n * L;
        so_graph_free(g);
        if (score == -1e30f) {
            res->status = -1;
            goto done;
        }
        res->score = score;
        { /* :509 */
            float v = 100.f * score;
            if (v < 0.f) v = 0.f;   /* std::max(0.f, v) */
            if (100.f < v) v = 100.f; /* std::min(100.f, .) */
            res->qual = (int)v;
        }
        { /* :443-453 calc_idty */
            float idty = 0;
            for (uint32_t i = 0; i < nfam; i++) {
                const float x = so_compare(out, vc[i], SO_CMP_IUPAC_OPTIMISTIC, SO_CMP_DIST_NONE, SO_CMP_COVER_OVERLAP, 0);
                if (idty < x) idty = x; /* std::max(idty, x) */
            }
            res->idty = 100.f * idty;
        }
        res->status = 0;
    }
done:
    for (uint32_t i = 0; i < F; i++) free(rb[i]);
    free(rb);
    free(contains);
    free(vc);
    free(bases);
    so_cseq_free(c);
}

/* ------------------------------------------------------------ cseq_comparator (section 8f-1) */

static int cmp_filtered(uint32_t ab, int filter_lc) { /* filter_lowercase / filter_none, :128-138 */
    return filter_lc && (((ab >> 24) & 0x10) != 0);      /* aligned_base isLowerCase: bit 4 of the mask byte */
}
static int cmp_bases(uint32_t a, uint32_t b, int iupac) { /* base_comp_*, :113-129; a = query base */
    const uint8_t ma = (uint8_t)((a >> 24) & 0xf), mb = (uint8_t)((b >> 24) & 0xf);
    switch (iupac) {
    case SO_CMP_IUPAC_OPTIMISTIC: return (ma & mb) != 0;                 /* aligned_base.h:153-155 */
    case SO_CMP_IUPAC_PESSIMISTIC: return !is_ambig(ma) && ma == mb;     /* :163-165 */
    default: return ma == mb;                                            /* :167-169 */
    }
}

/* traverse(), cseq_comparator.cpp:56-111, with match_counter::counter :165-206 */
void so_compare_counts(const so_cseq *A, const so_cseq *B, int iupac, int filter_lc, so_match_counts *m) {
    memset(m, 0, sizeof(*m));
    const uint32_t *a = A->ab, *a_end = A->ab + A->n;
    const uint32_t *b = B->ab, *b_end = B->ab + B->n;
#define POS(x) ((x) & 0xFFFFFFu)
    /* skip filtered bases at beginning */
    while (a != a_end && cmp_filtered(*a, filter_lc)) ++a;
    while (b != b_end && cmp_filtered(*b, filter_lc)) ++b;
    /* skip filtered bases at end */
    while (a != a_end && cmp_filtered(*(a_end - 1), filter_lc)) --a_end;
    while (b != b_end && cmp_filtered(*(b_end - 1), filter_lc)) --b_end;
    if (a == a_end || b == b_end) return; /* (the reference dereferences end() here) */
    /* count left overhang */
    if (POS(*a) < POS(*b)) {
        while (a != a_end && POS(*a) < POS(*b)) {
            if (!cmp_filtered(*a, filter_lc)) m->only_a_overhang++;
            ++a;
        }
    } else {
        while (b != b_end && POS(*a) > POS(*b)) {
            if (!cmp_filtered(*b, filter_lc)) m->only_b_overhang++;
            ++b;
        }
    }
    /* count overlapping zone */
    while (a != a_end && b != b_end) {
        const int diff = (int)POS(*a) - (int)POS(*b);
        if (diff > 0) {
            if (!cmp_filtered(*b, filter_lc)) m->only_b++;
            ++b;
        } else if (diff < 0) {
            if (!cmp_filtered(*a, filter_lc)) m->only_a++;
            ++a;
        } else {
            const int fa = cmp_filtered(*a, filter_lc), fb = cmp_filtered(*b, filter_lc);
            if (!fa && !fb) {
                if (cmp_bases(*a, *b, iupac)) m->match++;
                else m->mismatch++;
            } else if (!fa) {
                m->only_a++;
            } else if (!fb) {
                m->only_b++;
            }
            ++a;
            ++b;
        }
    }
    /* count right overhang */
    while (a != a_end) {
        if (!cmp_filtered(*a, filter_lc)) m->only_a_overhang++;
        ++a;
    }
    while (b != b_end) {
        if (!cmp_filtered(*b, filter_lc)) m->only_b_overhang++;
        ++b;
    }
#undef POS
}

/* cseq_comparator::operator() :240-296 */
float so_compare_score(const so_match_counts *m, int cover, int dist_rule) {
    int base;
    switch (cover) {
    case SO_CMP_COVER_ABS: base = 1; break;
    case SO_CMP_COVER_QUERY: base = m->match + m->mismatch + m->only_a + m->only_a_overhang; break;
    case SO_CMP_COVER_TARGET: base = m->match + m->mismatch + m->only_b + m->only_b_overhang; break;
    case SO_CMP_COVER_OVERLAP: base = m->match + m->mismatch + m->only_a + m->only_b; break;
    case SO_CMP_COVER_ALL:
        base = m->match + m->mismatch + m->only_a + m->only_b + m->only_a_overhang + m->only_b_overhang;
        break;
    case SO_CMP_COVER_AVERAGE:
        base = m->match + m->mismatch + (m->only_a + m->only_b + m->only_a_overhang + m->only_b_overhang) / 2;
        break;
    case SO_CMP_COVER_MIN: {
        const int x = m->only_a + m->only_a_overhang, y = m->only_b + m->only_b_overhang;
        base = m->match + m->mismatch + (x < y ? x : y);
        break;
    }
    case SO_CMP_COVER_MAX: {
        const int x = m->only_a + m->only_a_overhang, y = m->only_b + m->only_b_overhang;
        base = m->match + m->mismatch + (x > y ? x : y);
        break;
    }
    default: base = m->match + m->mismatch; break; /* NOGAP */
    }
    float dist = (float)m->match / base;
    if (dist_rule == SO_CMP_DIST_JC) dist = (float)(-3.0 / 4 * log(1.0 - 4.0 / 3 * dist)); /* jukes_cantor :42-44 */
    return dist;
}

float so_compare(const so_cseq *a, const so_cseq *b, int iupac, int dist, int cover, int filter_lc) {
    so_match_counts m;
    so_compare_counts(a, b, iupac, filter_lc, &m);
    return so_compare_score(&m, cover, dist);
}

/* ------------------------------------------------------------ search_filter (section 8f-1) */

void so_search_opts_default(so_search_opts *o) { /* search_filter.cpp:91-126, cseq_comparator.cpp:434-464 */
    o->kmer_candidates = 1000;
    o->max_result = 10;
    o->min_sim = .7f;
    o->lca_quorum = .7f;
    o->ignore_super = 0;
    o->search_all = 0;
    o->iupac = SO_CMP_IUPAC_OPTIMISTIC;
    o->dist = SO_CMP_DIST_NONE;
    o->cover = SO_CMP_COVER_QUERY;
    o->filter_lc = 0;
}

typedef struct {
    float score;
    uint32_t id;
    const char *name;
} search_item;
/* std::greater<result_item>: a > b == !(a < b), with a < b = score, then name (search.h:58-66) */
static int item_less(const search_item *a, const search_item *b) {
    if (a->score < b->score) return 1;
    if (a->score > b->score) return 0;
    return strcmp(a->name, b->name) < 0;
}
static int item_cmp_desc(const void *x, const void *y) {
    const search_item *a = (const search_item *)x, *b = (const search_item *)y;
    if (item_less(b, a)) return -1;
    if (item_less(a, b)) return 1;
    return 0;
}
/* boost::algorithm::contains(ref aligned bases, query aligned bases, a.comp(b)) :264-268 */
static int seq_contains(const so_cseq *hay, const so_cseq *needle) {
    if (needle->n == 0) return 1;
    if (needle->n > hay->n) return 0;
    for (uint32_t i = 0; i + needle->n <= hay->n; i++) {
        uint32_t j = 0;
        while (j < needle->n && (((hay->ab[i + j] >> 24) & (needle->ab[j] >> 24) & 0xf) != 0)) j++;
        if (j == needle->n) return 1;
    }
    return 0;
}

/* libstdc++'s std::partial_sort with std::greater<result_item> (bits/stl_heap.h, bits/stl_algo.h:
 * __heap_select + __sort_heap) and std::partition for bidirectional iterators, restated so that the
 * UNSORTED remainder they leave is the one a GCC build of the reference sees. */
static int item_gt(const search_item *a, const search_item *b) { return !item_less(a, b); }
static void gnu_push_heap(search_item *first, long hole, long top, search_item value) {
    long parent = (hole - 1) / 2;
    while (hole > top && item_gt(first + parent, &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
static void gnu_adjust_heap(search_item *first, long hole, long len, search_item value) {
    const long top = hole;
    long second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (item_gt(first + second, first + (second - 1))) second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    gnu_push_heap(first, hole, top, value);
}
static void gnu_pop_heap(search_item *first, search_item *last, search_item *result) {
    search_item value = *result;
    *result = *first;
    gnu_adjust_heap(first, 0, last - first, value);
}
static void gnu_partial_sort(search_item *first, search_item *middle, search_item *last) {
    const long len = middle - first;
    if (len >= 2) { /* __make_heap */
        long parent = (len - 2) / 2;
        for (;;) {
            search_item value = first[parent];
            gnu_adjust_heap(first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    for (search_item *i = middle; i < last; ++i) /* __heap_select */
        if (item_gt(i, first)) gnu_pop_heap(first, middle, i);
    while (middle - first > 1) { /* __sort_heap */
        --middle;
        gnu_pop_heap(first, middle, middle);
    }
}
static int seq_contains(const so_cseq *hay, const so_cseq *needle);
/* std::partition(first, last, contains_query); returns the number of elements in the first group */
static uint32_t gnu_partition_contains(search_item *first, search_item *last, const so_cseq *const *refs,
                                       const so_cseq *c) {
    search_item *const begin = first;
    for (;;) {
        for (;;) {
            if (first == last) return (uint32_t)(first - begin);
            else if (seq_contains(refs[first->id], c)) ++first;
            else break;
        }
        --last;
        for (;;) {
            if (first == last) return (uint32_t)(first - begin);
            else if (!seq_contains(refs[last->id], c)) --last;
            else break;
        }
        search_item t = *first;
        *first = *last;
        *last = t;
        ++first;
    }
}

int so_search(const so_index *idx, const so_cseq *const *refs, uint32_t n_refs, const so_cseq *c,
              const so_search_opts *o, uint32_t *out_ids, float *out_scores, uint32_t cap, so_log *log) {
    if (c->n < 20) {
        so_logf(log, "search:sequence too short (<20 bases);");
        return -1;
    }
    search_item *vc = NULL;
    uint32_t nvc = 0;
    uint32_t n_out = 0;
    if (o->search_all) { /* :271-296 */
        vc = (search_item *)malloc(sizeof(search_item) * (n_refs ? n_refs : 1));
        for (uint32_t r = 0; r < n_refs; r++) {
            vc[r].score = so_compare(c, refs[r], o->iupac, o->dist, o->cover, o->filter_lc);
            vc[r].id = r;
            vc[r].name = refs[r]->name;
        }
        /* :279-290, literally -- including what libstdc++'s partial_sort and partition leave
         * behind: when containing sequences are removed from the top, the second round sorts only
         * the slots that are left, and the range then handed to partition() extends into elements
         * partial_sort never ordered.  (A reference built with GCC behaves exactly like this.) */
        search_item *top = vc;
        uint32_t it = 0, end = n_refs;
        uint32_t middle = o->max_result < n_refs ? o->max_result : n_refs;
        do {
            gnu_partial_sort(vc + it, vc + middle, vc + end);
            if (o->ignore_super) {
                middle = it + o->max_result;
                if (middle > end) middle = end; /* (the reference would run past the end here) */
                it += gnu_partition_contains(vc + it, vc + middle, refs, c);
            }
        } while (middle != end && it + o->max_result > middle);
        top = vc + it;
        const uint32_t ntop = middle - it;
        for (uint32_t r = 0; r < ntop && top[r].score > o->min_sim; r++) {
            if (n_out < cap) {
                out_ids[n_out] = top[r].id;
                out_scores[n_out] = top[r].score;
            }
            n_out++;
        }
    } else { /* :297-331 */
        uint32_t kc = o->kmer_candidates < n_refs ? o->kmer_candidates : n_refs;
        uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (kc ? kc : 1));
        float *sc = (float *)malloc(sizeof(float) * (kc ? kc : 1));
        kc = so_index_find(idx, c, kc, ids, sc);
        vc = (search_item *)malloc(sizeof(search_item) * (kc ? kc : 1));
        for (uint32_t i = 0; i < kc; i++) {
            if (o->ignore_super && !seq_contains(refs[ids[i]], c)) continue; /* sic: partition + erase keeps the
                                                                                 containing ones (:307-310) */
            vc[nvc].id = ids[i];
            vc[nvc].name = refs[ids[i]]->name;
            vc[nvc].score = so_compare(c, refs[ids[i]], o->iupac, o->dist, o->cover, o->filter_lc);
            nvc++;
        }
        qsort(vc, nvc, sizeof(search_item), item_cmp_desc); /* partial_sort of the first max_result: same prefix */
        uint32_t middle = o->max_result < nvc ? o->max_result : nvc;
        for (uint32_t r = 0; r < middle && vc[r].score > o->min_sim; r++) {
            if (n_out < cap) {
                out_ids[n_out] = vc[r].id;
                out_scores[n_out] = vc[r].score;
            }
            n_out++;
        }
        free(ids);
        free(sc);
    }
    free(vc);
    return (int)n_out;
}

void so_search_nearest(const char *const *acc, const char *const *version, const char *const *start,
                       const char *const *stop, const uint32_t *ids, const float *scores, uint32_t n, so_log *out) {
    for (uint32_t i = 0; i < n; i++) /* fmt "{}.{}.{}.{}~{:.3f} " :357-363 */
        so_logf(out, "%s.%s.%s.%s~%.3f ", acc[ids[i]], version[ids[i]], start[ids[i]], stop[ids[i]],
                (double)scores[i]);
}

/* :339-352 (split) and :374-409 (vote) */
void so_search_lca(const char *const *tax, uint32_t n_results, float quorum, so_log *out) {
    /* group_names: one vector of names per result whose path is not "Unclassified;" */
    typedef struct {
        char **v;
        int n;
    } names;
    names *g = (names *)calloc(n_results ? n_results : 1, sizeof(names));
    int ng = 0;
    for (uint32_t i = 0; i < n_results; i++) {
        const char *p = tax[i];
        if (strcmp(p, "Unclassified;") == 0) continue;
        /* boost::split on ';' (no token compression): k delimiters give k+1 tokens */
        int ntok = 1;
        for (const char *q = p; *q; q++) ntok += (*q == ';');
        names nm;
        nm.v = (char **)malloc(sizeof(char *) * ntok);
        nm.n = 0;
        const char *tok = p;
        for (const char *q = p;; q++) {
            if (*q == ';' || *q == 0) {
                size_t len = (size_t)(q - tok);
                char *t = (char *)malloc(len + 1);
                memcpy(t, tok, len);
                t[len] = 0;
                nm.v[nm.n++] = t;
                tok = q + 1;
                if (*q == 0) break;
            }
        }
        if (nm.n > 0 && (nm.v[nm.n - 1][0] == 0 || strcmp(nm.v[nm.n - 1], " ") == 0)) free(nm.v[--nm.n]);
        /* reverse(vs): back() becomes the root; keep the order and pop from the FRONT instead */
        g[ng++] = nm;
    }
    int *front = (int *)calloc(ng ? ng : 1, sizeof(int)); /* index of the current "back()" of each vector */
    uint8_t *alive = (uint8_t *)malloc(ng ? ng : 1);
    memset(alive, 1, ng ? ng : 1);
    int n_alive = ng;
    so_log res;
    so_log_init(&res);
    int outliers = (int)((double)((float)n_results * (1 - quorum)) + .5);
#define EMPTY(i) (front[i] >= g[i].n)
    while (outliers >= 0 && n_alive > 0) {
        int first = 0;
        while (!alive[first]) first++;
        if (EMPTY(first)) {
            alive[first] = 0;
            n_alive--;
            outliers--;
            continue;
        }
        const char *name = g[first].v[front[first]];
        int it = first + 1;
        for (; it < ng; it++) {
            if (!alive[it]) continue;
            if (EMPTY(it) || strcmp(g[it].v[front[it]], name) != 0) break;
        }
        if (it < ng) {
            alive[it] = 0;
            n_alive--;
            outliers--;
            continue;
        }
        for (int i = 0; i < ng; i++)
            if (alive[i]) front[i]++;
        so_logf(&res, "%s;", name);
    }
#undef EMPTY
    const char *r = so_log_str(&res);
    size_t rl = strlen(r);
    if (rl > 1 && r[rl - 2] == ';' && r[rl - 1] == ';') r = r + rl - 1; /* res.substr(res.size()-1) */
    if (r[0] == 0 || strcmp(r, ";") == 0) r = "Unclassified;";
    so_logf(out, "%s", r);
    so_log_free(&res);
    for (int i = 0; i < ng; i++) {
        for (int x = 0; x < g[i].n; x++) free(g[i].v[x]);
        free(g[i].v);
    }
    free(g);
    free(front);
    free(alive);
}

