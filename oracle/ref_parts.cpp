/*
 * ref_parts.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * C-callable harness over the parts of the reference that compile from their
 * own sources with nothing but the standard library.  It #includes them from
 * where they lie (-I/root/reference/src) and links /root/reference/src/
 * aligned_base.cpp; no reference source is copied into this repository and the
 * output goes to oracle/_ref/ only (git-ignored).
 *
 *   real code used:  kmer.h (generator / prefix_filter / unique_filter /
 *                    iterable), idset.h (vlimap), aligned_base.{h,cpp}
 *                    (IUPAC tables, aligned_compact), graph.h (dag<T>:
 *                    insert/link/sort/reduce_edges, sentinel source/sink lists),
 *                    scoring_schemes.h (scoring_scheme_simple / _weighted)
 *   harness code:    the 20-line node type (mirrors src/mseq.h:41-65), the
 *                    column sweep that feeds dag<T> (follows src/mseq.cpp:47-118)
 *                    and the cell transition loop that calls the real scoring
 *                    scheme on the real dag (follows src/mesh.h:307-374,455-502).
 *
 * NOT buildable here (need Boost / TBB / libARBDB headers the image lacks):
 * cseq.cpp, mseq.cpp, mesh.h, kmer_search.cpp, famfinder.cpp, align.cpp.
 */
#include <cstdint>
#include <cstring>
#include <sstream>
#include <limits>
#include <string>
#include <unordered_set>
#include <vector>

#include "aligned_base.h"
#include "graph.h"
#include "idset.h"
#include "kmer.h"
#include "scoring_schemes.h"

using namespace sina;

extern "C" {

/* ---- IUPAC tables (aligned_base.cpp) */
int ref_char_to_mask(int c) {
    try {
        base_iupac b((unsigned char)c);
        aligned_base ab(0, (unsigned char)c);
        uint32_t raw;
        memcpy(&raw, &ab, 4);
        return (int)(raw >> 24);
    } catch (base_iupac::bad_character_exception &) {
        return -1;
    }
}
int ref_mask_to_rna(int m) { return base_iupac::bmask_to_iupac_rna_char[m & 31]; }
int ref_mask_to_dna(int m) { return base_iupac::bmask_to_iupac_dna_char[m & 31]; }
uint32_t ref_pack(uint32_t pos, int c) { /* aligned_compact layout */
    aligned_base ab(pos, (unsigned char)c);
    uint32_t raw;
    memcpy(&raw, &ab, 4);
    return raw;
}

/* ---- kmer.h: per-push trace of the generator stack */
void ref_kmer_trace(const char *seq, unsigned k, unsigned p_len, unsigned p_val, int unique,
                    uint8_t *good_out, uint32_t *val_out) {
    std::unordered_set<unsigned int> seen;
    size_t n = strlen(seq);
    if (!unique && !p_len) {
        kmer_generator g(k);
        for (size_t i = 0; i < n; i++) { g.push(seq[i]); good_out[i] = g.good(); val_out[i] = g.val(); }
    } else if (!unique) {
        prefix_kmer_generator g(k, p_len, p_val);
        for (size_t i = 0; i < n; i++) { g.push(seq[i]); good_out[i] = g.good(); val_out[i] = g.val(); }
    } else if (!p_len) {
        unique_kmer_generator g(seen, k);
        for (size_t i = 0; i < n; i++) { g.push(seq[i]); good_out[i] = g.good(); val_out[i] = g.val(); }
    } else {
        unique_prefix_kmer_generator g(seen, k, p_len, p_val);
        for (size_t i = 0; i < n; i++) { g.push(seq[i]); good_out[i] = g.good(); val_out[i] = g.val(); }
    }
}

static std::vector<aligned_base> unpack(const uint32_t *ab, uint32_t n) {
    std::vector<aligned_base> v(n);
    if (n) memcpy((void *)v.data(), ab, 4 * (size_t)n);
    return v;
}

/* ---- kmer.h: the range-for iterables exactly as kmer_search.cpp uses them */
uint32_t ref_kmers(const uint32_t *ab, uint32_t n, unsigned k, unsigned p_len, unsigned p_val,
                   int unique, uint32_t *out) {
    const std::vector<aligned_base> bases = unpack(ab, n);
    std::unordered_set<unsigned int> seen;
    uint32_t cnt = 0;
    if (!unique && !p_len) {
        for (unsigned int km : all_kmers(bases, k, 1)) out[cnt++] = km;
    } else if (!unique) {
        for (unsigned int km : prefix_kmers(bases, k, p_len, p_val)) out[cnt++] = km;
    } else if (!p_len) {
        for (const auto &km : unique_kmers(bases, seen, k)) out[cnt++] = km;
    } else {
        for (unsigned int km : unique_prefix_kmers(bases, seen, (int)k, p_len, p_val)) out[cnt++] = km;
    }
    return cnt;
}

/* ---- idset.h: vlimap */
void *ref_vlimap_new(uint32_t maxsize) { return new vlimap(maxsize); }
void ref_vlimap_free(void *v) { delete (vlimap *)v; }
void ref_vlimap_push_back(void *v, uint32_t n) { ((vlimap *)v)->push_back(n); }
int ref_vlimap_increment(void *v, int16_t *t, uint32_t n) {
    idset::inc_t tmp(t, t + n);
    int r = ((vlimap *)v)->increment(tmp);
    memcpy(t, tmp.data(), sizeof(int16_t) * n);
    return r;
}
void ref_vlimap_append(void *v, void *o) { ((vlimap *)v)->append(*(vlimap *)o); }
void ref_vlimap_invert(void *v) { ((vlimap *)v)->invert(); }
uint32_t ref_vlimap_size(void *v) { return (uint32_t)((vlimap *)v)->size(); }
/* serialised form = 16-byte header {inc,last,bytesize,size} + bytes (idset.h:386-398) */
uint32_t ref_vlimap_write(void *v, uint8_t *out, uint32_t cap) {
    std::ostringstream o;
    ((vlimap *)v)->write(o);
    std::string s = o.str();
    if (s.size() <= cap) memcpy(out, s.data(), s.size());
    return (uint32_t)s.size();
}

} /* extern "C" */

/* ---- graph.h dag<T> fed by the mseq column sweep */
namespace {
class rnode : public aligned_base {
public:
    rnode(const aligned_base &b) : aligned_base(b), weight(1.f) {}
    rnode(int i, char c) : aligned_base(i, c), weight(1.f) {}
    float getWeight() const { return weight; }
    bool operator<(const rnode &rhs) const { return aligned_base::operator<(rhs); }
    float weight;
};
typedef dag<rnode> rdag;

struct rgraph {
    rdag g;
    unsigned width;
};

rgraph *build(const uint32_t *const *fam, const uint32_t *fam_n, uint32_t F, uint32_t width, float weight) {
    rgraph *R = new rgraph();
    R->width = width;
    rdag &G = R->g;
    std::vector<std::vector<aligned_base>> seqs(F);
    for (uint32_t j = 0; j < F; j++) seqs[j] = unpack(fam[j], fam_n[j]);
    std::vector<size_t> cit(F, 0);
    std::vector<rdag::iterator> last(F);
    const size_t nodes_size = 255;
    std::vector<rdag::iterator> nodes(nodes_size);
    aligned_base::idx_type min_next = 0;
    for (unsigned int i = 0; i < width; i++) {
        if (min_next > i) continue;
        min_next = std::numeric_limits<int>::max();
        nodes.assign(nodes_size, rdag::iterator());
        for (unsigned int j = 0; j < F; j++) {
            if (cit[j] != seqs[j].size() && seqs[j][cit[j]].getPosition() == i) {
                rdag::iterator newnode;
                unsigned char base = seqs[j][cit[j]].getBase();
                if (nodes[base].isNull()) {
                    nodes[base] = newnode = G.insert(rnode(seqs[j][cit[j]]));
                } else {
                    newnode = nodes[base];
                    newnode->weight += 1.f;
                }
                if (!last[j].isNull()) G.link(last[j], newnode);
                last[j] = newnode;
                ++cit[j];
            }
            if (cit[j] != seqs[j].size()) min_next = std::min(min_next, seqs[j][cit[j]].getPosition());
        }
        for (auto &node : nodes) {
            if (!node.isNull()) node->weight = 1.0 / (weight + 1) + weight * (node->weight / F);
        }
    }
    G.sort();
    G.reduce_edges();
    return R;
}
} // namespace

extern "C" {

void *ref_dag_build(const uint32_t *const *fam, const uint32_t *fam_n, uint32_t F, uint32_t width,
                    float weight) {
    return build(fam, fam_n, F, width, weight);
}
void ref_dag_free(void *r) { delete (rgraph *)r; }
uint32_t ref_dag_size(void *r) { return ((rgraph *)r)->g.size(); }

/* dump in list order: id, pos, mask, weight; returns edge count. pred lists are
 * written in the order the real list holds them after reduce_edges(). */
uint32_t ref_dag_dump(void *r, uint32_t *ids, uint32_t *pos, uint8_t *mask, float *weight,
                      uint32_t *pred_off, uint32_t *pred, uint32_t *n_src, uint32_t *src,
                      uint32_t *n_snk, uint32_t *snk) {
    rdag &G = ((rgraph *)r)->g;
    uint32_t i = 0, e = 0;
    for (rdag::iterator it = G.begin(); it != G.end(); ++it, ++i) {
        ids[i] = get_node_id(G, it);
        pos[i] = it->getPosition();
        aligned_base ab = *it;
        uint32_t raw;
        memcpy(&raw, &ab, 4);
        mask[i] = (uint8_t)(raw >> 24);
        weight[i] = it->getWeight();
        pred_off[i] = e;
        for (rdag::pn_iterator p = prev_begin(G, it); p != prev_end(G, it); ++p) pred[e++] = get_node_id(G, p);
    }
    pred_off[i] = e;
    *n_src = 0;
    for (rdag::pn_iterator p = G.pn_first_begin(); p != G.pn_first_end(); ++p) src[(*n_src)++] = get_node_id(G, p);
    *n_snk = 0;
    for (rdag::pn_iterator p = G.pn_last_begin(); p != G.pn_last_end(); ++p) snk[(*n_snk)++] = get_node_id(G, p);
    return e;
}

/* ---- scoring_schemes.h: single-op probes of the real arithmetic.
 * op: 0 insertion 1 insertion_ext 2 deletion 3 deletion_ext 4 match */
float ref_score_op(int op, float prev, uint32_t mpos, int mchar, float mweight, int schar, int offset,
                   float ms, float mms, float gp, float gpe, const float *weights, uint32_t nw) {
    rnode b1((int)mpos, (char)mchar);
    b1.weight = mweight;
    aligned_base b2(0, (unsigned char)schar);
    if (!weights) {
        scoring_scheme_simple s(ms, mms, gp, gpe);
        switch (op) {
        case 0: return s.insertion(prev, b1, b2);
        case 1: return s.insertion_ext(prev, b1, b2, offset);
        case 2: return s.deletion(prev, b1, b2);
        case 3: return s.deletion_ext(prev, b1, b2, offset);
        default: return s.match(prev, b1, b2);
        }
    }
    std::vector<float> w(weights, weights + nw);
    scoring_scheme_weighted s(ms, mms, gp, gpe, w);
    switch (op) {
    case 0: return s.insertion(prev, b1, b2);
    case 1: return s.insertion_ext(prev, b1, b2, offset);
    case 2: return s.deletion(prev, b1, b2);
    case 3: return s.deletion_ext(prev, b1, b2, offset);
    default: return s.match(prev, b1, b2);
    }
}

/* ---- the cell loop on the REAL dag with the REAL scoring_scheme_simple.
 * cells layout = 7 x u32/f32 + gaps_max, same as oracle so_cell. */
struct rcell {
    uint32_t value_midx, value_sidx, gapm_idx, gaps_idx;
    float value, gapm_val, gaps_val;
    uint32_t gaps_max;
};

} /* extern "C" (the cell loop below is a template over the REAL scoring scheme classes) */

/* The cell loop of compute_node_simple::calc over transition_simple / transition_aspace_aware
 * (src/mesh.h:307-374,403-437,455-502) on the REAL dag, calling the REAL scoring scheme for every
 * operation.  The loop itself is harness code (mesh.h needs Boost / TBB headers the image lacks);
 * what it pins is the arithmetic, the order of candidates and the DAG it runs on.
 * forbid != 0: transition_aspace_aware (--insertion=forbid): an insertion run may be at most
 * max_insert = min over successors of (pos(succ)) - pos(m) - 1 columns long (:480-489). */
template <typename SCHEME>
static void mesh_cells(rdag &G, const std::vector<aligned_base> &q, SCHEME &s, int forbid, rcell *cells) {
    const uint32_t L = (uint32_t)q.size();
    for (rdag::iterator m = G.begin(); m != G.end(); ++m) {
        const uint32_t midx = get_node_id(G, m);
        uint32_t max_insert = 0;
        if (forbid) {
            unsigned int min_mpos = 1000000;
            for (rdag::pn_iterator n = m.next_begin(); n != m.next_end(); ++n)
                min_mpos = std::min<unsigned int>(min_mpos, n->getPosition());
            max_insert = (uint32_t)(int)(min_mpos - m->getPosition() - 1);
        }
        for (uint32_t sidx = 0; sidx < L; sidx++) {
            rcell d;
            const bool edge = (prev_begin(G, m) == prev_end(G, m)) || sidx == 0;
            d.value = d.gapm_val = d.gaps_val = edge ? 1 : 1000000;
            d.value_midx = d.value_sidx = d.gapm_idx = d.gaps_idx = 0;
            d.gaps_max = 0;
            for (rdag::pn_iterator p = prev_begin(G, m); p != prev_end(G, m); ++p) {
                uint32_t mi = get_node_id(G, p);
                const rcell &src = cells[(size_t)mi * L + sidx];
                float value = s.deletion(src.value, *m, q[sidx]);
                float gap_val = s.deletion_ext(src.gapm_val, *m, q[sidx], 0);
                if (value < gap_val) {
                    d.gapm_val = value;
                    d.gapm_idx = mi;
                } else {
                    d.gapm_val = gap_val;
                    d.gapm_idx = src.gapm_idx;
                    value = gap_val;
                    mi = src.gapm_idx;
                }
                if (value < d.value) {
                    d.value = value;
                    d.value_midx = mi;
                    d.value_sidx = sidx;
                }
            }
            if (sidx > 0) {
                const uint32_t si = sidx - 1;
                const rcell &src = cells[(size_t)midx * L + si];
                bool inserted = true;
                if (!forbid) {
                    if (src.gaps_val != src.value) {
                        d.gaps_val = s.insertion(src.value, *m, q[sidx]);
                        d.gaps_idx = si;
                    } else {
                        d.gaps_val = s.insertion_ext(src.gaps_val, *m, q[sidx], si - src.gaps_idx);
                        d.gaps_idx = src.gaps_idx;
                    }
                } else if (max_insert < 1) {
                    inserted = false;
                } else if (src.gaps_val != src.value) {
                    d.gaps_val = s.insertion(src.value, *m, q[sidx]);
                    d.gaps_idx = si;
                    d.gaps_max = max_insert - 1;
                } else if (src.gaps_max > 0) {
                    d.gaps_val = s.insertion_ext(src.gaps_val, *m, q[sidx], si - src.gaps_idx);
                    d.gaps_idx = src.gaps_idx;
                    d.gaps_max = src.gaps_max - 1;
                } else {
                    inserted = false;
                }
                if (inserted && d.gaps_val <= d.value) {
                    d.value = d.gaps_val;
                    d.value_sidx = d.gaps_idx;
                    d.value_midx = midx;
                }
                for (rdag::pn_iterator p = prev_begin(G, m); p != prev_end(G, m); ++p) {
                    uint32_t mi = get_node_id(G, p);
                    float value = s.match(cells[(size_t)mi * L + si].value, *m, q[sidx]);
                    if (value < d.value) {
                        d.value = value;
                        d.value_midx = mi;
                        d.value_sidx = si;
                    }
                }
            }
            cells[(size_t)midx * L + sidx] = d;
        }
    }
}

extern "C" {

/* weights == NULL: scoring_scheme_simple, else scoring_scheme_weighted over those column weights */
void ref_mesh_compute(void *r, const uint32_t *qab, uint32_t L, float ms, float mms, float gp, float gpe,
                      const float *weights, uint32_t nw, int forbid, rcell *cells) {
    rdag &G = ((rgraph *)r)->g;
    std::vector<aligned_base> q = unpack(qab, L);
    if (!weights) {
        scoring_scheme_simple s(ms, mms, gp, gpe);
        mesh_cells(G, q, s, forbid, cells);
    } else {
        std::vector<float> w(weights, weights + nw);
        scoring_scheme_weighted s(ms, mms, gp, gpe, w);
        mesh_cells(G, q, s, forbid, cells);
    }
}

void ref_mesh_compute_simple(void *r, const uint32_t *qab, uint32_t L, float ms, float mms, float gp,
                             float gpe, rcell *cells) {
    ref_mesh_compute(r, qab, L, ms, mms, gp, gpe, NULL, 0, 0, cells);
}

} /* extern "C" */
