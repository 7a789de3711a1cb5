// HOST-PROFILING STUB -- not part of the product, never shipped in sina_amd/, never loaded by tests or bench.py.
//
// A stand-in for libsina_hip.so that answers the C-ABI calls of the host stages (host/stages.cpp) with
// plausible, deterministic, WRONG results in no time, so that the CPU cost of the host side of the boundary
// (trays, famfinder's cascade, the aligner's glue, the sink) can be measured per query in a container
// without a GPU (tools/hoststub/host_perf.py).  It computes nothing: k-mer "results" are hash-picked
// reference ids with descending scores, "alignments" put query base i in column 3 i.  Anything that checks
// results against the oracle must use the real library.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "sina_hip.h"

struct sina_hip_ctx {
    sina_hip_ctx *root;
    uint32_t n_refs = 0, width = 0;
    std::vector<uint32_t> staged;
};
static thread_local std::string g_err;

extern "C" {
int sina_hip_abi_version(void) { return SINA_HIP_ABI_VERSION; }
const char *sina_hip_last_error(void) { return g_err.c_str(); }
int sina_hip_init(int, sina_hip_ctx **ctx) {
    *ctx = new sina_hip_ctx();
    (*ctx)->root = *ctx;
    return 0;
}
int sina_hip_fork(sina_hip_ctx *parent, sina_hip_ctx **ctx) {
    *ctx = new sina_hip_ctx();
    (*ctx)->root = parent->root;
    return 0;
}
int sina_hip_prewarm(sina_hip_ctx *, int) { return 0; }
void sina_hip_destroy(sina_hip_ctx *c) { delete c; }
int sina_hip_sync(sina_hip_ctx *) { return 0; }
int sina_hip_upload_refs(sina_hip_ctx *c, const uint32_t *, const uint64_t *, uint32_t n_refs, uint32_t width) {
    c->root->n_refs = n_refs;
    c->root->width = width;
    return 0;
}
int sina_hip_build_index(sina_hip_ctx *, unsigned, int) { return 0; }
int sina_hip_download_index(sina_hip_ctx *, uint32_t *, uint32_t *) { g_err = "stub"; return 1; }
int sina_hip_upload_index(sina_hip_ctx *, unsigned, int, const uint32_t *, const uint32_t *, uint64_t) { return 0; }
int sina_hip_store_view_get(sina_hip_ctx *, sina_hip_store_view *) { g_err = "stub"; return 1; }
int sina_hip_store_alloc_like(sina_hip_ctx *, sina_hip_store_view *) { g_err = "stub"; return 1; }
int sina_hip_kmer_topk(sina_hip_ctx *c, const uint8_t *qmask, const uint64_t *qoff, uint32_t nq, uint32_t max,
                       uint32_t *out_ids, float *out_scores, uint32_t *out_n) {
    const uint32_t n = c->root->n_refs;
    if (max > n) max = n;
    for (uint32_t q = 0; q < nq; q++) {
        uint64_t h = 0x9E3779B97F4A7C15ull * (qoff[q + 1] - qoff[q] + 1);
        for (uint64_t x = qoff[q]; x < qoff[q] + 16 && x < qoff[q + 1]; x++) h = (h ^ qmask[x]) * 0x100000001b3ull;
        const uint32_t base = (uint32_t)(h % n);
        for (uint32_t x = 0; x < max; x++) {
            out_ids[(size_t)q * max + x] = (base + x * 97u) % n;  // (distinct while 97 x < n)
            // (below the query's own k-mer count -- about a quarter of its bases in "fast" mode -- as the scores of a real
            // search nearly always are: a member that carries EVERY k-mer of the query sends the aligner into its
            // exact-relative test, a string search per member)
            const int top = (int)((qoff[q + 1] - qoff[q]) / 6);
            out_scores[(size_t)q * max + x] = (float)std::max(1, top - (int)(x % 200));
        }
        out_n[q] = max;
    }
    return 0;
}
int sina_hip_kmer_scores(sina_hip_ctx *, const uint8_t *, uint32_t, int16_t *) { g_err = "stub"; return 1; }
int sina_hip_compare(sina_hip_ctx *, const uint32_t *, const uint64_t *, uint32_t, const uint32_t *, const uint64_t *,
                     int, int, sina_hip_match_counts *) { g_err = "stub"; return 1; }
void sina_hip_align_params_default(sina_hip_align_params *p) {
    memset(p, 0, sizeof(*p));
    p->match_score = 2;
    p->mismatch_score = -1;
    p->gap_penalty = 5;
    p->gap_ext_penalty = 2;
    p->fs_weight = 1;
}
const uint32_t *sina_hip_staged_out_pos(sina_hip_ctx *c) { return c->staged.data(); }
int sina_hip_align_families(sina_hip_ctx *c, const uint32_t *, const uint64_t *, uint32_t nq, const uint8_t *qmask,
                            const uint64_t *qoff, const sina_hip_align_params *, sina_hip_align_out *out, uint32_t *) {
    const uint64_t total = qoff[nq] - qoff[0];
    if (c->staged.size() < total) c->staged.resize(total + total / 4);
    for (uint32_t q = 0; q < nq; q++) {
        const uint32_t L = (uint32_t)(qoff[q + 1] - qoff[q]);
        uint32_t *pos = c->staged.data() + (qoff[q] - qoff[0]);
        const uint8_t *m = qmask + qoff[q];
        for (uint32_t i = 0; i < L; i++) pos[i] = (3u * i) | ((uint32_t)(m[i] & 0xf) << 24);
        sina_hip_align_out &o = out[q];
        memset(&o, 0, sizeof o);
        o.end_s = L - 1;
        o.raw = -1800.f;
        o.sum_weight = -2000.f;
        o.aligned_bases = (int32_t)L;
        o.n_out = L;
        o.assembled = 1;
    }
    return 0;
}
int sina_hip_align_graphs(sina_hip_ctx *, const sina_hip_graph_batch *, const uint8_t *, const uint64_t *,
                          const sina_hip_align_params *, sina_hip_align_out *, uint32_t *) { g_err = "stub"; return 1; }
int sina_hip_get_stats(sina_hip_ctx *, sina_hip_stats *s) { memset(s, 0, sizeof *s); return 0; }
}
