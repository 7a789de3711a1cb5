// Mesh DP fill + backtrack walk for gfx950 (CDNA4).
//
// What it computes: the recurrence of compute_node_simple::calc over
// transition_simple / transition_aspace_aware with scoring_scheme_simple /
// scoring_scheme_weighted (reference src/mesh.h:263-528, src/scoring_schemes.h:
// 102-241; spec in SURVEY.md section 8a) and the cell walk of backtrack()
// (src/mesh.h:535-721).  IEEE float32 min-plus; this file must be compiled with
// -ffp-contract=off so that `a + b * c` stays mul-then-add like the reference's
// SSE build.
//
// How it maps to the hardware (no MFMA: this is a compare/select recurrence):
//   * one workgroup per query; thread j owns B consecutive query columns, the
//     whole workgroup sweeps the DAG rows in topological (id) order, so the
//     predecessor list of a row is wave-uniform (scalar loads, no divergence);
//   * the last W rows of {value, gapm_val, gapm_idx} live in an LDS ring (10 B
//     per column); the rare predecessors further back than W rows are read
//     from a per-query spill area in HBM that the producing row also wrote;
//   * the only dependency inside a row is the insertion chain along the query
//     (cell (m,s) needs the final (m,s-1)).  Each thread runs the chain over
//     its own B cells serially (exact reference order) and assumes no chain
//     enters from the left; after one barrier it re-runs the chain with the
//     left neighbour's real exit state and a workgroup vote repeats that only
//     while some exit state still changed (a chain crossing a whole thread
//     block, i.e. >= B consecutive insertions);
//   * the only per-cell HBM traffic is the write-once trace-back cell
//     (value_midx:16 | value_sidx:16), coalesced row-major.
#include "common.h"

namespace sina_hip {

namespace {

constexpr uint32_t kNoSpill = 0xFFFFFFFFu;

struct ChainState {  // canonical exit state of cell (m, s): what cell (m, s+1) can observe
    float v;         // final value
    uint32_t e;      // gaps_val == value  (the "extend" condition, mesh.h:340)
    uint32_t gsi;    // gaps_idx (only meaningful when e)
    uint32_t gmax;   // gaps_max (FORBID only, only meaningful when e)
};

__device__ __forceinline__ bool same_state(const ChainState &a, const ChainState &b) {
    return a.v == b.v && a.e == b.e && (!a.e || (a.gsi == b.gsi && a.gmax == b.gmax));
}

template <int T, int B, bool WEIGHTED, bool FORBID>
__global__ void __launch_bounds__(T) mesh_dp_kernel(DpArgs a) {
    constexpr int Lp = T * B;
    constexpr int NW = (T + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int j = threadIdx.x;
    const QDesc d = a.qd[blockIdx.x];
    const uint32_t N = d.N, L = d.L;
    const uint32_t s0 = (uint32_t)j * B;

    // ---- LDS carve (all dynamic; base 16-B aligned, every offset a multiple of 16)
    float *xs_v = reinterpret_cast<float *>(smem);
    uint32_t *xs_e = reinterpret_cast<uint32_t *>(smem + 4 * T);
    uint32_t *xs_gmax = reinterpret_cast<uint32_t *>(smem + 8 * T);
    float *red_v = reinterpret_cast<float *>(smem + 12 * T);
    uint32_t *red_s = reinterpret_cast<uint32_t *>(smem + 12 * T + 64);
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + 12 * T + 128);
    unsigned char *ring = smem + 12 * T + 256;
    constexpr size_t kValBytes = (size_t)(Lp + 4) * 4;
    constexpr size_t kGmBytes = (size_t)Lp * 4;
    constexpr size_t kSlotBytes = kValBytes + kGmBytes + (size_t)Lp * 2;
    const int W = a.W;

    const uint32_t *pred_off = a.pred_off + d.poff_off;
    const uint32_t *pred = a.pred + d.edge_off;
    const uint32_t *node_pos = a.node_pos + d.node_off;
    const uint8_t *node_mask = a.node_mask + d.node_off;
    const float *node_weight = a.node_weight + d.node_off;
    const uint32_t *spill_idx = a.spill_idx + d.node_off;
    const uint8_t *node_flags = a.node_flags + d.node_off;
    uint32_t *tb = a.tb + d.tb_off;
    float *spill = a.spill + d.spill_off * (size_t)(3 * Lp);

    // query masks of my columns (0 beyond L: never matches, never stored)
    uint32_t qm[B];
#pragma unroll
    for (int k = 0; k < B; k++) {
        uint32_t s = s0 + k;
        qm[k] = (s < L) ? (uint32_t)(a.qmask[d.q_off + s] & 0xf) : 0u;
    }

    // end-cell search state (mesh.h:567-592)
    const bool own_last = (L - 1) / B == (uint32_t)j;
    const int k_last = (int)((L - 1) % B);
    float lc_min = 0.f, lc_snk0 = 0.f;  // step 1: min over all rows at column L-1
    uint32_t lc_arg = 0;
    bool lc_any = false;
    float sk_min = 0.f;  // step 2: min over sink rows x all columns (first in scan order)
    uint32_t sk_m = 0, sk_s = 0, snk0 = 0;
    bool sk_any = false;

    for (uint32_t m = 0; m < N; ++m) {
        const uint32_t pb = pred_off[m], pe = pred_off[m + 1];
        const float wgt = node_weight[m];
        const uint32_t mmask = node_mask[m] & 0xfu;
        const bool edge_row = (pb == pe);
        uint32_t mpos = 0;
        float cM, cX, gd_open, gd_ext, gi_open;
        if constexpr (WEIGHTED) {
            mpos = node_pos[m];
            const uint32_t nw1 = a.n_weights - 1;
            const float wp = a.weights[mpos < nw1 ? mpos : nw1];
            const float wp1 = a.weights[mpos + 1 < nw1 ? mpos + 1 : nw1];
            cM = a.ms * wp * wgt;  // (c * weights[pos]) * weight, scoring_schemes.h:232
            cX = a.mms * wp * wgt;
            gd_open = a.gp * wp;   // :211
            gd_ext = a.gpe * wp;   // :222
            gi_open = a.gp * wp1;  // :187
        } else {
            cM = a.ms * wgt;       // scoring_schemes.h:154
            cX = a.mms * wgt;
            gd_open = a.gp;
            gd_ext = a.gpe;
            gi_open = a.gp;
        }
        uint32_t smax = 0;
        if constexpr (FORBID) {
            if (!WEIGHTED) mpos = node_pos[m];
            // int max_insert = min_mpos - pos - 1, passed as unsigned idx_type (mesh.h:480-489)
            smax = (uint32_t)(int)(a.succ_minpos[d.node_off + m] - mpos - 1);
        }
        const float init_v = edge_row ? 1.0f : 1000000.0f;

        // ---- phase 1: deletion + match candidates from every predecessor row
        float dv[B], gm[B], mt[B];
        uint32_t dvm[B], gmi[B], mtp[B];
        bool ddel[B];  // best-so-far is a deletion (value_sidx = s) rather than the init cell
#pragma unroll
        for (int k = 0; k < B; k++) {
            const float iv = (s0 + k == 0) ? 1.0f : init_v;
            dv[k] = iv;
            gm[k] = iv;
            mt[k] = __builtin_inff();
            dvm[k] = 0;
            gmi[k] = 0;
            mtp[k] = 0;
            ddel[k] = false;
        }
        for (uint32_t e = pb; e < pe; ++e) {
            const uint32_t p = pred[e];
            float sv[B], sg[B], svl;
            uint32_t sgi[B];
            if (m - p <= (uint32_t)W) {  // LDS ring
                const unsigned char *slot = ring + (size_t)(p % (uint32_t)W) * kSlotBytes;
                const float *pv = reinterpret_cast<const float *>(slot) + 4;
                const float *pg = reinterpret_cast<const float *>(slot + kValBytes);
                const uint16_t *pi = reinterpret_cast<const uint16_t *>(slot + kValBytes + kGmBytes);
                svl = pv[(int)s0 - 1];
#pragma unroll
                for (int k = 0; k < B; k++) {
                    sv[k] = pv[s0 + k];
                    sg[k] = pg[s0 + k];
                    sgi[k] = pi[s0 + k];
                }
            } else {  // spilled row in HBM
                const float *row = spill + (size_t)spill_idx[p] * (3 * Lp);
                svl = (s0 > 0) ? row[s0 - 1] : 0.f;
#pragma unroll
                for (int k = 0; k < B; k++) {
                    sv[k] = row[s0 + k];
                    sg[k] = row[Lp + s0 + k];
                    sgi[k] = reinterpret_cast<const uint32_t *>(row)[2 * Lp + s0 + k];
                }
            }
#pragma unroll
            for (int k = 0; k < B; k++) {
                // deletion (mesh.h:307-330): gapm_* is overwritten by every predecessor
                const float v = sv[k] + gd_open;
                const float g = sg[k] + gd_ext;
                const bool op = v < g;
                const float cand = op ? v : g;
                const uint32_t cm = op ? p : sgi[k];
                gm[k] = cand;
                gmi[k] = cm;
                if (cand < dv[k]) {
                    dv[k] = cand;
                    dvm[k] = cm;
                    ddel[k] = true;
                }
                // match from (p, s-1) (mesh.h:360-374); first predecessor with the minimum wins
                const float pvv = (k == 0) ? svl : sv[k - 1];
                const float mv = pvv + ((mmask & qm[k]) ? cM : cX);
                if ((s0 + k) > 0 && mv < mt[k]) {
                    mt[k] = mv;
                    mtp[k] = p;
                }
            }
        }

        // ---- phase 2: insertion chain along my B cells
        float fv[B];
        uint32_t fvm[B], fvs[B];
        ChainState ex;
        auto run_chain = [&](const ChainState &left) {
            ChainState c = left;
#pragma unroll
            for (int k = 0; k < B; k++) {
                const uint32_t s = s0 + k;
                float v = dv[k];
                uint32_t vm = dvm[k];
                uint32_t vs = ddel[k] ? s : 0u;
                float gs;
                uint32_t gsi = 0, gmax = 0;
                if (s > 0) {
                    bool ins = true;
                    if (!c.e) {  // opening gap (mesh.h:340-343 / :415-419)
                        if (FORBID && smax < 1) {
                            ins = false;
                        } else {
                            gs = c.v + gi_open;
                            gsi = s - 1;
                            if (FORBID) gmax = smax - 1;
                        }
                    } else {  // extending gap (:344-349 / :420-425); gaps_val == value here
                        if (FORBID && (smax < 1 || c.gmax == 0)) {
                            ins = false;
                        } else {
                            float gi_ext = a.gpe;
                            if constexpr (WEIGHTED) {
                                const uint32_t nw1 = a.n_weights - 1;
                                const uint32_t wi = mpos + 1 + ((s - 1) - c.gsi);
                                gi_ext = a.gpe * a.weights[wi < nw1 ? wi : nw1];
                            }
                            gs = c.v + gi_ext;
                            gsi = c.gsi;
                            if (FORBID) gmax = c.gmax - 1;
                        }
                    }
                    if (!ins) {  // cell keeps its initial gaps_val / gaps_idx / gaps_max
                        gs = init_v;
                        gsi = 0;
                        gmax = 0;
                    } else if (gs <= v) {  // mesh.h:351-357
                        v = gs;
                        vm = m;
                        vs = gsi;
                    }
                    if (mt[k] < v) {
                        v = mt[k];
                        vm = mtp[k];
                        vs = s - 1;
                    }
                } else {
                    gs = 1.0f;  // init_edge at s == 0, no insertion step
                }
                fv[k] = v;
                fvm[k] = vm;
                fvs[k] = vs;
                c.v = v;
                c.e = (gs == v) ? 1u : 0u;
                c.gsi = gsi;
                c.gmax = gmax;
            }
            ex = c;
        };

        // speculative pass: pretend the left neighbour's last cell did not end in a gap
        // and is so expensive that opening from it can never win.
        ChainState left;
        left.v = __builtin_inff();
        left.e = 0;
        left.gsi = 0;
        left.gmax = 0;
        run_chain(left);
        ChainState published = ex;
        xs_v[j] = ex.v;
        xs_e[j] = (ex.e << 31) | ex.gsi;
        if (FORBID) xs_gmax[j] = ex.gmax;
        __syncthreads();  // B1: all ring reads of this row done, exit states visible

        unsigned char *myslot = ring + (size_t)(m % (uint32_t)W) * kSlotBytes;
        float *wv = reinterpret_cast<float *>(myslot) + 4;
        float *wg = reinterpret_cast<float *>(myslot + kValBytes);
        uint16_t *wi = reinterpret_cast<uint16_t *>(myslot + kValBytes + kGmBytes);
#pragma unroll
        for (int k = 0; k < B; k++) {
            wg[s0 + k] = gm[k];
            wi[s0 + k] = (uint16_t)gmi[k];
        }
        for (;;) {
            if (j > 0) {
                left.v = xs_v[j - 1];
                const uint32_t pe_ = xs_e[j - 1];
                left.e = pe_ >> 31;
                left.gsi = pe_ & 0x7fffffffu;
                left.gmax = FORBID ? xs_gmax[j - 1] : 0u;
                run_chain(left);
            }
#pragma unroll
            for (int k = 0; k < B; k++) wv[s0 + k] = fv[k];
            const bool changed = !same_state(ex, published);
            // the vote is also the barrier that publishes this row's ring slot
            if (!__syncthreads_or(changed ? 1 : 0)) break;
            if (changed) {
                published = ex;
                xs_v[j] = ex.v;
                xs_e[j] = (ex.e << 31) | ex.gsi;
                if (FORBID) xs_gmax[j] = ex.gmax;
            }
            __syncthreads();
        }

        // ---- outputs of the row
        {
            uint32_t *trow = tb + (size_t)m * Lp + s0;
#pragma unroll
            for (int k = 0; k < B; k++) trow[k] = (fvm[k] << 16) | (fvs[k] & 0xffffu);
        }
        if (a.dbg_value != nullptr && blockIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < B; k++) a.dbg_value[(size_t)m * Lp + s0 + k] = fv[k];
        }
        const uint32_t sp = spill_idx[m];
        if (sp != kNoSpill) {
            float *row = spill + (size_t)sp * (3 * Lp);
#pragma unroll
            for (int k = 0; k < B; k++) {
                row[s0 + k] = fv[k];
                row[Lp + s0 + k] = gm[k];
                reinterpret_cast<uint32_t *>(row)[2 * Lp + s0 + k] = gmi[k];
            }
        }

        // step 1 of the end-cell search: rows at the last query column
        const bool is_sink = (node_flags[m] & 1u) != 0;
        if (own_last) {
            float v = fv[0];
#pragma unroll
            for (int k = 1; k < B; k++)
                if (k == k_last) v = fv[k];
            if (!lc_any || v < lc_min) {
                lc_min = v;
                lc_arg = m;
                lc_any = true;
            }
        }
        // step 2: sink rows, every column (uniform branch: is_sink is a row property)
        if (is_sink) {
            float bv = __builtin_inff();
            uint32_t bs = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < B; k++) {
                const uint32_t s = s0 + k;
                if (s < L && fv[k] < bv) {
                    bv = fv[k];
                    bs = s;
                }
            }
            // wave reduce: smaller value, then smaller column
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(bv, off);
                const uint32_t os = __shfl_xor(bs, off);
                if (ov < bv || (ov == bv && os < bs)) {
                    bv = ov;
                    bs = os;
                }
            }
            if ((j & 63) == 0) {
                red_v[j >> 6] = bv;
                red_s[j >> 6] = bs;
            }
            if (own_last) {  // value of this sink row at column L-1, for sinks[0]
                float v = fv[0];
#pragma unroll
                for (int k = 1; k < B; k++)
                    if (k == k_last) v = fv[k];
                red_v[NW] = v;
            }
            __syncthreads();
            float rv = red_v[0];
            uint32_t rs = red_s[0];
            for (int w = 1; w < NW; w++) {
                if (red_v[w] < rv || (red_v[w] == rv && red_s[w] < rs)) {
                    rv = red_v[w];
                    rs = red_s[w];
                }
            }
            if (!sk_any) {  // first sink == sinks[0] (ascending ids)
                snk0 = m;
                lc_snk0 = red_v[NW];
                sk_min = rv;
                sk_m = m;
                sk_s = rs;
                sk_any = true;
            } else if (rv < sk_min) {
                sk_min = rv;
                sk_m = m;
                sk_s = rs;
            }
            __syncthreads();
        }
    }

    // ---- combine (mesh.h:567-592)
    if (own_last) {
        misc[0] = lc_arg;
        misc[1] = __float_as_uint(lc_min);
    }
    __syncthreads();
    if (j == 0) {
        DpResult r;
        r.status = 0;
        const float v1min = __uint_as_float(misc[1]);
        // m = sinks[0]; replaced only by a strictly smaller value, first such row wins
        uint32_t em = snk0, es = L - 1;
        float ev = lc_snk0;
        if (v1min < lc_snk0) {
            em = misc[0];
            ev = v1min;
        }
        // sinks x columns, strict <, scan order (t asc, x asc)
        if (sk_any && sk_min < ev) {
            em = sk_m;
            es = sk_s;
            ev = sk_min;
        }
        if (!sk_any) r.status = -2;
        r.end_m = em;
        r.end_s = es;
        r.raw = ev;
        a.res[blockIdx.x] = r;
    }
}

// One thread per query: the cell walk of backtrack() (mesh.h:594-721).  Latency
// bound (dependent 4-byte loads); runs on few CUs and overlaps the next DP batch.
__global__ void backtrack_kernel(BtArgs a) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.nq) return;
    const QDesc d = a.qd[q];
    const uint32_t L = d.L;
    const uint32_t Lp = a.Lp_T * a.Lp_B;
    const uint32_t *tb = a.tb + d.tb_off;
    const uint32_t *pred_off = a.pred_off + d.poff_off;
    const uint32_t *node_pos = a.node_pos + d.node_off;
    const float *node_weight = a.node_weight + d.node_off;
    uint32_t *out = a.out_pos + d.q_off;
    const DpResult r = a.res[q];
    sina_hip_align_out o;
    o.status = r.status;
    o.end_m = r.end_m;
    o.end_s = r.end_s;
    o.raw = r.raw;
    o.sum_weight = 0.f;
    o.aligned_bases = 0;
    o.cutoff_head = o.cutoff_tail = 0;
    o.n_out = 0;
    if (r.status != 0) {
        a.out[q] = o;
        return;
    }
    const uint32_t width = a.width;
    uint32_t m = r.end_m, s = r.end_s;
    uint32_t n = 0;
    const uint32_t send = L - 1;

    // right hand overhang (:594-615)
    const int tail = (int)(send - s);
    o.cutoff_tail = tail;
    if (tail && a.overhang != SINA_OVERHANG_REMOVE) {
        int pos = (a.overhang == SINA_OVERHANG_ATTACH) ? (int)(width - 1 - node_pos[m] - (uint32_t)tail) : 0;
        for (int i = 0; i < tail; i++) {
            const int p = pos++;
            out[n++] = (uint32_t)(p > 0 ? p : 0);
        }
    }
    auto mscore = [&](uint32_t node) -> float {  // tr.s.match(sum, ab2, ab1) with comp()==true
        if (a.weights != nullptr) {
            const uint32_t nw1 = a.n_weights - 1;
            const uint32_t np = node_pos[node];
            return a.ms * a.weights[np < nw1 ? np : nw1] * node_weight[node];
        }
        return a.ms * node_weight[node];
    };
    unsigned int pos = width - 1 - node_pos[m];
    float sum_weight = 0.f;
    int aligned = 0;
    out[n++] = pos;
    aligned++;
    sum_weight = sum_weight + mscore(m);

    // :642-685
    while (s != 0 && pred_off[m + 1] != pred_off[m]) {
        const uint32_t c = tb[(size_t)m * Lp + s];
        const uint32_t snew = c & 0xffffu;
        m = c >> 16;
        if (snew != 0) {
            const uint32_t c2 = tb[(size_t)m * Lp + snew];
            if (snew == (c2 & 0xffffu)) m = c2 >> 16;
        }
        pos = width - 1 - node_pos[m];
        const float ms_w = mscore(m);
        while (s != snew) {
            --s;
            out[n++] = pos;
            aligned++;
            sum_weight = sum_weight + ms_w;
        }
    }
    // left hand overhang (:690-721)
    if (s != 0) {
        o.cutoff_head = (int)s;
        if (a.overhang == SINA_OVERHANG_ATTACH) {
            while (s-- != 0) {
                ++pos;
                out[n++] = (width - 1 < pos) ? width - 1 : pos;
            }
        } else if (a.overhang == SINA_OVERHANG_EDGE) {
            int k = (int)s;
            while (k--) out[n++] = width - (uint32_t)k - 1;
        }
    }
    o.sum_weight = sum_weight;
    o.aligned_bases = aligned;
    o.n_out = n;
    a.out[q] = o;
}

template <int T, int B>
int launch_tb(bool weighted, bool forbid, const DpArgs &a, uint32_t nq, size_t lds, hipStream_t s) {
#define SH_LAUNCH(WG, FB)                                                                            \
    do {                                                                                             \
        auto kfn = mesh_dp_kernel<T, B, WG, FB>;                                                     \
        SH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                            \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
        hipLaunchKernelGGL(kfn, dim3(nq), dim3(T), lds, s, a);                                       \
    } while (0)
    if (!weighted && !forbid) SH_LAUNCH(false, false);
    else if (weighted && !forbid) SH_LAUNCH(true, false);
    else if (!weighted && forbid) SH_LAUNCH(false, true);
    else SH_LAUNCH(true, true);
#undef SH_LAUNCH
    SH_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

static const DpGeom kGeoms[] = {{64, 4}, {128, 4}, {256, 4}, {256, 6}, {256, 8}, {512, 6}, {512, 8}, {512, 12}};

bool pick_geom(uint32_t maxL, DpGeom *g) {
    for (const DpGeom &c : kGeoms) {
        if ((uint32_t)c.Lp() >= maxL) {
            *g = c;
            return true;
        }
    }
    return false;
}

size_t dp_slot_bytes(const DpGeom &g) { return (size_t)(g.Lp() + 4) * 4 + (size_t)g.Lp() * 4 + (size_t)g.Lp() * 2; }
size_t dp_fixed_lds_bytes(const DpGeom &g) { return (size_t)12 * g.T + 256; }

int launch_mesh_dp(const DpGeom &g, bool weighted, bool forbid, const DpArgs &a, uint32_t nq,
                   size_t lds, hipStream_t s) {
    if (g.T == 64 && g.B == 4) return launch_tb<64, 4>(weighted, forbid, a, nq, lds, s);
    if (g.T == 128 && g.B == 4) return launch_tb<128, 4>(weighted, forbid, a, nq, lds, s);
    if (g.T == 256 && g.B == 4) return launch_tb<256, 4>(weighted, forbid, a, nq, lds, s);
    if (g.T == 256 && g.B == 6) return launch_tb<256, 6>(weighted, forbid, a, nq, lds, s);
    if (g.T == 256 && g.B == 8) return launch_tb<256, 8>(weighted, forbid, a, nq, lds, s);
    if (g.T == 512 && g.B == 6) return launch_tb<512, 6>(weighted, forbid, a, nq, lds, s);
    if (g.T == 512 && g.B == 8) return launch_tb<512, 8>(weighted, forbid, a, nq, lds, s);
    if (g.T == 512 && g.B == 12) return launch_tb<512, 12>(weighted, forbid, a, nq, lds, s);
    SH_FAIL("mesh_dp: unsupported geometry");
}

int launch_backtrack(const BtArgs &a, hipStream_t s) {
    const int threads = 64;
    hipLaunchKernelGGL(backtrack_kernel, dim3((a.nq + threads - 1) / threads), dim3(threads), 0, s, a);
    SH_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sina_hip
