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
//   * one workgroup per query; lane l of wave w owns B consecutive query columns
//     (wave w owns a contiguous block of 64*B columns);
//   * every wave sweeps the DAG rows in topological (id) order ON ITS OWN: the row
//     record {first pred, weight, #pred|mask|flags, spill slot} and the pred list
//     are wave-uniform and come through the scalar cache (s_load), so there is no
//     divergence on graph structure and no vector-memory latency on the row
//     critical path;
//   * the last W rows of {value, gapm_val, gapm_idx} of a wave's own columns live
//     in an LDS ring (12 B per column) that only that wave touches; the rare
//     predecessors further back than W rows are read from a per-query spill area
//     in HBM that the producing row also wrote;
//   * the only dependencies between column blocks are (a) the insertion chain
//     along the query (cell (m,s) needs the final (m,s-1)) and (b) the match
//     candidate from (p,s-1).  Inside a wave both travel by lane shuffle: each
//     lane runs the chain over its B cells serially (exact reference order),
//     first assuming that no gap enters from the left, then again with the left
//     lane's real exit state, repeated only while some exit state still changed
//     (wave vote, no barrier).  Between waves they travel through two small LDS
//     histories (boundary value and exit state per row) guarded by a per-wave
//     progress counter: wave w starts row m once wave w-1 has published it.  The
//     waves of a workgroup thus form a software pipeline skewed by one row and
//     the main loop contains no s_barrier at all;
//   * the only per-cell HBM traffic is the write-once trace-back cell
//     (value_midx:16 | value_sidx:16), row-major.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace sina_hip {

namespace {

constexpr uint32_t kNoSpill = 0xFFFFFFFFu;
constexpr int kHist = 32;  // rows of boundary/exit-state history kept per wave (power of two)

struct ChainState {  // canonical exit state of cell (m, s): what cell (m, s+1) can observe
    float v;         // final value
    uint32_t e;      // gaps_val == value  (the "extend" condition, mesh.h:340)
    uint32_t gsi;    // gaps_idx (only meaningful when e)
    uint32_t gmax;   // gaps_max (FORBID only, only meaningful when e)
};

__device__ __forceinline__ bool same_state(const ChainState &a, const ChainState &b) {
    return a.v == b.v && a.e == b.e && (!a.e || (a.gsi == b.gsi && a.gmax == b.gmax));
}

// value of `x` in lane-1 (lane 0 gets an unspecified value): one DPP move, no LDS round trip
__device__ __forceinline__ uint32_t lane_shr1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ float lane_shr1(float x) { return __uint_as_float(lane_shr1(__float_as_uint(x))); }

// calls f(integral_constant<S>) for the (wave-uniform) run-time slot number: a scalar branch
// chain, so that register-resident ring rows are only ever indexed with compile-time constants
template <int S, int RW, typename F>
__device__ __forceinline__ void slot_dispatch(uint32_t slot, F &&f) {
    if constexpr (S < RW) {
        if (slot == (uint32_t)S) f(std::integral_constant<int, S>{});
        else slot_dispatch<S + 1, RW>(slot, f);
    }
}

__device__ __forceinline__ uint32_t lds_load_relaxed(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// RW == 0: the ring of recent rows lives in LDS (run-time depth W).
// RW  > 0: the ring lives in REGISTERS (RW rows x B cells x {value, gapm_val, gapm_idx} per lane):
//          the register file of a CU is 3x its LDS, near predecessors cost no LDS traffic at all,
//          and the ring can be deeper at the same occupancy, so far fewer rows spill to HBM.
template <int T, int B, int RW, bool WEIGHTED, bool FORBID>
__global__ void __launch_bounds__(T, (RW > 0 ? 2 : (B <= 6 ? 4 : (B <= 8 ? 3 : 2))))
mesh_dp_kernel(const QDesc *__restrict__ qdv, const uint4 *__restrict__ recv, const uint32_t *__restrict__ predv,
               const uint32_t *__restrict__ node_posv, const uint32_t *__restrict__ succ_minposv,
               const uint8_t *__restrict__ qmaskv, const float *__restrict__ weights, uint32_t n_weights,
               uint32_t *__restrict__ tbv, float *__restrict__ dbg_value, float *spillv,
               DpResult *__restrict__ resv, float ms, float mms, float gp, float gpe, int W) {
    constexpr int Lp = T * B;
    constexpr int NW = T / 64;
    constexpr int WCOLS = 64 * B;  // columns per wave
    static_assert(T % 64 == 0, "whole waves only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int j = threadIdx.x;
    const int lane = j & 63;
    const int w = j >> 6;
    const QDesc d = qdv[blockIdx.x];
    const uint32_t N = d.N, L = d.L;
    const uint32_t s0 = (uint32_t)j * B;

    // ---- LDS carve (all dynamic; every offset a multiple of 16)
    uint32_t *progress = reinterpret_cast<uint32_t *>(smem);                    // [NW] rows done by wave
    float *bnd_val = reinterpret_cast<float *>(smem + 64);                      // [NW][kHist]
    float *xs_v = bnd_val + NW * kHist;                                          // [NW][kHist]
    uint32_t *xs_e = reinterpret_cast<uint32_t *>(xs_v + NW * kHist);            // [NW][kHist]
    uint32_t *xs_gmax = xs_e + NW * kHist;                                       // [NW][kHist]
    uint32_t *fin = xs_gmax + NW * kHist;                                        // [NW][16] per-wave results
    unsigned char *ring = smem + 64 + 16 * NW * kHist + 64 * NW;
    constexpr size_t kValBytes = (size_t)Lp * 4;
    constexpr size_t kGmBytes = (size_t)Lp * 4;
    constexpr size_t kSlotBytes = kValBytes + kGmBytes + (size_t)Lp * 4;  // value f32 | gapm_val f32 | gapm_idx u32

    const uint4 *__restrict__ rec = recv + d.node_off;
    const uint32_t *__restrict__ pred = predv + d.edge_off;
    const uint32_t *__restrict__ node_pos = node_posv + d.node_off;
    const uint32_t *__restrict__ succ_minpos = succ_minposv + d.node_off;
    uint32_t *__restrict__ tb = tbv + d.tb_off;
    float *spill = spillv + d.spill_off * (size_t)(3 * Lp);

    if (j < NW) progress[j] = 0;
    __syncthreads();

    // query masks of my columns (0 beyond L: never matches, never stored)
    uint32_t qm[B];
#pragma unroll
    for (int k = 0; k < B; k++) {
        const uint32_t s = s0 + k;
        qm[k] = (s < L) ? (uint32_t)(qmaskv[d.q_off + s] & 0xf) : 0u;
    }

    // end-cell search state (mesh.h:567-592)
    const bool own_last = (L - 1) / B == (uint32_t)j;
    const int k_last = (int)((L - 1) % B);
    float lc_min = 0.f, lc_snk0 = 0.f;  // step 1: rows at column L-1 (lane own_last only)
    uint32_t lc_arg = 0;
    bool lc_any = false;
    float sk_min = __builtin_inff();  // step 2: sink rows x my wave's columns, first in scan order
    uint32_t sk_m = 0, sk_s = 0xffffffffu, snk0 = 0;
    bool sk_any = false;

    const uint32_t throttle = (uint32_t)(kHist - W - 2);
    uint4 cur = rec[0];
    uint32_t my_slot = 0;  // ring slot of the current row (m % W), advanced incrementally
    struct RingRow {  // one ring row of this lane's cells, register resident when RW > 0
        float v[B], g[B];
        uint32_t i[B];
    };
    RingRow r0, r1, r2, r3, r4, r5;  // distinct objects (not an array) so that they scalarise
    for (uint32_t m = 0; m < N; ++m) {
        const uint4 nxt = rec[m + 1 < N ? m + 1 : m];  // scalar prefetch of the next row record
        const uint32_t pb = cur.x;
        const float wgt = __uint_as_float(cur.y);
        const uint32_t npred = cur.z & 0xffu;
        const uint32_t mmask = (cur.z >> 8) & 0xfu;
        const bool is_sink = ((cur.z >> 16) & 1u) != 0;
        const uint32_t sp = cur.w;
        const bool edge_row = (npred == 0);
        uint32_t mpos = 0;
        float cM, cX, gd_open, gd_ext, gi_open;
        if constexpr (WEIGHTED) {
            mpos = node_pos[m];
            const uint32_t nw1 = n_weights - 1;
            const float wp = weights[mpos < nw1 ? mpos : nw1];
            const float wp1 = weights[mpos + 1 < nw1 ? mpos + 1 : nw1];
            cM = ms * wp * wgt;  // (c * weights[pos]) * weight, scoring_schemes.h:232
            cX = mms * wp * wgt;
            gd_open = gp * wp;   // :211
            gd_ext = gpe * wp;   // :222
            gi_open = gp * wp1;  // :187
        } else {
            cM = ms * wgt;       // scoring_schemes.h:154
            cX = mms * wgt;
            gd_open = gp;
            gd_ext = gpe;
            gi_open = gp;
        }
        uint32_t smax = 0;
        if constexpr (FORBID) {
            if (!WEIGHTED) mpos = node_pos[m];
            // int max_insert = min_mpos - pos - 1, passed as unsigned idx_type (mesh.h:480-489)
            smax = (uint32_t)(int)(succ_minpos[m] - mpos - 1);
        }
        const float init_v = edge_row ? 1.0f : 1000000.0f;

        // ---- wave pipeline hand-shake (LDS flags, no barrier)
        if (w > 0) {  // the wave to my left must have published row m
            while (lds_load_relaxed(&progress[w - 1]) <= m) __builtin_amdgcn_s_sleep(1);
        }
        if (w < NW - 1) {  // do not lap the history slots the wave to my right still needs
            while (m >= lds_load_relaxed(&progress[w + 1]) + throttle) __builtin_amdgcn_s_sleep(1);
        }
        // LDS is in-order per CU: everything the publishing wave wrote before its progress
        // store is visible once the counter is; only the compiler must not hoist loads.
        asm volatile("" ::: "memory");

        // ---- phase 1: deletion + match candidates from every predecessor row
        float dv[B], gm[B], mt[B], csel[B];
        uint32_t dvm[B], dvs[B], gmi[B], mtp[B];
#pragma unroll
        for (int k = 0; k < B; k++) {
            const float iv = (s0 + k == 0) ? 1.0f : init_v;
            dv[k] = iv;
            gm[k] = iv;
            mt[k] = __builtin_inff();
            dvm[k] = 0;
            dvs[k] = 0;
            gmi[k] = 0;
            mtp[k] = 0;
            csel[k] = (mmask & qm[k]) ? cM : cX;  // comp(): optimistic IUPAC match (aligned_base.h:153)
        }
        auto relax = [&](uint32_t p, const float(&sv)[B], const float(&sg)[B], const uint32_t(&sgi)[B], float svl) {
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
                const bool better = cand < dv[k];
                dv[k] = better ? cand : dv[k];
                dvm[k] = better ? cm : dvm[k];
                dvs[k] = better ? s0 + k : dvs[k];  // value_sidx of a deletion is the column itself
                // match from (p, s-1) (mesh.h:360-374); first predecessor with the minimum wins
                const float pvv = (k == 0) ? svl : sv[k - 1];
                const float mv = pvv + csel[k];
                const bool mb = ((s0 + k) > 0) && (mv < mt[k]);
                mt[k] = mb ? mv : mt[k];
                mtp[k] = mb ? p : mtp[k];
            }
        };
        // pred entry: id | ring slot << 16 | far << 31.  Ids ascend, so the predecessors beyond
        // the LDS ring (spill rows in HBM) come first.
        uint32_t e = 0;
        for (; e < npred; ++e) {
            const uint32_t pe = pred[pb + e];
            if (!(pe >> 31)) break;
            const uint32_t p = pe & 0xffffu;
            const float *row = spill + (size_t)rec[p].w * (3 * Lp);
            float sv[B], sg[B];
            uint32_t sgi[B];
#pragma unroll
            for (int k = 0; k < B; k++) {
                sv[k] = row[s0 + k];
                sg[k] = row[Lp + s0 + k];
                sgi[k] = reinterpret_cast<const uint32_t *>(row)[2 * Lp + s0 + k];
            }
            float svl = lane_shr1(sv[B - 1]);  // value[p][s0-1] lives in the lane to my left
            if (lane == 0 && w > 0) svl = row[s0 - 1];
            relax(p, sv, sg, sgi, svl);
        }
        for (; e < npred; ++e) {
            const uint32_t pe = pred[pb + e];
            const uint32_t p = pe & 0xffffu;
            if constexpr (RW > 0) {
                const uint32_t sl = (pe >> 16) & 0xffu;
                float bnd = 0.f;
                if (lane == 0 && w > 0) bnd = bnd_val[(w - 1) * kHist + (p & (kHist - 1))];
#define SH_RELAX_FROM(R)                                          \
    {                                                             \
        float svl = lane_shr1(R.v[B - 1]);                        \
        if (lane == 0 && w > 0) svl = bnd;                        \
        relax(p, R.v, R.g, R.i, svl);                             \
    }
                if (sl == 0) SH_RELAX_FROM(r0)
                else if (RW > 1 && sl == 1) SH_RELAX_FROM(r1)
                else if (RW > 2 && sl == 2) SH_RELAX_FROM(r2)
                else if (RW > 3 && sl == 3) SH_RELAX_FROM(r3)
                else if (RW > 4 && sl == 4) SH_RELAX_FROM(r4)
                else if (RW > 5 && sl == 5) SH_RELAX_FROM(r5)
#undef SH_RELAX_FROM
            } else {
                const unsigned char *slot = ring + (size_t)((pe >> 16) & 0xffu) * kSlotBytes;
                const float *pv = reinterpret_cast<const float *>(slot);
                const float *pg = reinterpret_cast<const float *>(slot + kValBytes);
                const uint32_t *pi = reinterpret_cast<const uint32_t *>(slot + kValBytes + kGmBytes);
                float sv[B], sg[B];
                uint32_t sgi[B];
#pragma unroll
                for (int k = 0; k < B; k++) {
                    sv[k] = pv[s0 + k];
                    sg[k] = pg[s0 + k];
                    sgi[k] = pi[s0 + k];
                }
                float svl = lane_shr1(sv[B - 1]);
                if (lane == 0 && w > 0) svl = bnd_val[(w - 1) * kHist + (p & (kHist - 1))];  // left wave's boundary
                relax(p, sv, sg, sgi, svl);
            }
        }

        // ---- phase 2: insertion chain along my B cells
        float fv[B];
        uint32_t fvm[B], fvs[B];
        ChainState ex;
        auto ext_cost = [&](uint32_t s, uint32_t gsi_prev) -> float {
            if constexpr (WEIGHTED) {
                const uint32_t nw1 = n_weights - 1;
                const uint32_t wi = mpos + 1 + ((s - 1) - gsi_prev);
                return gpe * weights[wi < nw1 ? wi : nw1];
            } else {
                return gpe;
            }
        };
        auto run_chain = [&](const ChainState &left) {
            ChainState c = left;
#pragma unroll
            for (int k = 0; k < B; k++) {
                const uint32_t s = s0 + k;
                float v = dv[k];
                uint32_t vm = dvm[k];
                uint32_t vs = dvs[k];
                float gs = 1.0f;  // init_edge at s == 0, no insertion step there
                uint32_t gsi = 0, gmax = 0;
                if (s > 0) {
                    bool ins = true;
                    float gi_cost = gi_open;  // opening gap (mesh.h:340-343 / :415-419)
                    uint32_t gsi_n = s - 1, gmax_n = 0;
                    if (FORBID) {
                        ins = (smax >= 1) && (!c.e || c.gmax > 0);
                        gmax_n = c.e ? c.gmax - 1 : smax - 1;
                    }
                    if (c.e) {  // extending gap (:344-349 / :420-425); gaps_val == value here
                        gi_cost = ext_cost(s, c.gsi);
                        gsi_n = c.gsi;
                    }
                    gs = ins ? (c.v + gi_cost) : init_v;  // untouched cell keeps its initial gaps_*
                    gsi = ins ? gsi_n : 0u;
                    gmax = ins ? gmax_n : 0u;
                    const bool take = ins && (gs <= v);  // mesh.h:351-357
                    v = take ? gs : v;
                    vm = take ? m : vm;
                    vs = take ? gsi : vs;
                    const bool mtk = mt[k] < v;
                    v = mtk ? mt[k] : v;
                    vm = mtk ? mtp[k] : vm;
                    vs = mtk ? s - 1 : vs;
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

        // speculative pass: pretend the cell to my left did not end in a gap and is so
        // expensive that opening from it can never win.
        ChainState left;
        left.v = __builtin_inff();
        left.e = 0;
        left.gsi = 0;
        left.gmax = 0;
        run_chain(left);
        // lane 0 of wave > 0 knows its real left state already (published by the left wave)
        ChainState wave_left;
        wave_left.v = 0.f;
        wave_left.e = wave_left.gsi = wave_left.gmax = 0;
        if (w > 0) {
            const int h = (w - 1) * kHist + (int)(m & (kHist - 1));
            wave_left.v = xs_v[h];
            const uint32_t pe_ = xs_e[h];
            wave_left.e = pe_ >> 31;
            wave_left.gsi = pe_ & 0x7fffffffu;
            wave_left.gmax = FORBID ? xs_gmax[h] : 0u;
        }
        // The speculation is exact unless the gap arriving from the left wins my first cell
        // (gs <= value after deletions, and no match beats it): one add and two compares per
        // lane verify that.  Only if some lane's first cell does take the gap are the chains
        // re-run with the real left states, until no exit state changes any more.  (With
        // --insertion=forbid a cell that may NOT take a gap keeps its initial gaps_val, which
        // the shortcut cannot see: always re-run there.)
        bool rerun = true;
        if constexpr (!FORBID) {
            left.v = lane_shr1(ex.v);
            left.e = lane_shr1(ex.e);
            left.gsi = lane_shr1(ex.gsi);
            if (lane == 0) left = wave_left;
            bool take0 = false;
            if (j > 0) {
                const float g0 = left.v + (left.e ? ext_cost(s0, left.gsi) : gi_open);
                take0 = (g0 <= dv[0]) && !(mt[0] < g0);
            }
            rerun = __any(take0);
        }
        if (rerun) {
            for (;;) {
                const ChainState prev = ex;
                left.v = lane_shr1(ex.v);
                left.e = lane_shr1(ex.e);
                left.gsi = lane_shr1(ex.gsi);
                left.gmax = FORBID ? lane_shr1(ex.gmax) : 0u;
                if (lane == 0) left = wave_left;
                if (j > 0) run_chain(left);
                if (!__any(!same_state(ex, prev))) break;
            }
        }

        // ---- publish: ring (own columns), boundary + exit state for the wave to my right
        {
            if constexpr (RW > 0) {
#define SH_STORE_TO(R)                  \
    {                                   \
        _Pragma("unroll") for (int k = 0; k < B; k++) { \
            R.v[k] = fv[k];             \
            R.g[k] = gm[k];             \
            R.i[k] = gmi[k];            \
        }                               \
    }
                if (my_slot == 0) SH_STORE_TO(r0)
                else if (RW > 1 && my_slot == 1) SH_STORE_TO(r1)
                else if (RW > 2 && my_slot == 2) SH_STORE_TO(r2)
                else if (RW > 3 && my_slot == 3) SH_STORE_TO(r3)
                else if (RW > 4 && my_slot == 4) SH_STORE_TO(r4)
                else if (RW > 5 && my_slot == 5) SH_STORE_TO(r5)
#undef SH_STORE_TO
            } else {
                unsigned char *myslot = ring + (size_t)my_slot * kSlotBytes;
                float *wv = reinterpret_cast<float *>(myslot);
                float *wg = reinterpret_cast<float *>(myslot + kValBytes);
                uint32_t *wi = reinterpret_cast<uint32_t *>(myslot + kValBytes + kGmBytes);
#pragma unroll
                for (int k = 0; k < B; k++) {
                    wv[s0 + k] = fv[k];
                    wg[s0 + k] = gm[k];
                    wi[s0 + k] = gmi[k];
                }
            }
            if (lane == 63) {
                const int h = w * kHist + (int)(m & (kHist - 1));
                bnd_val[h] = fv[B - 1];
                xs_v[h] = ex.v;
                xs_e[h] = (ex.e << 31) | ex.gsi;
                if (FORBID) xs_gmax[h] = ex.gmax;
            }
            my_slot = (my_slot + 1 == (uint32_t)W) ? 0u : my_slot + 1;
        }
        if (sp != kNoSpill) {
            float *row = spill + (size_t)sp * (3 * Lp);
#pragma unroll
            for (int k = 0; k < B; k++) {
                row[s0 + k] = fv[k];
                row[Lp + s0 + k] = gm[k];
                reinterpret_cast<uint32_t *>(row)[2 * Lp + s0 + k] = gmi[k];
            }
        }
        // release: LDS writes (and the spill row, if any) before the progress counter.  Rows
        // without a spill row only need the LDS queue drained -- a full workgroup-scope release
        // would also wait (vmcnt) for the previous row's trace-back stores.
        if (sp != kNoSpill) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (lane == 0) __hip_atomic_store(&progress[w], m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

        // ---- trace-back cells: the only per-cell HBM traffic
        {
            uint32_t *trow = tb + (size_t)m * Lp + s0;
#pragma unroll
            for (int k = 0; k < B; k++) trow[k] = (fvm[k] << 16) | (fvs[k] & 0xffffu);
        }
        if (dbg_value != nullptr && blockIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < B; k++) dbg_value[(size_t)m * Lp + s0 + k] = fv[k];
        }

        // ---- end-cell search, step 1: rows at the last query column (one lane)
        if (own_last) {
            float v = fv[0];
#pragma unroll
            for (int k = 1; k < B; k++) v = (k == k_last) ? fv[k] : v;
            if (!lc_any || v < lc_min) {
                lc_min = v;
                lc_arg = m;
                lc_any = true;
            }
            if (is_sink && !sk_any) lc_snk0 = v;  // value of sinks[0] at column L-1
        }
        // step 2: sink rows x every column; each wave keeps the best of its own columns
        if (is_sink) {
            float bv = __builtin_inff();
            uint32_t bs = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < B; k++) {
                const uint32_t s = s0 + k;
                const bool b = (s < L) && (fv[k] < bv);
                bv = b ? fv[k] : bv;
                bs = b ? s : bs;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {  // smaller value, then smaller column
                const float ov = __shfl_xor(bv, off);
                const uint32_t os = __shfl_xor(bs, off);
                const bool b = (ov < bv) || (ov == bv && os < bs);
                bv = b ? ov : bv;
                bs = b ? os : bs;
            }
            if (!sk_any) snk0 = m;  // sinks ascend with the row id: the first one is sinks[0]
            if (bv < sk_min) {       // strict <: the first sink row with the minimum wins
                sk_min = bv;
                sk_m = m;
                sk_s = bs;
            }
            sk_any = true;
        }
        cur = nxt;
    }

    // ---- combine (mesh.h:567-592)
    if (lane == 0) {
        uint32_t *f = fin + w * 16;
        f[0] = __float_as_uint(sk_min);
        f[1] = sk_m;
        f[2] = sk_s;
        f[3] = snk0;
        f[4] = sk_any ? 1u : 0u;
    }
    if (own_last) {
        uint32_t *f = fin + w * 16;
        f[8] = 1u;
        f[9] = lc_arg;
        f[10] = __float_as_uint(lc_min);
        f[11] = __float_as_uint(lc_snk0);
    }
    __syncthreads();
    if (j == 0) {
        DpResult r;
        r.status = 0;
        // step 2 across waves: min value, then smaller sink id, then smaller column
        float bmin = __builtin_inff();
        uint32_t bm = 0, bs = 0;
        bool any = false;
        for (int x = 0; x < NW; x++) {
            const uint32_t *f = fin + x * 16;
            if (!f[4]) continue;
            const float v = __uint_as_float(f[0]);
            if (!any || v < bmin || (v == bmin && (f[1] < bm || (f[1] == bm && f[2] < bs)))) {
                bmin = v;
                bm = f[1];
                bs = f[2];
            }
            any = true;
        }
        const int wl = (int)(((L - 1) / B) >> 6);  // wave of the lane that owns column L-1
        const uint32_t *fl = fin + wl * 16;
        const float v1min = __uint_as_float(fl[10]);
        const float v_snk0 = __uint_as_float(fl[11]);
        // m = sinks[0]; replaced only by a strictly smaller value, first such row wins
        uint32_t em = fin[3], es = L - 1;
        float ev = v_snk0;
        if (v1min < v_snk0) {
            em = fl[9];
            ev = v1min;
        }
        // sinks x columns, strict <, scan order (t asc, x asc)
        if (any && bmin < ev) {
            em = bm;
            es = bs;
            ev = bmin;
        }
        if (!any) r.status = -2;
        r.end_m = em;
        r.end_s = es;
        r.raw = ev;
        resv[blockIdx.x] = r;
    }
}

// One thread per query: the cell walk of backtrack() (mesh.h:594-721).  Latency
// bound (dependent 4-byte loads); runs on few CUs and overlaps the next DP batch.
__global__ void backtrack_kernel(BtArgs a) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.nq) return;
    const QDesc d = a.qd[q];
    const uint32_t L = d.L;
    const uint32_t Lp = a.Lp;
    const uint32_t *tb = a.tb + d.tb_off;
    const uint4 *rec = a.rec + d.node_off;
    const uint32_t *node_pos = a.node_pos + d.node_off;
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
        const float wgt = __uint_as_float(rec[node].y);
        if (a.weights != nullptr) {
            const uint32_t nw1 = a.n_weights - 1;
            const uint32_t np = node_pos[node];
            return a.ms * a.weights[np < nw1 ? np : nw1] * wgt;
        }
        return a.ms * wgt;
    };
    unsigned int pos = width - 1 - node_pos[m];
    float sum_weight = 0.f;
    int aligned = 0;
    out[n++] = pos;
    aligned++;
    sum_weight = sum_weight + mscore(m);

    // :642-685 (a source node has no predecessors)
    while (s != 0 && (rec[m].z & 0xffu) != 0) {
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

template <int T, int B, int RW>
int launch_tb(bool weighted, bool forbid, const DpArgs &a, uint32_t nq, size_t lds, hipStream_t s) {
#define SH_LAUNCH(WG, FB)                                                                               \
    do {                                                                                                \
        auto kfn = mesh_dp_kernel<T, B, RW, WG, FB>;                                                        \
        SH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                               \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
        hipLaunchKernelGGL(kfn, dim3(nq), dim3(T), lds, s, a.qd, a.rec, a.pred, a.node_pos, a.succ_minpos, \
                           a.qmask, a.weights, a.n_weights, a.tb, a.dbg_value, a.spill, a.res, a.ms, a.mms, \
                           a.gp, a.gpe, a.W);                                                           \
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

// Few, fat lanes win: the per-row fixed work of a wave (row record, hand-shake, chain exchange,
// publish) is amortised over more cells.  Third field: register-ring depth (0 = LDS ring).
// Measured on MI355X (16S, 1024 queries, Gcell/s): LDS ring 128x12 152, 256x6 133, 512x3 97;
// register ring 256x6 depth 4: 90, 192x8 depth 4: 82 (2 waves/SIMD: occupancy beats ring depth).
static const DpGeom kGeoms[] = {{64, 4, 0},  {64, 8, 0},   {64, 12, 0},  {128, 8, 0}, {128, 12, 0},
                                {256, 8, 0}, {256, 12, 0}, {512, 8, 0}, {512, 12, 0}};

bool pick_geom(uint32_t maxL, DpGeom *g) {
    // tuning override: SINA_HIP_DP_GEOM="T,B,RW" (used if it covers the batch's longest query)
    if (const char *ov = getenv("SINA_HIP_DP_GEOM")) {
        int t = 0, b = 0, r = 0;
        if (sscanf(ov, "%d,%d,%d", &t, &b, &r) >= 2 && (uint32_t)(t * b) >= maxL) {
            g->T = t;
            g->B = b;
            g->RW = r;
            return true;
        }
    }
    for (const DpGeom &c : kGeoms) {
        if ((uint32_t)c.Lp() >= maxL) {
            *g = c;
            return true;
        }
    }
    return false;
}

size_t dp_slot_bytes(const DpGeom &g) { return g.RW > 0 ? 0 : (size_t)g.Lp() * 12; }
size_t dp_fixed_lds_bytes(const DpGeom &g) {
    const size_t nw = (size_t)g.T / 64;
    return 64 + 16 * nw * kHist + 64 * nw;
}
int dp_max_ring(const DpGeom &) { return kHist - 4; }

int launch_mesh_dp(const DpGeom &g, bool weighted, bool forbid, const DpArgs &a, uint32_t nq,
                   size_t lds, hipStream_t s) {
#define SH_GEOM(TT, BB, RR) \
    if (g.T == TT && g.B == BB && g.RW == RR) return launch_tb<TT, BB, RR>(weighted, forbid, a, nq, lds, s)
    SH_GEOM(64, 4, 0);
    SH_GEOM(64, 8, 0);
    SH_GEOM(64, 12, 0);
    SH_GEOM(128, 8, 0);
    SH_GEOM(128, 12, 0);
    SH_GEOM(256, 8, 0);
    SH_GEOM(256, 12, 0);
    SH_GEOM(512, 8, 0);
    SH_GEOM(512, 12, 0);
    // tuning alternatives (SINA_HIP_DP_GEOM)
    SH_GEOM(256, 6, 0);
    SH_GEOM(256, 6, 4);
#undef SH_GEOM
    SH_FAIL("mesh_dp: unsupported geometry");
}

int launch_backtrack(const BtArgs &a, hipStream_t s) {
    const int threads = 64;
    hipLaunchKernelGGL(backtrack_kernel, dim3((a.nq + threads - 1) / threads), dim3(threads), 0, s, a);
    SH_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sina_hip
