// The scout pass of the mesh DP: a bound U on the optimum, per query, from a real path.
//
// The certified row skip of mesh_dp_simple_kernel (mesh_dp.hip) needs a number U that the optimum's
// value does not exceed: rows that provably hold no cell of a path ending at U or below are not swept, and the
// tighter U, the fewer are.  Until round 5 U was a guess -- one ratio optimum / bound per store, learnt from the
// launches before, aimed below the smallest ratio seen.  Any REAL path's cost is a valid U, and a query's own.
//
// The path: the query against the CHAIN of its family's first member -- the reference famfinder ranked first, the
// query's nearest relative by shared k-mers.  Every member of a family is a path through the family's DAG (its
// bases' nodes, consecutive ones linked: mseq.cpp:47-118), so the query's banded alignment against that chain, scored
// with the DAG's node weights and the reference's operators (mesh.h:307-374 restricted to one predecessor per row),
// is the cost of a path the full mesh also contains.  It is not the optimum: where the query differs from its
// nearest relative and another member agrees with it, the DAG's best path gains what the chain's path loses -- 50 to
// 100 units of a 4000-unit optimum on the bench's queries (tools/scout_study.py), a fifth of what the store-wide
// guess leaves.  Nothing depends on how good it is: the skipping kernel certifies its result against U whatever U
// is, the store's guess stays as a guard against a chain that lost the query (a long gap the relative does not
// share), and a failed certificate costs a second sweep, never a different result.
//
// Round 6 first built the obvious scout -- the same band swept over the whole DAG, re-centred row by row on the
// best predecessor's minimum (credited for the bases it is behind) -- which finds the optimum ITSELF for 64 of 64
// bench queries (profiles/history/r06_scout_dag_band.hip.txt).  Measured and dropped: a row of the DAG is a chain of
// dependent scattered loads (its predecessors' bands, 64 different cache lines per instruction with a lane per
// query), 30 ms per launch at 16 columns; the waves doing it take the registers of two of a SIMD's three DP waves for
// as long; in the pipeline the rows saved (0.435 -> 0.376) bought nothing and everything beside it ran a sixth slower
// (258 k -> 215 k sequences/s).  The chain needs no memory for rows at all -- one predecessor, the row before, in
// registers -- and half the rows.
//
// How it maps to the hardware: ONE LANE PER QUERY, 64 queries per wave, 144 waves for a 9216-query launch; the band
// (8 columns) and the previous row in registers; per row one packed base of the member, the DAG rows of its column
// (node_pos, row record: a merge walk, the DAG is in column order) and, every fourth row, a word of query bases.
#include <algorithm>

#include "common.h"
#include "ctx.h"

namespace sina_hip {
namespace {

constexpr float kScoutDead = 1000000.0f;  // "unreached": the reference's initial cell value (mesh.h:290)

template <int K>
__global__ void __launch_bounds__(64) chain_scout_kernel(const QDesc *__restrict__ qdv, const uint32_t *__restrict__ orderv,
                                                         const uint4 *__restrict__ recv, const uint32_t *__restrict__ node_posv,
                                                         const uint8_t *__restrict__ qmaskv, const uint32_t *__restrict__ ref_ab,
                                                         const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ chain_ref,
                                                         uint32_t nq, float ms, float mms, float gp, float gpe,
                                                         float *__restrict__ out_u) {
    static_assert(K % 4 == 0 && K >= 8 && K <= 32, "band width");
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    if (slot >= nq) return;
    const uint32_t qi = orderv[slot];  // (longest first: the lanes of a wave run about equally long)
    const QDesc d = qdv[qi];
    const uint32_t N = d.N, L = d.L;
    const uint4 *__restrict__ rec = recv + d.node_off;
    const uint32_t *__restrict__ node_pos = node_posv + d.node_off;
    const uint32_t member = chain_ref[qi];
    const uint64_t b0 = ref_off[member], b1 = ref_off[member + 1];
    const int c_max = L > (uint32_t)K ? (int)(L - (uint32_t)K) : 0;
    const float credit = (float)d.gmin * kPruneUnit;  // the smallest "best gain of a column" of this DAG

    float best = __builtin_inff();
    float pv[K], pg[K];  // the row before: value and gapm_val of its band
#pragma unroll
    for (int k = 0; k < K; k++) pv[k] = pg[k] = kScoutDead;
    int c = 0;          // band start of the row before
    int a_rel = K / 2;  // ... and where in its band its (credited) minimum sat
    bool have_prev = false;
    uint32_t m = 0;  // merge cursor into the DAG's rows (column order)
    // query mask bytes [w_at, w_at + 4 (K/4 + 1)) as words, reloaded when the band leaves them
    uint32_t w[K / 4 + 1];
#pragma unroll
    for (int i = 0; i <= K / 4; i++) w[i] = 0u;
    uint64_t w_at = ~(uint64_t)0 - 16;
    // The member's next base and the DAG rows it can sit in are asked for a row AHEAD: a row of the sweep is then
    // arithmetic on what is already there instead of a chain of three dependent loads (base -> column -> node_pos ->
    // row record), each of them a few microseconds beside the device-filling kernels of the other batches.
    uint32_t ab_next = b0 < b1 ? ref_ab[b0] : 0u;
    uint32_t wpos[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};  // node_pos and records of rows m .. m + 2
    uint4 wrec[3] = {};
    auto load_window = [&](uint32_t at) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const uint32_t x = at + (uint32_t)i;
            wpos[i] = x < N ? node_pos[x] : 0xFFFFFFFFu;
            wrec[i] = rec[x < N ? x : 0];
        }
    };
    load_window(0);
    for (uint64_t bi = b0; bi < b1; ++bi) {
        const uint32_t ab = ab_next;
        if (bi + 1 < b1) ab_next = ref_ab[bi + 1];
        const uint32_t col = ab & 0xFFFFFFu, bmask = (ab >> 24) & 31u;
        // ---- the member's node: the DAG row of this column that carries this character (mseq.cpp:83-100)
        uint32_t row = N;
        uint4 r = wrec[0];
        bool found = false;
#pragma unroll
        for (int i = 0; i < 3; i++)  // (the usual case: one of the three rows behind the previous base's)
            if (!found && wpos[i] == col && ((wrec[i].z >> 8) & 31u) == bmask) {
                found = true;
                row = m + (uint32_t)i;
                r = wrec[i];
            }
        if (!found) {  // a column with more nodes, or columns the member skips: the walk itself
            while (m < N && node_pos[m] < col) ++m;
            row = m;
            r = rec[row < N ? row : 0];
            while (row < N && ((r.z >> 8) & 31u) != bmask) {
                ++row;
                if (row < N) r = rec[row];
            }
            if (row >= N || node_pos[row] != col) break;  // (cannot happen for a member of the family: give up, keep what was found)
        }
        m = row + 1;  // (the next base sits in a later column: behind this row)
        load_window(m);
        const uint32_t mmask = (r.z >> 8) & 0xfu;
        const float wgt = __uint_as_float(r.y);
        const float vM = ms * wgt, vX = mms * wgt;  // scoring_schemes.h:154
        // ---- where the band goes: a diagonal step, one column more or less where the minimum has drifted from the
        // band's middle (the previous row stays in registers: only shifts of 0, 1 and 2 columns are wired)
        int delta = 0;
        if (have_prev) {
            delta = a_rel + 1 - K / 2;
            delta = delta < 0 ? 0 : (delta > 2 ? 2 : delta);
        }
        int cn = c + delta;
        cn = cn > c_max ? c_max : cn;
        delta = cn - c;
        c = cn;
        // ---- match / mismatch score of my K columns against this row (comp(): aligned_base.h:153)
        float csel[K];
        {
            const uint64_t at = d.q_off + (uint64_t)c;  // (the launch's mask array starts on an allocation boundary)
            const uint64_t at4 = at & ~(uint64_t)3;
            if (at4 != w_at) {
                const uint32_t *qw = reinterpret_cast<const uint32_t *>(qmaskv + at4);
                if (at4 == w_at + 4) {  // (the usual step: one word further)
#pragma unroll
                    for (int i = 0; i < K / 4; i++) w[i] = w[i + 1];
                    w[K / 4] = qw[K / 4];
                } else {
#pragma unroll
                    for (int i = 0; i <= K / 4; i++) w[i] = qw[i];  // (up to K + 6 bytes past the last query: the buffer's slack, DevBuf::reserve)
                }
                w_at = at4;
            }
            const uint32_t sh = ((uint32_t)at & 3u) * 8u;
#pragma unroll
            for (int i = 0; i < K / 4; i++) {
                const uint32_t x = sh ? ((w[i] >> sh) | (w[i + 1] << (32u - sh))) : w[i];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = 4 * i + j;
                    const uint32_t s = (uint32_t)c + (uint32_t)k;
                    const uint32_t q = s < L ? ((x >> (8 * j)) & 0xfu) : 0u;
                    csel[k] = (q & mmask) != 0u ? vM : vX;
                }
            }
        }
        const bool col0 = c == 0;  // my cell 0 is query column 0: initial value 1, no match step, no insertion
        float loc[K], gm[K];
        if (!have_prev) {
            // the chain's first node: a source row of the DAG starts at 1 everywhere (init_edge); any other row can
            // be entered for free at column 0 only -- its predecessors are not of the chain
            const bool source = (r.z & 0xffu) == 0u;
#pragma unroll
            for (int k = 0; k < K; k++) loc[k] = gm[k] = (source || (k == 0 && col0)) ? 1.0f : kScoutDead;
        } else {
            // the previous row at my columns: sv[k] = value[p][c + k - 1], sg[k] = gapm_val[p][c + k]
            float sv[K + 1], sg[K];
#pragma unroll
            for (int k = 0; k <= K; k++) {
                const float d0 = k >= 1 ? pv[k - 1] : kScoutDead;     // delta 0: the previous band's cell k - 1
                const float d1 = k < K ? pv[k] : kScoutDead;          // delta 1: cell k
                const float d2 = k + 1 < K ? pv[k + 1] : kScoutDead;  // delta 2: cell k + 1
                sv[k] = delta == 0 ? d0 : (delta == 1 ? d1 : d2);
            }
#pragma unroll
            for (int k = 0; k < K; k++) {
                const float d0 = pg[k];
                const float d1 = k + 1 < K ? pg[k + 1] : kScoutDead;
                const float d2 = k + 2 < K ? pg[k + 2] : kScoutDead;
                sg[k] = delta == 0 ? d0 : (delta == 1 ? d1 : d2);
            }
#pragma unroll
            for (int k = 0; k < K; k++) {
                const float v = sv[k + 1] + gp;  // deletion (mesh.h:307-330): open ...
                const float g = sg[k] + gpe;     // ... or extend the predecessor's gap
                const float cand = v < g ? v : g;
                gm[k] = cand;
                float mv = sv[k] + csel[k];      // match from (p, s-1) (:360-374)
                float dv = cand;
                if (k == 0 && col0) {
                    mv = __builtin_inff();
                    dv = cand < 1.0f ? cand : 1.0f;
                }
                loc[k] = mv < dv ? mv : dv;
            }
        }
        // ---- the insertion chain (mesh.h:332-358): nothing enters the band from its left
        float fv[K];
        fv[0] = loc[0];
        bool e_prev = col0 && fv[0] == 1.0f;  // column 0 keeps gaps_val = 1: it "extends" iff its value is 1
#pragma unroll
        for (int k = 1; k < K; k++) {
            const float gsx = fv[k - 1] + (e_prev ? gpe : gp);
            const bool ins = gsx <= loc[k];
            fv[k] = ins ? gsx : loc[k];
            e_prev = ins;
        }
        // ---- the row's "minimum" (by value plus a credit per query base consumed: the cells of a row have consumed
        // different numbers; the credit is the smallest gain a column offers, so a matching step still lowers it),
        // the end-cell candidates (mesh.h:569-592), and the row becomes the previous one
        float vmin = __builtin_inff(), vcred = __builtin_inff();
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool real = (uint32_t)c + (uint32_t)k < L;
            fv[k] = real ? fv[k] : kScoutDead;
            gm[k] = real ? gm[k] : kScoutDead;
            vmin = fv[k] < vmin ? fv[k] : vmin;
            const float cr = fv[k] + credit * (float)k;
            if (cr < vcred) {
                vcred = cr;
                a_rel = k;
            }
            if ((uint32_t)c + (uint32_t)k == L - 1u) best = fv[k] < best ? fv[k] : best;
            pv[k] = fv[k];
            pg[k] = gm[k];
        }
        if (r.z & kRecSink) best = vmin < best ? vmin : best;
        have_prev = true;
    }
    out_u[qi] = best;
}

}  // namespace

// chain_ref: per query the reference id of its family's first member (device); out_u: [nq]
int launch_chain_scout(const DpArgs &a, uint32_t nq, const uint32_t *ref_ab, const uint64_t *ref_off,
                       const uint32_t *chain_ref, float *out_u, hipStream_t s) {
    const uint32_t blocks = (nq + 63u) / 64u;
    hipLaunchKernelGGL((chain_scout_kernel<kScoutBand>), dim3(blocks), dim3(64), 0, s, a.qd, a.order, a.rec, a.node_pos,
                       a.qmask, ref_ab, ref_off, chain_ref, nq, a.ms, a.mms, a.gp, a.gpe, out_u);
    SH_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sina_hip
