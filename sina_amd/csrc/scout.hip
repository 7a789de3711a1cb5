// The scout pass of the mesh DP: a bound U on the optimum, per query, from a real path.
//
// The certified row skip of mesh_dp_simple_kernel (mesh_dp.hip) needs a number U that the optimum's
// value does not exceed: rows that provably hold no cell of a path ending at U or below are not swept, and the
// tighter U, the fewer are.  Until round 5 U was a guess -- one ratio optimum / bound per store, learnt from the
// launches before, aimed below the smallest ratio seen: 3 % more rows than the queries' own optima would need on the
// bench's homogeneous queries, and everybody's band as wide as the most distant query's on mixed input.  Any REAL
// path's cost is a valid U.  This kernel finds one per query: the same recurrence (mesh.h:307-374, the reference's
// operators and tie rules, as restated in mesh_dp.hip) over a band of K query columns per DAG row, the band
// re-centred, row by row, on the column where the best predecessor row has its minimum; cells outside a row's band
// count as unreached (1e6, the reference's own initial value, mesh.h:290).  What comes out is the value of a path
// the full mesh also contains -- on the bench's queries (3 % substitutions, indels) and on queries four times as
// distant it IS the optimum, bit for bit, already at K = 8 (tools/scout_study.py).  Nothing depends on that: the
// skipping kernel certifies its result against U whatever U is, and sweeps again if the certificate fails.
//
// How it maps to the hardware: ONE LANE PER QUERY.  A band of 16 cells per row is a dozen instructions per cell
// -- as a wave per query it would be 3000 rows x 200 wave-instructions of mostly idle lanes (a tenth of the main
// sweep); a lane per query shares every instruction among 64 queries: 150 waves for a 9216-query launch, a per
// cent of the main kernel's instruction issue.  The price is latency -- a row is a chain of dependent loads (row
// record -> predecessor headers -> predecessor bands, each lane its own addresses) -- which nobody waits for: the
// kernel runs on the context's own stream beside the device-filling kernels of the other batches in flight, like
// the lane walk of the trace-back (mesh_dp.hip backtrack_lanes_kernel).  Finished rows go to a scratch array in
// HBM (header {band start, column of the minimum, minimum} + K values + K gapm values: 144 bytes per row at
// K = 16), read back by their successors at whatever offset the two bands have to each other.
#include <algorithm>

#include "common.h"
#include "ctx.h"

namespace sina_hip {
namespace {

constexpr float kScoutDead = 1000000.0f;  // "unreached": the reference's initial cell value (mesh.h:290)

typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load at a 4-byte aligned address

constexpr int kScoutRing = 32;  // rows whose headers {band start, column of the minimum, minimum} are kept in LDS

template <int K>
__global__ void __launch_bounds__(64) mesh_scout_kernel(const QDesc *__restrict__ qdv, const uint32_t *__restrict__ orderv,
                                                        const uint4 *__restrict__ recv, const uint32_t *__restrict__ predv,
                                                        const uint8_t *__restrict__ qmaskv, float *scratch,
                                                        uint32_t Lp, uint32_t nq, float ms, float mms, float gp, float gpe,
                                                        float *__restrict__ out_u, uint64_t rows_total) {
    static_assert(K % 4 == 0 && K >= 8 && K <= 32, "band width");
    constexpr int kRow = 4 + 2 * K;  // floats per stored row
    // headers of the last kScoutRing rows of every lane's query: what a row needs of its predecessors BEFORE it can
    // ask for their bands (where its own band goes, at which offset theirs lie) comes out of LDS, not out of a
    // second round trip to memory -- a row is one round trip: the predecessors' bands
    __shared__ uint2 ring[kScoutRing][64];
    const int lane = threadIdx.x;
    // Two sweeps per query, in different waves: they differ in how much a predecessor row that is AHEAD in the query
    // is held back when a row chooses whom to follow (below) -- 0.5 and 0.9 of the sweep's own gain per base.  Either
    // value is the cost of a real path; the skipping kernel takes the smaller.  (One setting loses the optimum behind
    // a long gap early in the query, the other one on a stretch of weak matches: tools/scout_study.py.)
    const uint32_t blocks_per_sweep = (nq + 63u) / 64u;
    const uint32_t variant = blockIdx.x / blocks_per_sweep;
    const float rate_frac = variant ? 0.9f : 0.5f;
    const uint32_t slot = (blockIdx.x - variant * blocks_per_sweep) * 64u + threadIdx.x;
    if (slot >= nq) return;
    const uint32_t qi = orderv[slot];  // (longest first: the lanes of a wave run about equally long)
    const QDesc d = qdv[qi];
    const uint32_t N = d.N, L = d.L;
    const uint4 *__restrict__ rec = recv + d.node_off;
    const uint32_t *__restrict__ pred = predv + d.edge_off;
    // (rows of this query: the launch's trace-back rows are numbered the same way, tb_off = rows before it * Lp;
    // K + 8 floats of slack in front of the array: a band read at an offset may start before its row.  NOT
    // restrict: a row is written and, a row later, read through this one pointer)
    float *rows = scratch + (K + 8) + ((size_t)variant * rows_total + (size_t)(d.tb_off / Lp)) * kRow;
    const int c_max = L > (uint32_t)K ? (int)(L - (uint32_t)K) : 0;

    float best = __builtin_inff();
    int c_prev = 0;
    const float credit = (float)d.gmin * kPruneUnit;  // the smallest "best gain of a column" of this DAG
    float last_v = 0.f;  // the row before: value and column of its (credited) minimum
    int last_a = 0;
    // row record and first four predecessor entries, fetched a row ahead (they do not depend on the sweep)
    uint4 r = rec[0];
    uint4 pe4 = *reinterpret_cast<const uint4 *>(pred + (r.x & ~3u));
    for (uint32_t m = 0; m < N; ++m) {
        const uint4 r_next = rec[m + 1 < N ? m + 1 : m];
        const uint32_t npred = r.z & 0xffu, mmask = (r.z >> 8) & 0xfu;
        const float wgt = __uint_as_float(r.y);
        const float vM = ms * wgt, vX = mms * wgt;  // scoring_schemes.h:154
        // predecessor e of this row (the first few out of the words fetched ahead)
        const uint32_t pb = r.x;
        auto pred_id = [&](uint32_t e) -> uint32_t {
            const uint32_t at = (pb & 3u) + e;
            const uint32_t w = at == 0 ? pe4.x : (at == 1 ? pe4.y : (at == 2 ? pe4.z : (at == 3 ? pe4.w : pred[pb + e])));
            return w & 0xffffu;
        };
        // header of predecessor row p: {band start | column of its minimum << 16, minimum}
        auto header = [&](uint32_t p) -> uint2 {
            if (m - p <= (uint32_t)kScoutRing) return ring[p % kScoutRing][lane];
            const float4_u h = *reinterpret_cast<const float4_u *>(rows + (size_t)p * kRow);
            return uint2{(uint32_t)__float_as_int(h.x) | ((uint32_t)__float_as_int(h.y) << 16), __float_as_uint(h.z)};
        };
        // ---- where the band goes: a diagonal step behind the "minimum" of the predecessor row to follow.  Rows that
        // have consumed different numbers of query bases do not compare by value alone: behind a long gap of the query
        // (a family member with the same gap gives the DAG an edge across it) the rows just above hold cells that went
        // on matching the query against the gap's columns -- lower values, a worse path.  Each candidate is credited
        // with what the bases it is behind would gain at a fraction of the rate the sweep has gained so far.
        int c = c_prev;
        if (npred != 0) {
            const float rate = rate_frac * fmaxf(0.f, -last_v) / fmaxf(1.f, (float)last_a);
            float bv = __builtin_inff();
            int ba = 0;
            for (uint32_t e = 0; e < npred; ++e) {
                const uint2 h = header(pred_id(e));
                const int a = (int)(h.x >> 16);
                const float score = __uint_as_float(h.y) + rate * (float)a;
                if (score < bv) {
                    bv = score;
                    ba = a;
                }
            }
            c = ba + 1 - K / 2;
        }
        c = c < 0 ? 0 : (c > c_max ? c_max : c);
        c_prev = c;
        // ---- match / mismatch score of my K columns against this row (comp(): aligned_base.h:153).  The K mask
        // bytes from column c on as K/4 + 1 aligned words, shifted into place: a byte load per column would be K
        // memory instructions of 64 cache lines each -- the kernel is bound by those, not by arithmetic
        float csel[K];
        {
            const uint64_t at = d.q_off + (uint64_t)c;  // (the launch's mask array starts on an allocation boundary)
            const uint32_t sh = ((uint32_t)at & 3u) * 8u;
            const uint32_t *qw = reinterpret_cast<const uint32_t *>(qmaskv + (at & ~(uint64_t)3));
            uint32_t w[K / 4 + 1];
#pragma unroll
            for (int i = 0; i <= K / 4; i++) w[i] = qw[i];  // (up to K + 6 bytes past the last query: inside the buffer's slack, DevBuf::reserve)
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
        if (npred == 0) {  // a source row: every cell starts at 1 (init_edge) and stays untouched
#pragma unroll
            for (int k = 0; k < K; k++) loc[k] = gm[k] = 1.0f;
        } else {
            float dv[K], mt[K];
            // the band of predecessor p at my columns: sv[k] = value[p][c + k - 1], sg[k] = gapm_val[p][c + k]
            // (loaded at the offset the two bands have to each other; cells outside p's band: unreached)
            auto load_band = [&](uint32_t p, float (&sv)[K + 1], float (&sg)[K], int &delta) {
                const float *prow = rows + (size_t)p * kRow;
                delta = c - (int)(header(p).x & 0xffffu);  // my cell k is the predecessor's cell k + delta
                delta = delta < -(K + 1) ? -(K + 1) : (delta > K + 1 ? K + 1 : delta);
                const float *pv = prow + 4 + delta - 1;
                const float *pg = prow + 4 + K + delta;
#pragma unroll
                for (int i = 0; i < K / 4; i++) {
                    const float4_u a = *reinterpret_cast<const float4_u *>(pv + 4 * i);
                    const float4_u b = *reinterpret_cast<const float4_u *>(pg + 4 * i);
                    sv[4 * i] = a.x, sv[4 * i + 1] = a.y, sv[4 * i + 2] = a.z, sv[4 * i + 3] = a.w;
                    sg[4 * i] = b.x, sg[4 * i + 1] = b.y, sg[4 * i + 2] = b.z, sg[4 * i + 3] = b.w;
                }
                sv[K] = pv[K];
            };
            auto relax = [&](float (&sv)[K + 1], float (&sg)[K], int delta, bool first) {
#pragma unroll
                for (int k = 0; k <= K; k++) sv[k] = (uint32_t)(k - 1 + delta) < (uint32_t)K ? sv[k] : kScoutDead;
#pragma unroll
                for (int k = 0; k < K; k++) sg[k] = (uint32_t)(k + delta) < (uint32_t)K ? sg[k] : kScoutDead;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const float v = sv[k + 1] + gp;  // deletion (mesh.h:307-330): open ...
                    const float g = sg[k] + gpe;     // ... or extend the predecessor's gap
                    const float cand = v < g ? v : g;
                    gm[k] = cand;                    // (the LAST predecessor defines gapm)
                    float mv = sv[k] + csel[k];      // match from (p, s-1) (:360-374)
                    if (k == 0 && col0) mv = __builtin_inff();
                    if (first) {
                        dv[k] = (k == 0 && col0) ? (cand < 1.0f ? cand : 1.0f) : cand;
                        mt[k] = mv;
                    } else {
                        dv[k] = cand < dv[k] ? cand : dv[k];
                        mt[k] = mv < mt[k] ? mv : mt[k];
                    }
                }
            };
            // (two predecessors' bands asked for at a time: a row is a wave's LONGEST predecessor list -- 64 queries --
            // of round trips to memory if they go one by one)
            for (uint32_t e = 0; e < npred; e += 2) {
                float sv0[K + 1], sg0[K], sv1[K + 1], sg1[K];
                int d0 = 0, d1 = 0;
                const bool two = e + 1 < npred;
                load_band(pred_id(e), sv0, sg0, d0);
                if (two) load_band(pred_id(e + 1), sv1, sg1, d1);
                relax(sv0, sg0, d0, e == 0);
                if (two) relax(sv1, sg1, d1, false);
            }
#pragma unroll
            for (int k = 0; k < K; k++) loc[k] = mt[k] < dv[k] ? mt[k] : dv[k];
        }
        // (the next row's predecessor entries: asked for now, needed a row from here)
        const uint4 pe4_next = *reinterpret_cast<const uint4 *>(pred + (r_next.x & ~3u));
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
        // ---- publish: the row for its successors, its minimum for their bands, the end-cell candidates
        // (the row's "minimum": by value plus a credit per query base consumed -- the cells of a row have consumed
        // different numbers; the credit is the smallest gain a column offers, so a matching step still lowers it)
        float vmin = __builtin_inff(), vcred = __builtin_inff();
        int amin = c;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool real = (uint32_t)c + (uint32_t)k < L;
            fv[k] = real ? fv[k] : kScoutDead;
            gm[k] = real ? gm[k] : kScoutDead;
            vmin = fv[k] < vmin ? fv[k] : vmin;
            const float cr = fv[k] + credit * (float)(c + k);
            if (cr < vcred) {
                vcred = cr;
                amin = c + k;
                last_v = fv[k];
            }
            if ((uint32_t)c + (uint32_t)k == L - 1u) best = fv[k] < best ? fv[k] : best;  // rows x the last column (mesh.h:569-575)
        }
        if (r.z & kRecSink) best = vmin < best ? vmin : best;  // sink rows x every column (:577-592)
        last_a = amin;
        ring[m % kScoutRing][lane] = uint2{(uint32_t)c | ((uint32_t)amin << 16), __float_as_uint(last_v)};
        float *mine = rows + (size_t)m * kRow;
        {
            // (a row's successors are rows of this same lane: its own stores and loads, which the memory pipeline keeps
            // in order -- no wait, no fence)
            float4_u h;
            h.x = __int_as_float(c);
            h.y = __int_as_float(amin);
            h.z = last_v;
            h.w = 0.f;
            *reinterpret_cast<float4_u *>(mine) = h;
#pragma unroll
            for (int i = 0; i < K / 4; i++) {
                float4_u a, b;
                a.x = fv[4 * i], a.y = fv[4 * i + 1], a.z = fv[4 * i + 2], a.w = fv[4 * i + 3];
                b.x = gm[4 * i], b.y = gm[4 * i + 1], b.z = gm[4 * i + 2], b.w = gm[4 * i + 3];
                *reinterpret_cast<float4_u *>(mine + 4 + 4 * i) = a;
                *reinterpret_cast<float4_u *>(mine + 4 + K + 4 * i) = b;
            }
        }
        r = r_next;
        pe4 = pe4_next;
    }
    out_u[(size_t)variant * nq + qi] = best;
}

}  // namespace

size_t scout_scratch_floats(uint64_t tb_rows) { return (size_t)(2 * tb_rows + 2) * (4 + 2 * kScoutBand) + 2 * (kScoutBand + 8); }

// out_u: [2][nq] (the two sweeps' values)
int launch_mesh_scout(const DpArgs &a, uint32_t nq, uint32_t Lp, uint64_t tb_rows, float *scratch, float *out_u, hipStream_t s) {
    const uint32_t blocks = 2u * ((nq + 63u) / 64u);
    hipLaunchKernelGGL((mesh_scout_kernel<kScoutBand>), dim3(blocks), dim3(64), 0, s, a.qd, a.order, a.rec, a.pred, a.qmask,
                       scratch, Lp, nq, a.ms, a.mms, a.gp, a.gpe, out_u, tb_rows);
    SH_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sina_hip
