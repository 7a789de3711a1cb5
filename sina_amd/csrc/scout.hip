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

template <int K>
__global__ void __launch_bounds__(64) mesh_scout_kernel(const QDesc *__restrict__ qdv, const uint32_t *__restrict__ orderv,
                                                        const uint4 *__restrict__ recv, const uint32_t *__restrict__ predv,
                                                        const uint8_t *__restrict__ qmaskv, float *__restrict__ scratch,
                                                        uint32_t Lp, uint32_t nq, float ms, float mms, float gp, float gpe,
                                                        float *__restrict__ out_u) {
    static_assert(K % 4 == 0 && K >= 8 && K <= 32, "band width");
    constexpr int kRow = 4 + 2 * K;  // floats per stored row
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
    if (slot >= nq) return;
    const uint32_t qi = orderv[slot];  // (longest first: the lanes of a wave run about equally long)
    const QDesc d = qdv[qi];
    const uint32_t N = d.N, L = d.L;
    const uint4 *__restrict__ rec = recv + d.node_off;
    const uint32_t *__restrict__ pred = predv + d.edge_off;
    const uint8_t *__restrict__ qm = qmaskv + d.q_off;
    // (rows of this query: the launch's trace-back rows are numbered the same way, tb_off = rows before it * Lp;
    // K + 8 floats of slack in front of the array: a band read at an offset may start before its row)
    float *__restrict__ rows = scratch + (K + 8) + (size_t)(d.tb_off / Lp) * kRow;
    const int c_max = L > (uint32_t)K ? (int)(L - (uint32_t)K) : 0;

    float best = __builtin_inff();
    int c_prev = 0;
    for (uint32_t m = 0; m < N; ++m) {
        const uint4 r = rec[m];
        const uint32_t npred = r.z & 0xffu, mmask = (r.z >> 8) & 0xfu;
        const float wgt = __uint_as_float(r.y);
        const float vM = ms * wgt, vX = mms * wgt;  // scoring_schemes.h:154
        // ---- where the band goes: a diagonal step behind the minimum of the best predecessor row
        int c = c_prev;
        if (npred != 0) {
            float bv = __builtin_inff();
            int ba = 0;
            for (uint32_t e = 0; e < npred; ++e) {
                const uint32_t p = pred[r.x + e] & 0xffffu;
                const float4_u h = *reinterpret_cast<const float4_u *>(rows + (size_t)p * kRow);
                if (h.z < bv) {
                    bv = h.z;
                    ba = __float_as_int(h.y);
                }
            }
            c = ba + 1 - K / 2;
        }
        c = c < 0 ? 0 : (c > c_max ? c_max : c);
        c_prev = c;
        // ---- match / mismatch score of my K columns against this row (comp(): aligned_base.h:153)
        float csel[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t s = (uint32_t)c + (uint32_t)k;
            const uint32_t q = s < L ? (uint32_t)(qm[s] & 0xf) : 0u;
            csel[k] = (q & mmask) != 0u ? vM : vX;
        }
        const bool col0 = c == 0;  // my cell 0 is query column 0: initial value 1, no match step, no insertion
        float loc[K], gm[K];
        if (npred == 0) {  // a source row: every cell starts at 1 (init_edge) and stays untouched
#pragma unroll
            for (int k = 0; k < K; k++) loc[k] = gm[k] = 1.0f;
        } else {
            float dv[K], mt[K];
            for (uint32_t e = 0; e < npred; ++e) {
                const uint32_t p = pred[r.x + e] & 0xffffu;
                const float *__restrict__ prow = rows + (size_t)p * kRow;
                int delta = c - __float_as_int(prow[0]);  // my cell k is the predecessor's cell k + delta
                delta = delta < -(K + 1) ? -(K + 1) : (delta > K + 1 ? K + 1 : delta);
                float sv[K + 1], sg[K];  // sv[k] = value[p][c + k - 1], sg[k] = gapm_val[p][c + k]
                {
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
                }
                // (cells outside the predecessor's band: unreached)
#pragma unroll
                for (int k = 0; k <= K; k++) sv[k] = (uint32_t)(k - 1 + delta) < (uint32_t)K ? sv[k] : kScoutDead;
#pragma unroll
                for (int k = 0; k < K; k++) sg[k] = (uint32_t)(k + delta) < (uint32_t)K ? sg[k] : kScoutDead;
                const bool first = e == 0;
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
            }
#pragma unroll
            for (int k = 0; k < K; k++) loc[k] = mt[k] < dv[k] ? mt[k] : dv[k];
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
        // ---- publish: the row for its successors, its minimum for their bands, the end-cell candidates
        float vmin = __builtin_inff();
        int amin = c;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool real = (uint32_t)c + (uint32_t)k < L;
            fv[k] = real ? fv[k] : kScoutDead;
            gm[k] = real ? gm[k] : kScoutDead;
            if (fv[k] < vmin) {
                vmin = fv[k];
                amin = c + k;
            }
            if ((uint32_t)c + (uint32_t)k == L - 1u) best = fv[k] < best ? fv[k] : best;  // rows x the last column (mesh.h:569-575)
        }
        if (r.z & kRecSink) best = vmin < best ? vmin : best;  // sink rows x every column (:577-592)
        float *__restrict__ mine = rows + (size_t)m * kRow;
        {
            float4_u h;
            h.x = __int_as_float(c);
            h.y = __int_as_float(amin);
            h.z = vmin;
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
        // (my successors are rows of this lane: its own stores, in program order -- but they come back through the
        // vector cache by another instruction: wait for them to have left)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    out_u[qi] = best;
}

}  // namespace

size_t scout_scratch_floats(uint64_t tb_rows) { return (size_t)(tb_rows + 2) * (4 + 2 * kScoutBand) + 2 * (kScoutBand + 8); }

int launch_mesh_scout(const DpArgs &a, uint32_t nq, uint32_t Lp, float *scratch, float *out_u, hipStream_t s) {
    const uint32_t blocks = (nq + 63u) / 64u;
    hipLaunchKernelGGL((mesh_scout_kernel<kScoutBand>), dim3(blocks), dim3(64), 0, s, a.qd, a.order, a.rec, a.pred, a.qmask,
                       scratch, Lp, nq, a.ms, a.mms, a.gp, a.gpe, out_u);
    SH_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sina_hip
