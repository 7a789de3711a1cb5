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
//   * one WAVE per query (a 64-thread workgroup); the query's columns are cut into strips
//     of 64*B columns, lane l owns B consecutive columns of the current strip, and the wave
//     sweeps all DAG rows for one strip before it starts the next (left to right);
//   * the rows are swept in topological (id) order: the row record {first pred, weight,
//     #pred|mask|flags, spill slot} and the pred list are wave-uniform and come through
//     the scalar cache (s_load), so there is no divergence on graph structure and no
//     vector-memory latency on the row critical path;
//   * finished rows {value, gapm_val} of a wave's own columns are kept for their
//     successors in W LDS slots (8 B per column) that only that wave touches; slots
//     are handed out by liveness when the graph is built, and a row that finds every
//     slot busy goes to a per-query spill area in HBM instead (read back by the lane
//     that wrote it; 3 % of the predecessor reads at three slots);
//   * gapm_idx is not carried at all: the trace-back cell records whether a deletion
//     extends its predecessor's gap and whether the row's own gap was opened from its
//     last predecessor, and backtrack() resolves the index for the few cells on the
//     final path (common.h, kTbExt / kTbOpLast);
//   * the only dependencies between column blocks are (a) the insertion chain
//     along the query (cell (m,s) needs the final (m,s-1)) and (b) the match
//     candidate from (p,s-1).  Inside a wave both travel by lane shuffle (DPP).  For
//     the chain each lane first finds its exit state as if no gap entered from the
//     left; then the exit states are propagated lane to lane -- a gap either runs
//     through all B cells of a lane or dies inside it, B adds + B compares per step
//     -- until none changes (wave vote, no barrier), and one full pass over the B
//     cells with the converged left states finishes the row (weighted / forbid
//     schemes: full chain passes are iterated instead).  Between strips they travel through a
//     per-row edge record in global memory {value of the strip's last column, exit state}: lane 63
//     writes it, and the next strip -- the same wave, a whole sweep later -- reads it back through
//     the scalar cache together with the row record.  There is no other wave to wait for: the
//     kernel has no barrier, no flag and no spin loop at all;
//   * the only per-cell HBM traffic is the write-once trace-back cell (2 bytes; 4 with
//     --insertion=forbid), row-major.
//
// Two kernels share this design: mesh_dp_kernel<B, WEIGHTED, FORBID, BELOW_INIT> is the general one;
// mesh_dp_simple_kernel<B> further down is the same recurrence written for the default scheme (what every
// BASELINE configuration runs) with a third fewer instructions per row.  assemble_kernel finishes the
// alignments behind backtrack_kernel (container steps + NAST fix-up) where that is plain.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace sina_hip {

namespace {

struct ChainState {  // canonical exit state of cell (m, s): what cell (m, s+1) can observe
    float v;         // final value
    uint32_t e;      // gaps_val == value  (the "extend" condition, mesh.h:340)
    uint32_t gsi;    // gaps_idx (only meaningful when e)
    uint32_t gmax;   // gaps_max (FORBID only, only meaningful when e)
};

// (GSI: the variant reads gaps_idx / gaps_max -- weighted gap costs, --insertion=forbid; the
// simple scheme never looks at them)
template <bool GSI>
__device__ __forceinline__ bool same_state(const ChainState &a, const ChainState &b) {
    if constexpr (GSI) return a.v == b.v && a.e == b.e && (!a.e || (a.gsi == b.gsi && a.gmax == b.gmax));
    else return a.v == b.v && a.e == b.e;
}

// value of `x` in lane-1 (lane 0 gets an unspecified value): one DPP move, no LDS round trip
__device__ __forceinline__ uint32_t lane_shr1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ float lane_shr1(float x) { return __uint_as_float(lane_shr1(__float_as_uint(x))); }

// Bare v_min_f32 / v_add_f32 as the compiler's own instructions.  This file is built with
// -fno-honor-nans -mno-amdgpu-ieee (no operand canonicalisation `v_max x, x` in front of a minimum:
// the values here are never NaN; infinities ARE used and stay honoured) and -fno-slp-vectorize (the
// SLP vectoriser otherwise pairs unrelated adds into v_pk_add_f32 and pays two register moves per
// pair to line the operands up).  Until round 3 these three were inline asm, which the hazard
// recogniser cannot see through: every v_cmp -> v_cndmask pair with such an asm in between was
// padded with an s_nop, and asm operands forced SGPR -> VGPR copies.
__device__ __forceinline__ float min2_raw(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float min3_raw(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float add_raw(float a, float b) { return a + b; }
// keeps a wave-uniform float in a VGPR as an opaque value (stops the compiler from re-deriving it
// per use from its scalar inputs with vector multiplies)
__device__ __forceinline__ float opaque_v(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

// B consecutive floats as B/4 16-byte vectors: what a row read delivers, used in place (element
// access with a compile-time index is a sub-register, no copy)
template <int B>
struct Cells {
    using V = __attribute__((ext_vector_type(4))) float;
    V v[B / 4];
    __device__ __forceinline__ float operator[](int k) const { return v[k >> 2][k & 3]; }
    __device__ __forceinline__ void load(const float *__restrict__ src) {
#pragma unroll
        for (int i = 0; i < B / 4; i++) v[i] = *reinterpret_cast<const V *>(src + 4 * i);
    }
    // LDS row slot layout: the i-th group of four cells of all 64 lanes is contiguous ([B/4][64][4]
    // floats), so that every 16-byte access of a wave covers 1 KiB without a gap -- lane-major rows
    // of B floats put lanes l and l + 8 on the same banks for B = 8 (2-way conflict on every access)
    __device__ __forceinline__ void load_slot(const float *__restrict__ arr, int lane) {
#pragma unroll
        for (int i = 0; i < B / 4; i++) v[i] = *reinterpret_cast<const V *>(arr + (i * 64 + lane) * 4);
    }
};
template <int B>
__device__ __forceinline__ void store_slot(float *__restrict__ arr, int lane, const float (&src)[B]) {
    using V = __attribute__((ext_vector_type(4))) float;
#pragma unroll
    for (int i = 0; i < B / 4; i++) {
        V v;
        v.x = src[4 * i];
        v.y = src[4 * i + 1];
        v.z = src[4 * i + 2];
        v.w = src[4 * i + 3];
        *reinterpret_cast<V *>(arr + (i * 64 + lane) * 4) = v;
    }
}

// B consecutive 4-byte cells to a 16-byte aligned address (B % 4 == 0) or 8-byte aligned one
// (B % 2 == 0): wide stores, both for LDS and for global memory
template <int B, typename T>
__device__ __forceinline__ void store_cells(T *__restrict__ dst, const T (&src)[B]) {
    static_assert(sizeof(T) == 4 && B % 2 == 0, "4-byte cells, even count");
    if constexpr (B % 4 == 0) {
        using V = __attribute__((ext_vector_type(4))) T;
#pragma unroll
        for (int k = 0; k < B; k += 4) {
            V v;
            v.x = src[k];
            v.y = src[k + 1];
            v.z = src[k + 2];
            v.w = src[k + 3];
            *reinterpret_cast<V *>(dst + k) = v;
        }
    } else {
        using V = __attribute__((ext_vector_type(2))) T;
#pragma unroll
        for (int k = 0; k < B; k += 2) {
            V v;
            v.x = src[k];
            v.y = src[k + 1];
            *reinterpret_cast<V *>(dst + k) = v;
        }
    }
}

#ifdef SINA_DP_PROFILE
// Profiling builds only (make PROFILE=1: per-phase s_memtime totals summed over all waves + ablation
// switches; make PROFILE=2: the ablation switches alone, without the timers' own cost).
__device__ unsigned long long g_dp_prof[32];
__device__ unsigned long long g_dp_span[2 * 16384];  // start / end (100 MHz wall clock) of every query's wave: the launch's drain
__device__ int g_dp_abl;  // ablation mask (timing experiments only, results are WRONG when set)
#define SH_ABL(bit) (abl_ & (bit))
#if SINA_DP_PROFILE == 1
#define SH_PROF_DECL unsigned long long pa_[32] = {}; unsigned long long pt_ = __builtin_amdgcn_s_memtime();
#define SH_PROF(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pa_[i] += t_ - pt_; pt_ = t_; }
#define SH_PROF_CNT(i, n) pa_[i] += (n);
#define SH_PROF_FLUSH if (lane == 0) { for (int i_ = 0; i_ < 32; i_++) atomicAdd(&g_dp_prof[i_], pa_[i_]); }
#define SH_PROF_TIMERS 1
#else
#define SH_PROF_DECL
#define SH_PROF(i)
#define SH_PROF_CNT(i, n)
#define SH_PROF_FLUSH
#endif
#else
#define SH_ABL(bit) false
#define SH_PROF_DECL
#define SH_PROF(i)
#define SH_PROF_CNT(i, n)
#define SH_PROF_FLUSH
#endif

// The insertion chain's log-step guess (mesh_dp_simple_kernel, phase 2) is made after this many plain iterations have
// failed to settle the row's exit states.  Of the rows whose first iteration changes something (0.53 per swept row), a
// third are settled by the second (a gap that runs one lane further and dies there); guessing only behind it: 0.53 ->
// 0.37 guesses per row for 0.21 more iterations (a third of a guess's price each), and a guess made from two steps'
// states is almost never off -- rows that needed four and more iterations fell from 0.12 to 0.01 per row.
#ifndef SINA_DP_SCAN_AT
#define SINA_DP_SCAN_AT 1
#endif
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
// 16 bytes through the scalar cache from a wave-uniform address (see uniform()).  A load from the
// CONSTANT address space, so that the compiler emits s_load_dwordx4 and keeps track of it being in
// flight: it waits (lgkmcnt) before the first use -- including a register spill.  (An inline-asm
// s_load looks finished to the compiler when the asm statement ends: the B = 12 --insertion=forbid
// kernels, short of SGPRs, spilled the destination registers right behind it and restored garbage;
// tools/check_inflight_spills.py looks for that pattern.)  The memory is not constant -- edge
// records are written earlier by this very wave -- but the wave has waited for its stores
// (vmcnt(0) at the end of a strip) and nothing else writes them.
__device__ __forceinline__ u32x4 sload16(uint64_t addr) {
    // (predecessor entries are 4-byte aligned: the load's type says so -- s_load_dwordx4 needs no more)
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef const __attribute__((address_space(4))) u32x4_a4 *cptr;
    return *reinterpret_cast<cptr>(addr);
}
// the point where the loaded value must have arrived (an empty asm that reads it: the compiler puts
// its s_waitcnt here and not behind LDS traffic further down)
__device__ __forceinline__ void sload_wait(u32x4 &v) { asm volatile("" : "+s"(v) : : "memory"); }
__device__ __forceinline__ uint32_t uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t uniform(uint64_t x) {
    return (uint64_t)uniform((uint32_t)x) | ((uint64_t)uniform((uint32_t)(x >> 32)) << 32);
}

// c ? a : b as a v_cndmask (2-cycle class) even where the condition is the operands' own compare, which
// the compiler would turn into v_min_f32 (4-cycle class, tools/ubench/valu_rate.hip): the kernel needs
// the compare for the trace-back tags anyway.  (An integer select; the empty asm hides the pairing.)
__device__ __forceinline__ float sel(bool c, float a, float b) {
    uint32_t bu = __float_as_uint(b);
    asm("" : "+v"(bu));
    return __uint_as_float(c ? __float_as_uint(a) : bu);
}

// (the same with the empty asm on `a`: for a `b` that has further uses -- hiding that one costs a register copy)
__device__ __forceinline__ float sel_a(bool c, float a, float b) {
    uint32_t au = __float_as_uint(a);
    asm("" : "+v"(au));
    return __uint_as_float(c ? au : __float_as_uint(b));
}

// "is the predicate true in any lane": the wave mask itself (HIP's __any() goes through a 0 / 1 VGPR and
// a second compare)
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// Issue priority of the wave by how far it is through its query (called every 128 rows): 3 for the first three
// eighths, then 2 2 1 1 and 0 for the last eighth.  The scheduler otherwise favours a SIMD's oldest wave, the
// three waves of a SIMD end one after the other, and a launch whose queue has run dry empties SIMD by SIMD --
// 8 ms before the end of a 9216-query launch a third of the wave slots were idle, 4 ms before it two thirds
// (tools/perf_dp.py on the profiling build).  A wave that is behind catches up instead, the SIMD's waves end
// together: 3072 / 9216 queries in 21.8 / 58.1 instead of 23.4 / 61.4 ms (597 -> 641, 683 -> 721 Gcell/s).
// Results do not depend on it.  (Tried: quarters 3 2 1 0 -- 623 / 710; the reverse -- no change; high priority
// for the first eighths only -- 607 / 688; updates every 32 or 512 rows -- the same; a priority that merely
// ROTATES with the row count, unrelated to progress -- 655 / 703-722, i.e. as good: what the scheduler's default
// lacks is waves of a SIMD taking turns at being preferred, and the steady state gains as much as the drain
// (19.0 -> 18.2 ms per round of 3072 queries); a STATIC priority by hardware wave slot -- 597 / 656, worse
// than none.  Round 4, chained launches: a floor of 1 for the last eighth, above the waves of the kernel that fills the
// drain -- 142.1 k sequences/s against 142.1 k: nothing.)
__device__ __forceinline__ void issue_priority_by_progress(uint32_t rows_done, uint32_t rows_total) {
#ifndef SINA_DP_NO_PRIO
    const uint32_t r8 = (8u * rows_done) / rows_total;
    if (r8 < 3u) __builtin_amdgcn_s_setprio(3);
    else if (r8 < 5u) __builtin_amdgcn_s_setprio(2);
    else if (r8 < 7u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#else
    (void)rows_done;
    (void)rows_total;
#endif
}

template <int B, bool WEIGHTED, bool FORBID, bool BELOW_INIT, bool DBG>
__global__ void __launch_bounds__(64, (B <= 4 ? 4 : (B <= 8 ? 3 : 2)))
mesh_dp_kernel(const QDesc *__restrict__ qdv, const uint32_t *__restrict__ orderv, const uint4 *__restrict__ recv, const uint32_t *__restrict__ predv,
               const uint32_t *__restrict__ node_posv, const uint32_t *__restrict__ succ_minposv,
               const uint8_t *__restrict__ qmaskv, const float *__restrict__ weights, uint32_t n_weights,
               void *__restrict__ tbv, float *__restrict__ dbg_value, float *spillv, EdgeRec *edgev,
               uint64_t edge_stride, uint32_t n_strips, DpResult *__restrict__ resv, float ms, float mms, float gp,
               float gpe, const float *__restrict__ prof16v, DryArgs dry) {
    static_assert(B % 4 == 0, "16-byte accesses per array");
    constexpr int kStrip = 64 * B;  // columns per strip
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x;
    dry_signal(dry, gridDim.x, lane == 0);  // (the launch queued behind this one may start once the last of us has: ctx.h)
    // workgroups are dispatched in blockIdx order: the launch lists the queries by decreasing work
    // (longest first), so that a launch of more workgroups than the GPU holds at once ends evenly
    const uint32_t qi = orderv[blockIdx.x];
    const QDesc d = qdv[qi];
    const uint32_t N = uniform(d.N), L = uniform(d.L);
    const uint32_t Lp = n_strips * (uint32_t)kStrip;  // row stride of the trace-back plane and of the spill rows
    // strips this query needs: those up to the one that holds its last column (a launch's geometry
    // follows its longest query; a short one beside it is done after its own strips -- the columns
    // beyond L are read by nothing)
    const uint32_t S = min(n_strips, (L - 1) / (uint32_t)kStrip + 1);
    const uint64_t node_off = uniform(d.node_off);
    const uint64_t erec_off = uniform(d.erec_off);  // (a multiple of 4 records: 64-byte lines of this query's own)

    // ---- LDS: W row slots, each value[kStrip] | gapm_val[kStrip] | value of the column left of the strip
    unsigned char *ring = smem;
    constexpr size_t kValBytes = (size_t)kStrip * 4;
    constexpr size_t kSlotBytes = 2 * kValBytes + 16;

    const uint4 *__restrict__ rec = recv + node_off;
    const uint32_t *__restrict__ pred = predv + uniform(d.edge_off);
    const uint32_t *__restrict__ node_pos = node_posv + node_off;
    const uint32_t *__restrict__ succ_minpos = succ_minposv + node_off;
    // (16-bit trace-back cells except with --insertion=forbid: common.h)
    using TbCell = std::conditional_t<FORBID, uint32_t, uint16_t>;
    TbCell *__restrict__ tb = reinterpret_cast<TbCell *>(tbv) + uniform(d.tb_off);
    float *spill = spillv + uniform(d.spill_off) * (size_t)(2 * Lp);
    const uint64_t q_off = uniform(d.q_off);

    // end-cell search (mesh.h:567-592), accumulated over the strips
    const uint32_t strip_last = (L - 1) / (uint32_t)kStrip;         // the strip that owns column L-1
    const int lane_last = (int)(((L - 1) / B) & 63u);               // ... and the lane
    const int k_last = (int)((L - 1) % B);
    const int kg_last = k_last >> 2, kr_last = k_last & 3;
    const uint32_t kr_is1 = kr_last == 1 ? ~0u : 0u, kr_is2 = kr_last == 2 ? ~0u : 0u,
                   kr_is3 = kr_last == 3 ? ~0u : 0u;  // select masks
    float lc_min = 0.f, lc_snk0 = 0.f;  // step 1: rows at column L-1 (lane lane_last of strip strip_last only)
    uint32_t lc_arg = 0;
    bool lc_any = false;
    float all_min = __builtin_inff();   // step 2 over the strips done so far: sink rows x columns
    uint32_t all_m = 0, all_s = 0xffffffffu, snk0 = 0;
    bool all_any = false;

    SH_PROF_DECL
    // kLazy: trace-back cells carry a type code, value_sidx is derived in backtrack (common.h);
    // kGsi: gaps_idx is read by the recurrence itself (gap costs by length, --insertion=forbid)
    constexpr bool kLazy = !FORBID;
    constexpr bool kGsi = WEIGHTED || FORBID;
    constexpr uint32_t kTagNone = kLazy ? kTbNone : 0u;
    constexpr uint32_t kTagOpLast = kTbOpLast;  // (32-bit cells; 16-bit cells carry the inverse, kTb16XLast)
    // BELOW_INIT (chosen by the host per launch, dp_below_init()): no value of this launch can reach
    // the 1e6 initial value of rows with predecessors, so their first deletion candidate always
    // replaces it and needs no compare.
#ifdef SINA_DP_PROFILE
    const int abl_ = g_dp_abl;
#endif

    // Per-row scalars (wave-uniform, from the row record through the scalar cache).
    struct Row {
        uint32_t pb, npred, mmask, keep, z, mpos, smax;
        float cM, cX, gd_open, gd_ext, gi_open, init_v;
    };
    auto setup_row = [&](const uint4 r, uint32_t m) {
        Row o;
        o.pb = r.x;
        const float wgt = __uint_as_float(r.y);
        o.z = r.z;
        o.npred = r.z & 0xffu;
        o.mmask = (r.z >> 8) & 0xfu;
        o.keep = r.w;
        o.mpos = 0;
        if constexpr (WEIGHTED) {
            o.mpos = node_pos[m];
            const uint32_t nw1 = n_weights - 1;
            const float wp = weights[o.mpos < nw1 ? o.mpos : nw1];
            const float wp1 = weights[o.mpos + 1 < nw1 ? o.mpos + 1 : nw1];
            o.cM = ms * wp * wgt;  // (c * weights[pos]) * weight, scoring_schemes.h:232
            o.cX = mms * wp * wgt;
            o.gd_open = gp * wp;   // :211
            o.gd_ext = gpe * wp;   // :222
            o.gi_open = gp * wp1;  // :187
        } else {
            o.cM = ms * wgt;       // scoring_schemes.h:154
            o.cX = mms * wgt;
            o.gd_open = gp;
            o.gd_ext = gpe;
            o.gi_open = gp;
        }
        o.smax = 0;
        if constexpr (FORBID) {
            if (!WEIGHTED) o.mpos = node_pos[m];
            // int max_insert = min_mpos - pos - 1, passed as unsigned idx_type (mesh.h:480-489)
            o.smax = (uint32_t)(int)(succ_minpos[m] - o.mpos - 1);
        }
        o.init_v = (o.npred == 0) ? 1.0f : 1000000.0f;  // edge rows start at 1 (mesh.h init_edge)
        return o;
    };

    for (uint32_t strip = 0; strip < S; ++strip) {
    const uint32_t s0 = (strip * 64u + (uint32_t)lane) * B;  // my first column
    const bool col0_mine = (strip == 0) && (lane == 0);      // my cell 0 is query column 0
    const bool has_col_left = !col0_mine;                     // something is to the left of my cells
    // edge records: what the strip to my left left behind per row (read), what I leave behind (written)
    const uint64_t e_in = uniform((uint64_t)(edgev + (size_t)(strip ? strip - 1 : 0) * edge_stride + erec_off));
    EdgeRec *e_out = edgev + (size_t)strip * edge_stride + erec_off;
    const bool have_left_strip = strip > 0, have_right_strip = strip + 1 < S;

    // query masks of my columns (0 beyond L: never matches, never stored)
    uint32_t qm[B];
#pragma unroll
    for (int k = 0; k < B; k++) {
        const uint32_t s = s0 + k;
        qm[k] = (s < L) ? (uint32_t)(qmaskv[q_off + s] & 0xf) : 0u;
    }
    const bool own_last = (strip == strip_last) && (lane == lane_last);
    float sk_min = __builtin_inff();  // step 2: sink rows x this strip's columns, first in scan order
    uint32_t sk_m = 0, sk_s = 0xffffffffu;
    bool sk_any = false;

    // Everything a row reads through the scalar cache is requested a row ahead and waited for at a
    // point of the previous row where no LDS access is outstanding (SMEM and LDS share one counter,
    // and SMEM returns out of order: waiting for a scalar load means waiting for everything): the row
    // record, the first four predecessor entries, the edge record.
    const uint64_t pred_addr = uniform((uint64_t)pred);
    uint4 cur = rec[0];
    u32x4 cur_pe = sload16(pred_addr + (uint64_t)cur.x * 4);
    u32x4 cur_edge = {0, 0, 0, 0};
    if (have_left_strip) cur_edge = sload16(e_in);
    sload_wait(cur_pe);
    sload_wait(cur_edge);
    // The row just finished stays in registers: it is the last predecessor of more than half of the
    // rows, and a row whose ONLY successor is the next row is kept nowhere else (common.h).
    Cells<B> prev_v, prev_g;
    float prev_edge_val = 0.f;
#pragma unroll
    for (int i = 0; i < B / 4; i++) prev_v.v[i] = prev_g.v[i] = typename Cells<B>::V{0.f, 0.f, 0.f, 0.f};
    for (uint32_t m = 0; m < N; ++m) {
        if ((m & 127u) == 0) issue_priority_by_progress(strip * N + m, S * N);
        const uint32_t m_next = m + 1 < N ? m + 1 : m;
        const uint4 nrec = rec[m_next];  // scalar prefetch of the next row record ...
        u32x4 nedge = {0, 0, 0, 0};
        if (have_left_strip) nedge = sload16(e_in + (uint64_t)m_next * sizeof(EdgeRec));  // ... and of its edge record
        // phase-1 results of the current row: deletion / match candidates of my B cells
        // (dvm / mtp hold value_midx already shifted into its trace-back position, dvm with kTbExt)
        float dv[B], gm[B], mt[B];
        uint32_t dvm[B], dvs[B], mtp[B];
        bool oplast[B];  // gapm of cell k was opened from the row's last predecessor (lane masks in SGPRs)
        auto init_cells = [&](const Row &r) {
#pragma unroll
            for (int k = 0; k < B; k++) {
                const float iv = (k == 0 && col0_mine) ? 1.0f : r.init_v;
                dv[k] = iv;
                gm[k] = iv;
                mt[k] = __builtin_inff();
                dvm[k] = kTagNone;
                dvs[k] = 0;
                mtp[k] = 0;
                oplast[k] = false;
            }
        };

        // ---- phase 1: deletion / match candidates of my B cells from the predecessor rows
        const Row r = setup_row(cur, m);
        // what is left of my strip in this row: value of the last column there, exit state of its chain
        const float edge_val = __uint_as_float(cur_edge.x);
        SH_PROF(1)
        if (r.npred == 0) init_cells(r);  // (otherwise the first predecessor's relax initialises)
        float gdo_v, gde_v;  // gap open / extend cost of a deletion, in VGPRs
        float csel[B];
        {
            const float vM = opaque_v(r.cM), vX = opaque_v(r.cX), vgo = opaque_v(r.gd_open), vge = opaque_v(r.gd_ext);
            gdo_v = vgo;
            gde_v = vge;
#pragma unroll
            for (int k = 0; k < B; k++) csel[k] = (r.mmask & qm[k]) ? vM : vX;  // comp(): optimistic IUPAC match (aligned_base.h:153)
            if (prof16v != nullptr) {
                // --fs-no-graph: the row is a profile column, its match term a table over the query's base
                // (scoring_scheme_profile::match, scoring_schemes.h:84-92; the table is the host's, exact)
                const float *tab = prof16v + 16 * (size_t)(node_off + m);
#pragma unroll
                for (int k = 0; k < B; k++) csel[k] = tab[qm[k]];
            }
        }
        // (FIRST: the row's first predecessor meets the initial values -- constants -- instead of
        // registers that would have to be initialised first)
        const float iv0 = col0_mine ? 1.0f : r.init_v;  // initial value of my cell 0 (column 0 starts at 1)
        auto relax = [&](auto first_tag, uint32_t p, const Cells<B> &sv, const Cells<B> &sg, float svl) {
            // first_tag: 0 = a later predecessor, 1 = the first one, 2 = the first one in a launch whose
            // values provably stay below the 1e6 initial value (below_init): every deletion candidate
            // then beats the initial value and needs no compare (column 0, initial value 1, excepted)
            constexpr bool FIRST = decltype(first_tag)::value != 0;
            constexpr bool BELOW = decltype(first_tag)::value == 2;
            // trace-back tags of this predecessor: its ordinal in the row's list (16-bit cells) or its
            // row id (32-bit cells), plus type / Ext bits (type code kTbDel == 0)
            const uint32_t p_open = kLazy ? (p << kTb16OrdShift) : (p << 16);
            const uint32_t p_ext = p_open | (kLazy ? kTb16Ext : kTbExt);
            const uint32_t p_match = kLazy ? (p_open | kTbMatch) : p_open;
#pragma unroll
            for (int k = 0; k < B; k++) {
                // deletion (mesh.h:307-330): gapm_* is overwritten by every predecessor
                const float v = add_raw(sv[k], gdo_v);
                const float g = add_raw(sg[k], gde_v);
                // (values by v_min -- the same number as the reference's compare-and-assign, there are
                // no NaNs -- so that only the trace-back tags wait for the compares)
                const bool op = v < g;
                const float cand = min2_raw(v, g);
                gm[k] = cand;
                oplast[k] = op;  // (every predecessor overwrites: the last one stays)
                const float dv_old = FIRST ? (k == 0 ? iv0 : r.init_v) : dv[k];
                if (BELOW && k > 0) {  // (cell 0 of lane 0 may be column 0: its initial value is 1)
                    dv[k] = cand;
                    dvm[k] = op ? p_open : p_ext;
                    dvs[k] = s0 + k;
                } else {
                    const bool better = cand < dv_old;
                    dv[k] = min2_raw(cand, dv_old);
                    dvm[k] = better ? (op ? p_open : p_ext) : (FIRST ? kTagNone : dvm[k]);
                    dvs[k] = better ? s0 + k : (FIRST ? 0u : dvs[k]);  // value_sidx of a deletion is the column itself
                }
                // match from (p, s-1) (mesh.h:360-374); first predecessor with the minimum wins
                const float pvv = (k == 0) ? svl : sv[k - 1];
                const float mv = add_raw(pvv, csel[k]);
                if constexpr (FIRST) {
                    // against the initial +inf every candidate wins: values are finite (a cell never
                    // exceeds its finite deletion candidates or initial value), so mv < inf always
                    const bool mb = (k > 0) || has_col_left;  // s > 0
                    mt[k] = mb ? mv : __builtin_inff();
                    mtp[k] = mb ? p_match : 0u;
                } else {
                    const float mt_old = mt[k];
                    const bool mb = ((k > 0) || has_col_left) && (mv < mt_old);
                    mt[k] = (k == 0) ? (mb ? mv : mt_old) : min2_raw(mv, mt_old);
                    mtp[k] = mb ? p_match : mtp[k];
                }
            }
        };
        // Predecessors in ascending id order (the reference's order: the first minimum wins, the
        // last one defines gapm).  Entry = id | (LDS slot or spill row) << 16 | spilled << 31.
        // The single value I need from left of my strip (column s0-1, lane 0 only) sits in the row's
        // LDS slot behind the two arrays, or, for a spill row, in the spill row itself (the previous
        // strip wrote that column).
        for (uint32_t e = 0; e < r.npred; ++e) {
            if (SH_ABL(16) && e > 0) break;
            const uint32_t pe = e == 0 ? cur_pe.x : (e == 1 ? cur_pe.y : (e == 2 ? cur_pe.z : (e == 3 ? cur_pe.w : pred[r.pb + e])));
            const uint32_t p = pe & 0xffffu;
            Cells<B> sv, sg;
            float left_of_strip = 0.f;
            if (p + 1 == m) {  // the previous row: still in registers
                sv = prev_v;
                sg = prev_g;
                left_of_strip = prev_edge_val;
            } else if (pe & kPredSpilled) {
                if (SH_ABL(2)) continue;
                const float *row = spill + (size_t)((pe >> 16) & 0x7FFFu) * (2 * Lp);
                sv.load(row + s0);
                sg.load(row + Lp + s0);
                if (lane == 0 && have_left_strip) left_of_strip = row[s0 - 1];
                // consume the global loads HERE: the compiler then waits for them (vmcnt) inside
                // this rare branch instead of after the merge with the LDS path, where the wait
                // would also drain the previous row's trace-back stores on every row
#pragma unroll
                for (int i = 0; i < B / 4; i++) {
                    asm volatile("" : "+v"(sv.v[i]));
                    asm volatile("" : "+v"(sg.v[i]));
                }
                asm volatile("" : "+v"(left_of_strip));
                SH_PROF_CNT(9, 1)
            } else {
                const unsigned char *slot = ring + (size_t)(pe >> 16) * kSlotBytes;
                sv.load_slot(reinterpret_cast<const float *>(slot), lane);
                sg.load_slot(reinterpret_cast<const float *>(slot + kValBytes), lane);
                if (have_left_strip) left_of_strip = *reinterpret_cast<const float *>(slot + 2 * kValBytes);
            }
            float svl = lane_shr1(sv[B - 1]);  // value[p][s0-1] lives in the lane to my left
            if (lane == 0) svl = left_of_strip;
            const uint32_t p_tag = kLazy ? e : p;
            if (e == 0) {
                relax(std::integral_constant<int, BELOW_INIT ? 2 : 1>{}, p_tag, sv, sg, svl);
            } else {
                relax(std::integral_constant<int, 0>{}, p_tag, sv, sg, svl);
            }
        }
        SH_PROF(3)
        const bool is_sink = (r.z & kRecSink) != 0;
        // (the relaxation has just waited for its last LDS read: the next row's record is here by
        // now -- consume it, so that the compiler's wait for it sits here and not at the top of the
        // next row behind this row's LDS stores -- and ask for that row's predecessor entries)
        uint32_t next_pb = nrec.x;
        asm volatile("" : "+s"(next_pb));
        u32x4 npe = sload16(pred_addr + (uint64_t)next_pb * 4);

        // ---- phase 2: insertion chain along my B cells
        float fv[B];
        uint32_t fvm[B], fvs[B];
        ChainState ex;
        auto ext_cost = [&](uint32_t s, uint32_t gsi_prev) -> float {
            if constexpr (WEIGHTED) {
                const uint32_t nw1 = n_weights - 1;
                const uint32_t wi = r.mpos + 1 + ((s - 1) - gsi_prev);
                return gpe * weights[wi < nw1 ? wi : nw1];
            } else {
                return gpe;
            }
        };
        // One cell: gs = left value + gap cost; value = the smallest of {gs, deletion, match}.  Ties
        // do not change the value (gs wins over the deletion, both win over the match: mesh.h:351-
        // 374), so the value is a plain min3 and only the trace-back indices look at the order --
        // that keeps the cell-to-cell dependency at add -> min3 -> compare -> select instead of
        // seven dependent operations.  (No NaN and no -0 can occur among these values: they are
        // sums that start at 1 or 1e6.)
        const uint32_t m_ins = kLazy ? kTbIns : (m << 16);  // tag of an insertion cell (value_midx = this row)
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
                if constexpr (FORBID) {
                    if (s > 0) {
                        bool ins = (r.smax >= 1) && (!c.e || c.gmax > 0);
                        const uint32_t gmax_n = c.e ? c.gmax - 1 : r.smax - 1;
                        float gi_cost = r.gi_open;  // opening gap (mesh.h:415-419)
                        uint32_t gsi_n = s - 1;
                        if (c.e) {  // extending gap (:420-425); gaps_val == value here
                            gi_cost = ext_cost(s, c.gsi);
                            gsi_n = c.gsi;
                        }
                        gs = ins ? (c.v + gi_cost) : r.init_v;  // untouched cell keeps its initial gaps_*
                        gsi = ins ? gsi_n : 0u;
                        gmax = ins ? gmax_n : 0u;
                        const bool take = ins && (gs <= v);  // mesh.h:351-357
                        v = take ? gs : v;
                        vm = take ? m_ins : vm;
                        vs = take ? gsi : vs;
                        const bool mtk = mt[k] < v;
                        v = mtk ? mt[k] : v;
                        vm = mtk ? mtp[k] : vm;
                        vs = mtk ? s - 1 : vs;
                    }
                } else {
                    const bool has_left = (k > 0) || has_col_left;  // s > 0
                    float gi_cost = r.gi_open;  // opening gap (mesh.h:340-343)
                    uint32_t gsi_n = s - 1;
                    if (c.e) {  // extending gap (:344-349); gaps_val == value here
                        gi_cost = ext_cost(s, c.gsi);
                        gsi_n = c.gsi;
                    }
                    const float gsx = c.v + gi_cost;  // (left.v = +inf where nothing is to my left)
                    const float a = min2_raw(gsx, dv[k]);
                    v = min3_raw(gsx, dv[k], mt[k]);
                    const bool take = gsx <= dv[k];    // mesh.h:351-357
                    const bool mtk = mt[k] < a;        // :360-374
                    vm = mtk ? mtp[k] : (take ? m_ins : vm);
                    vs = mtk ? s - 1 : (take ? gsi_n : vs);
                    gs = has_left ? gsx : 1.0f;
                    gsi = has_left ? gsi_n : 0u;
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

        // lane 0 of a later strip knows its real left state already (the edge record of the row)
        ChainState strip_left;
        strip_left.v = 0.f;
        strip_left.e = strip_left.gsi = strip_left.gmax = 0;
        if (have_left_strip) {
            strip_left.v = __uint_as_float(cur_edge.y);
            strip_left.e = cur_edge.z >> 31;
            strip_left.gsi = cur_edge.z & 0x7fffffffu;
            strip_left.gmax = FORBID ? cur_edge.w : 0u;
        }
        // "no gap enters from the left": the cell to my left did not end in a gap and is so
        // expensive that opening from it can never win
        ChainState none;
        none.v = __builtin_inff();
        none.e = none.gsi = none.gmax = 0;
        ChainState left = none;
#ifdef SH_PROF_TIMERS
        int it_ = 0;
#endif
        bool done = false;
        // Simple scheme (gap_open >= gap_extend): a gap that enters my cells from the left either
        // runs through ALL of them (every cell: gap <= min(deletion, match) candidates) and leaves
        // as the same gap, B adds later -- or it dies at some cell, and from that cell on everything
        // is what the cells compute without it (the gap candidate they see from their own left
        // neighbour is no smaller than the real one, so it loses there as well).  So:
        //   1. my cells without any gap from the left: values, trace-back tags, exit state sx;
        //   2. exit states propagate lane to lane, B adds + B compares per step (after the first
        //      step a log-step scan guesses all of them, the steps then verify), until none changes;
        //   3. the cells the entering gap wins -- a prefix of my cells -- are overwritten.
        if constexpr (!WEIGHTED && !FORBID) {
            if (gp >= gpe && !SH_ABL(1)) {
                // 1. the cells as if no gap entered: values, tags and my exit state sx
                run_chain(none);
                const ChainState sx = ex;
                float loc[B];
#pragma unroll
                for (int k = 0; k < B; k++) loc[k] = min2_raw(dv[k], mt[k]);
                SH_PROF(4)
                // does a gap enter any lane at all?  (in half of the rows none does: ex = sx, done)
                left.v = lane_shr1(sx.v);
                left.e = lane_shr1(sx.e);
                left.gsi = 0;
                if (lane == 0) left = strip_left;
                bool enter = has_col_left && (left.v + (left.e ? gpe : gp) <= loc[0]);  // the gap from my left wins at least my first cell
                if (__any(enter)) {
                    for (int guard = 0; guard < (1 << 20); ++guard) {
                        if (guard > 0) {
                            left.v = lane_shr1(ex.v);
                            left.e = lane_shr1(ex.e);
                            if (lane == 0) left = strip_left;
                        }
                        const ChainState prev = ex;
                        float g = left.v + (left.e ? gpe : gp);
                        enter = has_col_left && (g <= loc[0]);
                        bool pass = enter;
#pragma unroll
                        for (int k = 1; k < B; k++) {
                            g = g + gpe;
                            pass = pass && (g <= loc[k]);
                        }
                        ex.v = pass ? g : sx.v;
                        ex.e = pass ? 1u : sx.e;
                        if (!__any(!same_state<kGsi>(ex, prev))) break;
                        SH_PROF_CNT(10, 1)
#ifdef SH_PROF_TIMERS
                        it_++;
#endif
                        if (guard == 0 && !SH_ABL(8)) {
                            // A gap runs through a whole lane.  Stepping lane by lane costs one iteration per
                            // lane a run crosses, so first GUESS all exit states with a log-step scan and
                            // let the iterations above verify the guess (any start converges to the one
                            // consistent set of states, so this only changes the number of iterations).
                            // A stretch of lanes acts on the gap candidate x arriving at its first cell
                            // as  x <= th ? (x + cells * gpe, extending) : C  with a constant
                            // state C; two stretches compose to one of the same form.  th and the sums
                            // are computed with single adds where the cells do repeated ones, which is
                            // the same float except at rare roundings -- hence a guess, not the result.
                            float th = loc[0];
#pragma unroll
                            for (int k = 1; k < B; k++) th = min2_raw(th, loc[k] - (float)k * gpe);
                            float th1 = lane_shr1(th);               // (the step just done was offset 1:
                            th = min2_raw(th1, th - (float)B * gpe);  //  ex = lane (i) after lane (i-1)'s sx)
                            if (lane <= 1) th = -__builtin_inff();  // lane 0's left state is known: constant
                            float cv = ex.v;
                            uint32_t ce = ex.e;
                            // (unrolled, the 8-column variants no longer fit 3 waves per SIMD)
                            constexpr int kScanUnroll = B > 8 ? 5 : 1;
#pragma unroll kScanUnroll
                            for (int o = 2; o < 64; o *= 2) {
                                const int src = (lane - o) << 2;
                                // (the gap candidate a lane hands to its right neighbour is computed where
                                // the state is: two values travel instead of three)
                                const float xout = cv + (ce ? gpe : gp);
                                const float pth = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(th)));
                                const float x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(xout)));
                                const bool valid = lane >= o;
                                const bool hit = valid && (x <= th);
                                cv = hit ? x + (float)(o * B - 1) * gpe : cv;
                                ce = hit ? 1u : ce;
                                th = valid ? min2_raw(pth, th - (float)(o * B) * gpe) : th;
                            }
                            ex.v = cv;
                            ex.e = ce;
                        }
                    }
                    // 3. cells the entering gap wins (a prefix of my cells: it extends while it is no
                    // worse than the local candidates) become insertion cells; behind the cell where it
                    // loses everything is as computed in 1.  (The last iteration above ran with the
                    // final left states: `left` and `enter` are current.)
                    float g = left.v + (left.e ? gpe : gp);
                    bool alive = enter;
#pragma unroll
                    for (int k = 0; k < B; k++) {
                        if (k > 0) {
                            g = g + gpe;
                            alive = alive && (g <= loc[k]);
                        }
                        fv[k] = alive ? g : fv[k];
                        fvm[k] = alive ? m_ins : fvm[k];
                    }
                }
                done = true;
            }
        }
        // General case (weighted / forbid schemes): full chains, first without an incoming gap.
        // The speculation is exact unless the gap arriving from the left wins my first cell
        // (gs <= value after deletions, and no match beats it): one add and two compares per
        // lane verify that.  (With --insertion=forbid a cell that may NOT take a gap keeps its
        // initial gaps_val, which the shortcut cannot see: always re-run there.)  Then re-run the
        // chains with the real left states until no exit state changes.
        if (!done) {
            run_chain(none);
            bool rerun = true;
            if constexpr (!FORBID) {
                left.v = lane_shr1(ex.v);
                left.e = lane_shr1(ex.e);
                left.gsi = lane_shr1(ex.gsi);
                if (lane == 0) left = strip_left;
                bool take0 = false;
                if (has_col_left) {
                    const float g0 = left.v + (left.e ? ext_cost(s0, left.gsi) : r.gi_open);
                    take0 = (g0 <= dv[0]) && !(mt[0] < g0);
                }
                rerun = __any(take0);
                if (SH_ABL(1)) rerun = false;
            }
            SH_PROF(4)
            if (rerun) {
                for (int guard = 0; guard < (1 << 20); ++guard) {
                    SH_PROF_CNT(10, 1)
#ifdef SH_PROF_TIMERS
                    it_++;
#endif
                    const ChainState prev = ex;
                    left.v = lane_shr1(ex.v);
                    left.e = lane_shr1(ex.e);
                    left.gsi = lane_shr1(ex.gsi);
                    left.gmax = FORBID ? lane_shr1(ex.gmax) : 0u;
                    if (lane == 0) left = strip_left;
                    if (has_col_left) run_chain(left);
                    if (!__any(!same_state<kGsi>(ex, prev))) break;
                }
            }
        }
#ifdef SH_PROF_TIMERS
        {   // histogram of rerun iterations per row: 0,1,2,3,4,5-8,9-16,17-32,33+
            const int b_ = it_ <= 4 ? it_ : (it_ <= 8 ? 5 : (it_ <= 16 ? 6 : (it_ <= 32 ? 7 : 8)));
            pa_[16 + b_] += 1;
        }
#endif
        SH_PROF(5)

        // (the scalar loads for the next row: waited for here, before this row's LDS stores go out)
        sload_wait(npe);
        if (have_left_strip) sload_wait(nedge);
        // ---- publish: the edge record for the strip to my right (its column s0-1 value and chain
        // state); the row itself into its LDS slot or spill row, unless nothing will ever read it
        if (lane == 63 && have_right_strip) {
            EdgeRec er;
            er.bnd = fv[B - 1];
            er.xv = ex.v;
            er.xe = (ex.e << 31) | (kGsi ? ex.gsi : 0u);
            er.gmax = FORBID ? ex.gmax : 0u;
            e_out[m] = er;
        }
        if (r.keep != kRowNone && !SH_ABL(32)) {
            if (!(r.keep & kRowSpilled)) {
                unsigned char *myslot = ring + (size_t)r.keep * kSlotBytes;
                store_slot<B>(reinterpret_cast<float *>(myslot), lane, fv);
                store_slot<B>(reinterpret_cast<float *>(myslot + kValBytes), lane, gm);
                if (lane == 0 && have_left_strip) *reinterpret_cast<float *>(myslot + 2 * kValBytes) = edge_val;
            } else if (!SH_ABL(2)) {
                float *row = spill + (size_t)(r.keep & ~kRowSpilled) * (2 * Lp);
                store_cells<B>(row + s0, fv);
                store_cells<B>(row + Lp + s0, gm);
            }
        }
        SH_PROF(6)

        // ---- trace-back cells: the only per-cell HBM traffic
        if (!SH_ABL(4)) {
            uint32_t tc[B];
#pragma unroll
            for (int k = 0; k < B; k++)
                tc[k] = kLazy ? (fvm[k] | (oplast[k] ? 0u : kTb16XLast)) : (fvm[k] | fvs[k] | (oplast[k] ? kTagOpLast : 0u));
            if constexpr (kLazy) {
                uint32_t tp[B / 2];  // two 16-bit cells per word
#pragma unroll
                for (int k = 0; k < B / 2; k++) tp[k] = tc[2 * k] | (tc[2 * k + 1] << 16);
                store_cells<B / 2>(reinterpret_cast<uint32_t *>(tb + (size_t)m * Lp + s0), tp);
            } else {
                store_cells<B>(tb + (size_t)m * Lp + s0, tc);
            }
        }
        if constexpr (DBG) {  // (test hook sina_hip_debug_mesh: the value plane of the launch's first query)
            if (qi == 0) store_cells<B>(dbg_value + (size_t)m * Lp + s0, fv);
        }

        // ---- end-cell search, step 1: rows at the last query column (one lane of one strip)
        if (strip == strip_last && !SH_ABL(64)) {
            // fv[k_last], k_last wave-uniform: a scalar branch picks the group of four, three selects
            // the cell (B select masks would not fit the SGPR budget and come back from spill lanes
            // every row)
            float v = 0.f;
            auto pick4 = [&](auto g4_tag) {
                constexpr int g4 = decltype(g4_tag)::value;
                if constexpr (4 * g4 < B) {
                    if (g4 == kg_last) {
                        // (bit-field inserts with 32-bit all-or-nothing masks instead of selects, and a
                        // volatile asm to keep the branch: otherwise the compiler turns the whole
                        // construct into a dynamically indexed array in scratch memory)
                        uint32_t t = __float_as_uint(fv[4 * g4]);
                        t = (kr_is1 & __float_as_uint(fv[4 * g4 + 1])) | (~kr_is1 & t);
                        t = (kr_is2 & __float_as_uint(fv[4 * g4 + 2])) | (~kr_is2 & t);
                        t = (kr_is3 & __float_as_uint(fv[4 * g4 + 3])) | (~kr_is3 & t);
                        asm volatile("" : "+v"(t));
                        v = __uint_as_float(t);
                    }
                }
            };
            pick4(std::integral_constant<int, 0>{});
            pick4(std::integral_constant<int, 1>{});
            pick4(std::integral_constant<int, 2>{});
            static_assert(B <= 12, "pick4 covers three groups of four");
            if (own_last && (!lc_any || v < lc_min)) {
                lc_min = v;
                lc_arg = m;
                lc_any = true;
            }
            if (own_last && is_sink && !sk_any) lc_snk0 = v;  // value of sinks[0] at column L-1
        }
        // step 2: sink rows x every column of this strip
        if (is_sink && !SH_ABL(64)) {
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
        SH_PROF(7)
        SH_PROF_CNT(8, 1)
        cur = nrec;
        cur_edge = nedge;
        cur_pe = npe;
#pragma unroll
        for (int i = 0; i < B / 4; i++) {
            prev_v.v[i] = typename Cells<B>::V{fv[4 * i], fv[4 * i + 1], fv[4 * i + 2], fv[4 * i + 3]};
            prev_g.v[i] = typename Cells<B>::V{gm[4 * i], gm[4 * i + 1], gm[4 * i + 2], gm[4 * i + 3]};
        }
        prev_edge_val = edge_val;
    }
    // this strip's best sink cell joins the earlier strips': smaller value, then smaller sink id,
    // then smaller column (mesh.h:579-592 scans sinks ascending, columns ascending, strict <)
    if (sk_any && (!all_any || sk_min < all_min || (sk_min == all_min && sk_m < all_m))) {
        all_min = sk_min;
        all_m = sk_m;
        all_s = sk_s;
    }
    all_any = all_any || sk_any;
    // my edge records and spill rows must have left this CU before the next strip reads them back
    // (through the scalar cache / from another lane)
    if (have_right_strip) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_dcache_inv();  // (the scalar cache is not coherent with those stores)
    }
    }  // strips
    SH_PROF_FLUSH

    // ---- combine (mesh.h:567-592): the lane that owns column L-1 has step 1
    const float v1min = __shfl(lc_min, lane_last);
    const float v_snk0 = __shfl(lc_snk0, lane_last);
    const uint32_t v1arg = __shfl(lc_arg, lane_last);
    if (lane == 0) {
        DpResult r;
        r.status = 0;
        // m = sinks[0]; replaced only by a strictly smaller value, first such row wins
        uint32_t em = snk0, es = L - 1;
        float ev = v_snk0;
        if (v1min < v_snk0) {
            em = v1arg;
            ev = v1min;
        }
        // sinks x columns, strict <, scan order (t asc, x asc)
        if (all_any && all_min < ev) {
            em = all_m;
            es = all_s;
            ev = all_min;
        }
        if (!all_any) r.status = -2;
        r.end_m = em;
        r.end_s = es;
        r.raw = ev;
        r.rows_done = r.cells_done = r.attempts = r.pad_ = 0u;  // (no row skip in the general kernel)
        r.gain0 = 0.f;
        r.ubound = __builtin_inff();
        resv[qi] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// The same recurrence for the case every BASELINE configuration runs: scoring_scheme_simple (constant
// gap costs, gap_open >= gap_extend), transition_simple, 16-bit trace-back cells, a launch whose values
// provably stay below the 1e6 initial value (dp_below_init).  Same data structures, same strips, same
// scalar prefetch, same edge records as mesh_dp_kernel above -- what differs is the instruction
// count per row, the binding resource of this kernel (VALU issue):
//   * comp(row base, query base) for a row with a single-base mask (all but IUPAC ambiguity rows) is ONE
//     v_cmp_class_f32 per cell: the row's base is handed over as a float of one of four classes
//     (+0, +1, +inf, -1), the query base of a cell is kept as the matching class mask -- instead of
//     and + compare;
//   * the insertion chain needs ONE compare per cell: with loc = min(deletion, match candidates),
//     the gap candidate gs wins the cell iff gs <= loc (it beats the deletions on <=, mesh.h:351,
//     and a match only wins on <, :369), the cell's value is then gs itself, and "gaps_val == value",
//     the extend condition of the next cell (:340), is that same predicate; which of deletion / match
//     a cell falls back to is decided once, off the chain (ltag);
//   * "extends" flags travel between lanes as a shifted wave mask (one scalar shift), values by DPP;
//   * the OpLast bit of a trace-back cell is not kept as a lane mask per cell and merged with two
//     instructions: the deletion tag of an extending predecessor carries a second bit (kTb16XLast)
//     and one bit-field insert moves the LAST predecessor's into the finished cell;
//   * no inline asm in the recurrence (the hazard recogniser pads v_cmp -> v_cndmask pairs it cannot
//     see through).
// Measured and NOT kept (round 3, 3072 16S queries, tools/perf_dp.py; kept version 600 Gcell/s):
// scalar loads two rows ahead + the first predecessor's LDS row fetched while the previous row is
// being finished (605 with spill rows sharing the prefetch registers -- every row then waits for the
// previous row's trace-back store --, 563 with three code copies per source, 587 with the scalar
// prefetch alone: the loop is bound by instruction issue of its three waves per SIMD, not by these
// latencies); four waves per SIMD (128 VGPRs, 12 spilled dwords, two LDS slots: 437 against 466 on
// a 4096-query launch); B = 12 (418); the relaxation's adds two at a time as v_pk_add_f32 (2.6 % fewer VALU
// instructions, 592: a packed add issues slower than the two adds it replaces).
// Everything else (weighted scheme, --insertion=forbid, gap_open < gap_extend, huge gap costs) runs
// mesh_dp_kernel.  Results are bit-identical between the two: tests/test_gpu_parity.py runs every
// simple-scheme plane test through both (SINA_HIP_TEST=generic=1 forces the generic kernel).
template <int B, bool DBG, bool PRUNE>
#ifndef SINA_DP_SIMPLE_WAVES8
#define SINA_DP_SIMPLE_WAVES8 3  // waves per SIMD the B = 8 kernel is compiled for
#endif
__global__ void __launch_bounds__(64, (B <= 4 ? 4 : (B <= 8 ? SINA_DP_SIMPLE_WAVES8 : 2)))
mesh_dp_simple_kernel(const QDesc *__restrict__ qdv, const uint32_t *__restrict__ orderv, const uint4 *__restrict__ recv,
                      const uint32_t *__restrict__ predv, const uint8_t *__restrict__ qmaskv, void *__restrict__ tbv,
                      float *__restrict__ dbg_value, float *spillv, EdgeRec *edgev, uint64_t edge_stride,
                      uint32_t n_strips, DpResult *__restrict__ resv, float ms, float mms, float gp, float gpe, DryArgs dry,
                      const uint2 *__restrict__ reachv, float prune_rho, uint32_t prune_amax, uint32_t n_slots,
                      const float *__restrict__ scout_uv, float scout_bias) {
    static_assert(B % 4 == 0, "16-byte accesses per array");
    constexpr int kStrip = 64 * B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    dry_signal(dry, gridDim.x, lane == 0);  // (the launch queued behind this one may start once the last of us has: ctx.h)
    const uint32_t qi = orderv[blockIdx.x];
    const QDesc d = qdv[qi];
    const uint32_t N = uniform(d.N), L = uniform(d.L);
    const uint32_t Lp = n_strips * (uint32_t)kStrip;
    const uint32_t S = min(n_strips, (L - 1) / (uint32_t)kStrip + 1);
    const uint64_t node_off = uniform(d.node_off);
    const uint64_t erec_off = uniform(d.erec_off);  // (a multiple of 4 records: 64-byte lines of this query's own)
    unsigned char *ring = smem;
    constexpr size_t kValBytes = (size_t)kStrip * 4;
    constexpr size_t kSlotBytes = 2 * kValBytes + 16;
    const uint4 *__restrict__ rec = recv + node_off;
    const uint32_t *__restrict__ pred = predv + uniform(d.edge_off);
    uint16_t *__restrict__ tb = reinterpret_cast<uint16_t *>(tbv) + uniform(d.tb_off);
    float *spill = spillv + uniform(d.spill_off) * (size_t)(2 * Lp);
    const uint64_t q_off = uniform(d.q_off);

    const uint32_t strip_last = (L - 1) / (uint32_t)kStrip;
    const int lane_last = (int)(((L - 1) / B) & 63u);
    const int k_last = (int)((L - 1) % B);

    // ---- certified row skip (PRUNE; the bound and its units: common.h, the proof: DESIGN.md 3.1).  A cell whose
    // value exceeds T(m, s) = U + min(a * (L-1-s), R(m)) cannot lie on a path that ends at U or below, and whatever
    // is computed FROM such a cell exceeds T where it arrives: a row whose inputs in this strip all exceed their T
    // is not swept at all -- its successors read kDead in its place -- and every cell at or below its T still comes
    // out bit for bit.  If the end cell found has a value <= U, it, its value and every cell of its trace-back
    // path are the full sweep's (certificate); if not, the wave sweeps again with a bound that cannot fail twice:
    // attempt 1 U = -rho * (bound on the whole gain), rho the launch's guess; 2: U = what attempt 1 found (a real
    // path's cost); 3: no bound.  All bounds are integers in units of 1/64 (exact in float32 below 2^24 units).
    constexpr int32_t kNoBound = 1 << 29;
    constexpr float kDead = 1000000.0f;  // what a skipped row shows its successors: the reference's own "unreached" (mesh.h:290)
    // per node {R(m) in units, id of its last successor (0: none) | C(m) << 16}: what a row found alive can still reach
    const uint2 *__restrict__ reach = PRUNE ? reachv + node_off : nullptr;
    const uint32_t gmin = uniform(d.gmin);  // what a path forfeits per column it leaves out, units

    const uint32_t first_sink = uniform(d.first_sink);
    int32_t U64 = kNoBound, U_scout = kNoBound;
    float gain0 = 0.f;
    // U + R(m) - gmin * max(0, C(m) - r), units: the bound's second term at a cell with r query bases to come
    auto col_bound = [&](uint2 rc, int32_t r) -> int32_t {
        const int32_t excess = (int32_t)(rc.y >> 16) - r;
        return U64 + (int32_t)rc.x - (excess > 0 ? (int32_t)gmin * excess : 0);
    };
    if constexpr (PRUNE) {
        // the whole alignment right of the first node, its own column included (<= one step's largest gain)
        // (all columns: those right of node 0's and its own)
        const uint2 r0 = reach[0];
        const int32_t excess0 = (int32_t)(r0.y >> 16) + 1 - (int32_t)(L - 1);
        const uint32_t g_cols = uniform(r0.x) + prune_amax - (excess0 > 0 ? gmin * (uint32_t)excess0 : 0u), g_len = prune_amax * (L - 1);
        const uint32_t g0 = g_cols < g_len ? g_cols : g_len;
        gain0 = (float)g0 * kPruneUnit;
        U64 = -(int32_t)uniform((uint32_t)(int32_t)(prune_rho * (float)g0));  // (rounded towards zero: the looser side)
        // ... or, better than any guess, what the scout pass found: the cost of a real path of THIS query (scout.hip:
        // its alignment against the chain of its nearest relative), rounded up to a unit.  (Beyond +-1e5 the units leave float32's exact
        // integers: the guess stands.)  The store's guess stays as a guard, six per cent looser than it is used alone:
        // a scout that lost the query (a long gap its relative does not share: a path found, but a poor one) would have
        // its query sweep twice the rows of the others, and a launch ends with its slowest wave.  If the guard is too bold for
        // this query -- a query much further from its family than the store's others -- the first attempt dies early
        // and cheaply, and the second runs under the scout's value (below).
        if (scout_uv != nullptr) {
            const float su = __uint_as_float(uniform(__float_as_uint(scout_uv[qi]))) + scout_bias;  // (bias: a test hook, 0)
            if (su > -100000.0f && su < 100000.0f) {
                const float up = su * 64.0f;
                int32_t u = (int32_t)up;
                if ((float)u < up) u++;
                U_scout = u;
                const int32_t guard = -(int32_t)uniform((uint32_t)(int32_t)(fmaxf(prune_rho - 0.06f, 0.f) * (float)g0));
                U64 = u < guard ? u : guard;
            }
        }
    }
    uint32_t rows_done = 0, cells_done = 0, attempt = 0;
    const float gmin_f = (float)gmin * kPruneUnit;
    SH_PROF_DECL
    uint32_t res_m = 0, res_s = 0;
    float res_v = 0.f;
    int32_t res_status = 0;
    for (;;) {  // attempts (one without PRUNE)
    ++attempt;
    const float U_f = (float)U64 * kPruneUnit;
    // end-cell search (mesh.h:567-592), accumulated over the strips.  (PRUNE: the search starts at sinks[0], a row
    // the wave may never visit -- its id comes with the query, its value is kDead unless it is swept)
    float lc_min = PRUNE ? __builtin_inff() : 0.f, lc_snk0 = PRUNE ? kDead : 0.f;
    uint32_t lc_arg = 0;
    bool lc_any = false;
    float all_min = __builtin_inff();
    uint32_t all_m = 0, all_s = 0xffffffffu, snk0 = PRUNE ? first_sink : 0u;
    bool all_any = PRUNE;  // (a DAG has a sink: its last row)
    // Where an alignment may start for free -- column 0 of any row, any column of a row without predecessors: initial
    // value 1 -- is above its bound from this row on, whatever the column: the bound's second term falls with the row
    // (a column further right: R loses that column's gain, gmin or more, and C - r one column, worth gmin) and, in a
    // row, with the column, so its value at column 0 is a threshold, found once per attempt.
    uint32_t m_free_dead = 0;
    // rows of the strip just finished whose last cell was at or below its bound (first, last; none: first = ~0): all
    // a later strip can start from
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t out_first = kNone, out_last = 0;
    if constexpr (PRUNE) {
        if (U64 + (int32_t)(prune_amax * (L - 1)) >= 64) {
            uint32_t lo = 0, hi = N;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (col_bound(reach[mid], (int32_t)(L - 1)) < 64) hi = mid;
                else lo = mid + 1;
            }
            m_free_dead = lo;
        }
    }

#ifdef SH_PROF_TIMERS
    if (attempt == 1 && lane == 0 && blockIdx.x < 16384) g_dp_span[2 * blockIdx.x] = wall_clock64();
#endif
    // trace-back tags (16-bit cells, common.h)
    constexpr uint32_t kXL = kTb16XLast, kExtXL = kTb16Ext | kTb16XLast;
    const float gpv = opaque_v(gp), gpev = opaque_v(gpe);  // gap costs as select operands

    for (uint32_t strip = 0; strip < S; ++strip) {
    const uint32_t s0 = (strip * 64u + (uint32_t)lane) * B;
    const bool lane0 = lane == 0;
    const bool col0_mine = (strip == 0) && lane0;  // my cell 0 is query column 0
    const uint64_t e_in = uniform((uint64_t)(edgev + (size_t)(strip ? strip - 1 : 0) * edge_stride + erec_off));
    EdgeRec *e_out = edgev + (size_t)strip * edge_stride + erec_off;
    const bool have_left_strip = uniform((uint32_t)(strip > 0)) != 0, have_right_strip = strip + 1 < S;

    // the query base of my columns as v_cmp_class_f32 masks: A -> +0, G -> +normal, C -> +inf,
    // U -> -normal (class bits 6, 8, 9, 3); 0 beyond L: matches nothing
    uint32_t qcls[B];
#pragma unroll
    for (int k = 0; k < B; k++) {
        const uint32_t s = s0 + k;
        const uint32_t q = (s < L) ? (uint32_t)(qmaskv[q_off + s] & 0xf) : 0u;
        qcls[k] = ((q & 1u) << 6) | ((q & 2u) << 7) | ((q & 4u) << 7) | (q & 8u);
    }
    const bool own_last = (strip == strip_last) && (lane == lane_last);
    float sk_min = __builtin_inff();
    uint32_t sk_m = 0, sk_s = 0xffffffffu;
    bool sk_any = false;

    const uint64_t pred_addr = uniform((uint64_t)pred);
    // ---- row skip, the strip's limits.  A row of this strip can only be at or below its bound through the cell left
    // of it (strip 0: column 0), a predecessor row, or by being a free start.  So the sweep starts at the first row
    // the strip to the left left alive and ends -- cut, at stop_at -- once the row is past the last of those
    // (edge_end), past every successor of every row found alive, and free starts are out of the question (their
    // threshold row, or a strip whose first column already rules them out).
    uint32_t m_begin = 0, edge_end = 0, stop_at = 0;
    if constexpr (PRUNE) {
        // (a row without predecessors below m_free_dead can be at or below its bound in this strip)
        const bool free_starts = U64 + (int32_t)prune_amax * ((int32_t)(L - 1) - (int32_t)(strip * (uint32_t)kStrip)) >= 64;
        if (strip == 0) {
            edge_end = m_free_dead;
        } else {
            edge_end = out_first != kNone ? out_last + 1u : 0u;
            if (!free_starts) m_begin = out_first != kNone ? out_first : N;
        }
        stop_at = (free_starts && m_free_dead > edge_end) ? m_free_dead : edge_end;
        out_first = kNone;
        out_last = 0;
    }
    const uint32_t jump_lo = m_begin;  // rows below were not swept in this strip: what reads one reads kDead
    uint4 cur = rec[m_begin < N ? m_begin : 0u];
    u32x4 cur_pe = sload16(pred_addr + (uint64_t)cur.x * 4);
    u32x4 cur_edge = {0, 0, 0, 0};
    if (have_left_strip) cur_edge = sload16(e_in + (uint64_t)(m_begin < N ? m_begin : 0u) * sizeof(EdgeRec));
    // (the row's {R(m), last successor | C(m)}: fetched a row ahead like its record -- loaded where it is used, the
    // wait for it was a wait for every scalar load in flight, the next row's record and edge record included)
    uint2 reach_cur = {0u, 0u};
    if constexpr (PRUNE) reach_cur = reach[m_begin < N ? m_begin : 0u];
    sload_wait(cur_pe);
    sload_wait(cur_edge);
    Cells<B> prev_v, prev_g;  // the row just finished (common.h: a row whose only successor is the next row is kept nowhere else)
    float prev_edge_val = 0.f;
#pragma unroll
    for (int i = 0; i < B / 4; i++) prev_v.v[i] = prev_g.v[i] = typename Cells<B>::V{0.f, 0.f, 0.f, 0.f};

    // ---- row skip, per strip: the bound of my first cell (the largest of my cells' bounds), how many rows in a row
    // right before this one were dead in this strip, which LDS slots already show kDead, whether the row before was
    // not swept (the hand-over registers then hold some older row: a skipped row touches NO vector register -- anything
    // it changed there would be a copy on the swept rows' path, where the loop's two back edges meet), the rows swept
    float A_lane = 0.f, r_lane = 0.f;
    uint32_t dead_run = (PRUNE && m_begin > 0) ? 64u : 0u;
    uint32_t slot_dead = 0, rows_strip = 0;
    bool prev_dead = PRUNE && m_begin > 0;
    // whether cur_pe holds THIS row's predecessor entries (a skipped row does not fetch the next row's)
    bool pe_valid = true;
    if constexpr (PRUNE) {
        A_lane = (float)(U64 + (int32_t)prune_amax * ((int32_t)(L - 1) - (int32_t)s0)) * kPruneUnit;
        r_lane = (float)((int32_t)(L - 1) - (int32_t)s0);  // query bases to come behind my first cell
        if (m_begin > 0) {
            // the sweep starts behind rows that may own row slots: every slot shows kDead until a swept row takes it
            // (what a row of this strip reads in a slot then is its predecessor's or kDead, never an older sweep's)
            float dead_cells[B];
#pragma unroll
            for (int k = 0; k < B; k++) dead_cells[k] = kDead;
            for (uint32_t x = 0; x < n_slots; x++) {
                unsigned char *slot = ring + (size_t)x * kSlotBytes;
                store_slot<B>(reinterpret_cast<float *>(slot), lane, dead_cells);
                store_slot<B>(reinterpret_cast<float *>(slot + kValBytes), lane, dead_cells);
                if (lane0) *reinterpret_cast<float *>(slot + 2 * kValBytes) = kDead;
            }
            slot_dead = (1u << n_slots) - 1u;
        }
    }

    for (uint32_t m = m_begin; m < N; ++m) {
        if constexpr (PRUNE) {
            if (m >= stop_at) break;  // the cut: nothing from here on can be at or below its bound
        }
        if ((m & 127u) == 0) issue_priority_by_progress(strip * N + m, S * N);
        const uint32_t m_next = m + 1 < N ? m + 1 : m;
        const uint4 nrec = rec[m_next];
        uint2 nreach = {0u, 0u};
        if constexpr (PRUNE) nreach = reach[m_next];
        u32x4 nedge = {0, 0, 0, 0};
        if (have_left_strip) nedge = sload16(e_in + (uint64_t)m_next * sizeof(EdgeRec));

        SH_PROF(0)
        if constexpr (PRUNE) {
            // (the strip to the left was cut before this row: what memory holds in place of its edge record is some
            // earlier sweep's -- the row's cells there are above their bounds, it shows what a skipped row shows)
            if (have_left_strip && m >= edge_end) {
                cur_edge.x = cur_edge.y = __float_as_uint(kDead);
                cur_edge.z = cur_edge.w = 0u;
            }
        }
        // ---- row scalars
        const uint32_t r_pb = cur.x, r_z = cur.z, r_keep = cur.w;
        const uint32_t npred = r_z & 0xffu, mmask = (r_z >> 8) & 0xfu;
        const float wgt = __uint_as_float(cur.y);
        const float vM = ms * wgt, vX = mms * wgt;  // scoring_schemes.h:154
        const float edge_val = __uint_as_float(cur_edge.x);
        const bool is_sink = (r_z & kRecSink) != 0;

        // ---- row skip: nothing that enters this row in this strip can still lie on a path that ends at U or below
        // -- the cell left of my strip (strip 0: column 0's initial value 1) is above its bound, and so were all
        // cells (this strip's and the one left of it) of all rows back to the furthest predecessor -- so neither
        // can anything in it.  Scalar only: the row's record, a shift register of the last 64 rows' verdicts, the
        // flag the strip to the left stored with its edge record (EdgeRec::gmax).
        bool edge_dead = false;
        uint2 cur_reach = {0u, 0u};
        float Bm_row = 0.f, nc_row = 0.f;
        if constexpr (PRUNE) {
            // (edge records of rows at or beyond edge_end may be stale: the strip to the left was cut before them)
            edge_dead = m >= edge_end || (have_left_strip && cur_edge.w == 0u);
            const uint32_t dist = (r_z >> kRecDistShift) & 63u;  // (63 = "further, or no predecessor": never skipped)
            const bool skip = edge_dead && dist != kRecDistFar && dist <= dead_run;
            if (skip) {
                if (lane == 63 && have_right_strip) {
                    EdgeRec er;
                    er.bnd = kDead;
                    er.xv = kDead;
                    er.xe = 0u;
                    er.gmax = 0u;
                    e_out[m] = er;
                }
                if (r_keep != kRowNone) {
                    float dead_cells[B];
#pragma unroll
                    for (int k = 0; k < B; k++) dead_cells[k] = kDead;
                    if (!(r_keep & kRowSpilled)) {
                        if (!((slot_dead >> r_keep) & 1u)) {
                            unsigned char *myslot = ring + (size_t)r_keep * kSlotBytes;
                            store_slot<B>(reinterpret_cast<float *>(myslot), lane, dead_cells);
                            store_slot<B>(reinterpret_cast<float *>(myslot + kValBytes), lane, dead_cells);
                            if (lane0) *reinterpret_cast<float *>(myslot + 2 * kValBytes) = kDead;
                            slot_dead |= 1u << r_keep;
                        }
                    } else {
                        float *row = spill + (size_t)(r_keep & ~kRowSpilled) * (2 * Lp);
                        store_cells<B>(row + s0, dead_cells);
                        store_cells<B>(row + Lp + s0, dead_cells);
                    }
                }
                prev_dead = true;
                SH_PROF_CNT(28, 1)
                if constexpr (DBG) {
                    if (qi == 0) {
                        float dead_cells[B];
#pragma unroll
                        for (int k = 0; k < B; k++) dead_cells[k] = kDead;
                        store_cells<B>(dbg_value + (size_t)m * Lp + s0, dead_cells);
                    }
                }
                ++dead_run;
                if (have_left_strip) sload_wait(nedge);
                cur = nrec;
                cur_edge = nedge;
                reach_cur = nreach;
                pe_valid = false;
                continue;
            }
            ++rows_strip;
            if (!pe_valid) {  // (the row before was skipped: my predecessor entries were not fetched ahead)
                cur_pe = sload16(pred_addr + (uint64_t)r_pb * 4);
                sload_wait(cur_pe);
            }
            cur_reach = reach_cur;
            // (U + R(m)) and C(m) of this row, as floats: every term a multiple of 1/64 below 2^18 -- exact in float32
            Bm_row = (float)(U64 + (int32_t)cur_reach.x) * kPruneUnit;
            nc_row = (float)(cur_reach.y >> 16);
        }

        // ---- match / mismatch score of my cells against this row: comp() = (row mask & query mask) != 0
        // (aligned_base.h:153), one class test per set bit of the row's mask (an empty mask: NaN, in no class mask)
        // (computed BEHIND the first predecessor row's LDS reads, in each of the branches below: the reads' round trip
        // then runs under these twenty instructions instead of in front of the first candidate)
        float csel[B];
        auto match_scores = [&]() {
            auto base_float = [](uint32_t bit) -> float {  // class of the lowest set bit of `bit`
                return __uint_as_float((bit & 1u) ? 0u : ((bit & 2u) ? 0x3f800000u : ((bit & 4u) ? 0x7f800000u : ((bit & 8u) ? 0xbf800000u : 0x7fc00000u))));
            };
            const float rf = base_float(mmask);
#pragma unroll
            for (int k = 0; k < B; k++) csel[k] = __builtin_amdgcn_classf(rf, (int)qcls[k]) ? vM : vX;
            for (uint32_t mm = mmask & (mmask - 1); mm != 0; mm &= mm - 1) {  // IUPAC ambiguity rows: further bases
                const float rf2 = base_float(mm);
#pragma unroll
                for (int k = 0; k < B; k++) csel[k] = __builtin_amdgcn_classf(rf2, (int)qcls[k]) ? vM : csel[k];
            }
        };

        SH_PROF(1)
        // ---- phase 1: deletion / match candidates from the predecessor rows, in ascending id order
        // (first minimum wins; the LAST predecessor defines gapm, mesh.h:315-323).  What the chain needs
        // of it: loc = the best of them, ltag = its tag (a match only wins on <, the first deletion keeps
        // a tie), gm = the last predecessor's deletion candidate, tl = that candidate's tag.
        float loc[B], gm[B];
        uint32_t ltag[B], tl[B];
        // a predecessor row {value, gapm_val} of my columns + value[p][s0-1]: out of its LDS slot or spill row
        auto load_pred = [&](uint32_t pe, Cells<B> &sv, Cells<B> &sg, float &left_of_strip) {
            if (pe & kPredSpilled) {
                if (PRUNE && (pe & 0xffffu) < jump_lo) {  // a row the strip's sweep started behind: its spill row is some older sweep's
#pragma unroll
                    for (int i = 0; i < B / 4; i++) sv.v[i] = sg.v[i] = typename Cells<B>::V{kDead, kDead, kDead, kDead};
                    left_of_strip = kDead;
                    return;
                }
                const float *row = spill + (size_t)((pe >> 16) & 0x7FFFu) * (2 * Lp);
                sv.load(row + s0);
                sg.load(row + Lp + s0);
                left_of_strip = 0.f;
                // (PRUNE: a row at or beyond edge_end was not swept in the strip to the left -- its spill row's
                // columns there are some earlier sweep's)
                if (lane0 && have_left_strip) left_of_strip = (PRUNE && (pe & 0xffffu) >= edge_end) ? kDead : row[s0 - 1];
                // (consume the global loads inside this rare branch: mesh_dp_kernel)
#pragma unroll
                for (int i = 0; i < B / 4; i++) {
                    asm volatile("" : "+v"(sv.v[i]));
                    asm volatile("" : "+v"(sg.v[i]));
                }
                asm volatile("" : "+v"(left_of_strip));
            } else {
                const unsigned char *slot = ring + (size_t)(pe >> 16) * kSlotBytes;
                sv.load_slot(reinterpret_cast<const float *>(slot), lane);
                sg.load_slot(reinterpret_cast<const float *>(slot + kValBytes), lane);
                left_of_strip = 0.f;
                if (have_left_strip) left_of_strip = *reinterpret_cast<const float *>(slot + 2 * kValBytes);
            }
        };
        // the last predecessor (only it can be the previous row, ids ascend): out of the registers the
        // previous row left it in, or loaded INTO them -- the previous row is needed by nothing else then
        auto last_pred = [&](uint32_t pe, float &left_of_strip) {
            if (PRUNE && prev_dead && (pe & 0xffffu) + 1 == m) {  // the previous row was not swept: it shows kDead
                // (an opaque value: as a plain constant the compiler filled 17 registers with it IN FRONT of this
                // test, on every row, and copied the previous row's 16 into place behind it for all the others)
                const float kd = opaque_v(kDead);
#pragma unroll
                for (int i = 0; i < B / 4; i++) prev_v.v[i] = prev_g.v[i] = typename Cells<B>::V{kd, kd, kd, kd};
                left_of_strip = kd;
            } else if ((pe & 0xffffu) + 1 == m) {
                left_of_strip = prev_edge_val;
            } else {
                load_pred(pe, prev_v, prev_g, left_of_strip);
            }
        };
        auto left_value = [&](const Cells<B> &sv, float left_of_strip) -> float {
            float svl = lane_shr1(sv[B - 1]);  // value[p][s0-1] lives in the lane to my left
            if (lane0) svl = left_of_strip;
            return svl;
        };
        if (npred == 1) {
            // ---- one predecessor (60 % of the rows): its candidates ARE the best ones
            float los;
            last_pred(cur_pe.x, los);
            match_scores();
            const float svl = left_value(prev_v, los);
#pragma unroll
            for (int k = 0; k < B; k++) {
                const float v = prev_v[k] + gp;  // deletion (mesh.h:307-330)
                const float g = prev_g[k] + gpe;
                const bool op = v < g;
                const float cand = sel(op, v, g);
                const uint32_t ts = op ? 0u : kExtXL;  // ordinal 0, kTbDel == 0
                gm[k] = cand;
                tl[k] = ts;
                const float mv = ((k == 0) ? svl : prev_v[k - 1]) + csel[k];  // match from (p, s-1) (:360-374)
                if (k > 0) {  // below_init: the deletion candidate always beats the 1e6 initial value
                    const bool mwin = mv < cand;
                    loc[k] = sel_a(mwin, mv, cand);  // (cand lives on as gm[k]: hiding IT cost a copy per cell)
                    ltag[k] = mwin ? kTbMatch : ts;
                } else {      // ... but my cell 0 may be column 0: initial value 1, no match step
                    const bool better = !col0_mine || cand < 1.0f;
                    const float dv0 = better ? cand : 1.0f;
                    const uint32_t dvm0 = better ? ts : kTbNone;
                    const bool mwin = !col0_mine && mv < dv0;
                    loc[k] = mwin ? mv : dv0;
                    ltag[k] = mwin ? kTbMatch : dvm0;
                }
            }
        } else if (npred == 0) {
            // ---- a source row: every cell starts at 1 (init_edge) and stays untouched
#pragma unroll
            for (int k = 0; k < B; k++) {
                loc[k] = gm[k] = 1.0f;
                ltag[k] = kTbNone;
                tl[k] = 0;
            }
        } else {
            // ---- several predecessors
            float dv[B], mt[B];
            uint32_t dvm[B], mtp[B];
            auto relax = [&](auto first_tag, auto last_tag, uint32_t ord, const Cells<B> &sv, const Cells<B> &sg, float svl) {
                constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
                const uint32_t p_open = ord << kTb16OrdShift;  // kTbDel == 0
                const uint32_t p_ext = p_open | kExtXL;
                const uint32_t p_match = p_open | kTbMatch;
#pragma unroll
                for (int k = 0; k < B; k++) {
                    const float v = sv[k] + gp;
                    const float g = sg[k] + gpe;
                    const bool op = v < g;
                    const float cand = sel(op, v, g);
                    const uint32_t ts = op ? p_open : p_ext;
                    if constexpr (LAST) {
                        gm[k] = cand;
                        tl[k] = ts;
                    }
                    const float mv = ((k == 0) ? svl : sv[k - 1]) + csel[k];
                    if constexpr (FIRST) {
                        if (k > 0) {
                            dv[k] = cand;
                            dvm[k] = ts;
                            mt[k] = mv;  // (values are finite: beats the initial +inf)
                            mtp[k] = p_match;
                        } else {
                            const bool better = !col0_mine || cand < 1.0f;
                            dv[k] = better ? cand : 1.0f;
                            dvm[k] = better ? ts : kTbNone;
                            mt[k] = col0_mine ? __builtin_inff() : mv;
                            mtp[k] = col0_mine ? 0u : p_match;
                        }
                    } else {
                        const bool better = cand < dv[k];
                        dv[k] = sel(better, cand, dv[k]);
                        dvm[k] = better ? ts : dvm[k];
                        const bool mb = (k > 0) ? (mv < mt[k]) : (!col0_mine && mv < mt[k]);
                        mt[k] = sel(mb, mv, mt[k]);
                        mtp[k] = mb ? p_match : mtp[k];
                    }
                }
            };
            {
                Cells<B> sv, sg;
                float los;
                load_pred(cur_pe.x, sv, sg, los);
                match_scores();
                relax(std::true_type{}, std::false_type{}, 0u, sv, sg, left_value(sv, los));
                for (uint32_t e = 1; e + 1 < npred; ++e) {
                    const uint32_t pe = e == 1 ? cur_pe.y : (e == 2 ? cur_pe.z : (e == 3 ? cur_pe.w : pred[r_pb + e]));
                    load_pred(pe, sv, sg, los);
                    relax(std::false_type{}, std::false_type{}, e, sv, sg, left_value(sv, los));
                }
            }
            {
                const uint32_t e = npred - 1;
                const uint32_t pe = e == 1 ? cur_pe.y : (e == 2 ? cur_pe.z : (e == 3 ? cur_pe.w : pred[r_pb + e]));
                float los;
                last_pred(pe, los);
                relax(std::false_type{}, std::true_type{}, e, prev_v, prev_g, left_value(prev_v, los));
            }
#pragma unroll
            for (int k = 0; k < B; k++) {
                const bool mwin = mt[k] < dv[k];
                loc[k] = sel(mwin, mt[k], dv[k]);
                ltag[k] = mwin ? mtp[k] : dvm[k];
            }
        }
        uint32_t next_pb = nrec.x;
        asm volatile("" : "+s"(next_pb));
        u32x4 npe = sload16(pred_addr + (uint64_t)next_pb * 4);

        SH_PROF(3)
        // ---- phase 2: the insertion chain
        // 1. my cells as if no gap entered from the left (cell 0 then takes no gap at all)
        float fv[B];
        uint32_t fvm[B];
        fv[0] = loc[0];
        fvm[0] = ltag[0];
        // column 0 has no insertion step and keeps gaps_val = 1 (init_edge): it "extends" iff its value is 1
        bool e_prev = col0_mine && (fv[0] == 1.0f);
#pragma unroll
        for (int k = 1; k < B; k++) {
            const float gsx = fv[k - 1] + (e_prev ? gpev : gpv);  // mesh.h:340-349
            const bool ins = gsx <= loc[k];                        // :351; = "gaps_val == value" of this cell
            fv[k] = ins ? gsx : loc[k];
            fvm[k] = ins ? kTbIns : ltag[k];
            e_prev = ins;
        }
        const float sx_v = fv[B - 1];
        SH_PROF(4)
        // ("extends" flags of the whole wave as 64-bit lane masks in scalar registers from here on: a
        // neighbour's flag is a shift away, and the compiler cannot turn them into byte vectors)
        const uint64_t sx_em = __builtin_amdgcn_ballot_w64(e_prev);
        // 2. exit states lane to lane (see mesh_dp_kernel): a gap that enters my cells either runs
        // through all of them or dies inside and leaves my exit state as computed above
        const float sl_v = have_left_strip ? __uint_as_float(cur_edge.y) : __builtin_inff();  // left of lane 0
        const uint64_t sl_e = have_left_strip ? (uint64_t)(cur_edge.z >> 31) : 0ull;
        float ex_v = sx_v;
        uint64_t ex_em = sx_em;
        {
            auto mask = [](uint64_t m) -> bool { return __builtin_amdgcn_inverse_ballot_w64(m); };
            float left_v = lane_shr1(sx_v);
            if (lane0) left_v = sl_v;
            float g0 = left_v + (mask((sx_em << 1) | sl_e) ? gpev : gpv);
            // (row skip: a gap that enters my cells ABOVE the bound of my first cell -- the largest of my cells' bounds
            // -- is not followed.  Whatever it would set is above its bound, and stays so to the right: bounds fall
            // with the column by more than a gap costs nothing.  The invariant asks of such cells only that they stay
            // above their bounds; dropping a candidate can only raise them.  Right of a query's band every lane is
            // entered by the chain running out of the band: half the rows' chain phases were spent on cells nobody reads.)
            float t_first = __builtin_inff();
            if constexpr (PRUNE) t_first = min2_raw(A_lane, __builtin_fmaf(-gmin_f, __builtin_fmaxf(nc_row - r_lane, 0.f), Bm_row));
            // (two wave masks ANDed on the scalar unit: the ballot of the conjunction went through a 0 / 1 register
            // and a second compare)
            auto enters = [&](float gin) -> uint64_t {
                uint64_t e = __builtin_amdgcn_ballot_w64(gin <= loc[0]);
                if constexpr (PRUNE) e &= __builtin_amdgcn_ballot_w64(gin <= t_first);
                return e;
            };
            uint64_t enter = enters(g0);  // (column 0: left_v = +inf)
            if (enter != 0) {
                SH_PROF_CNT(12, 1)
                float g[B];
                uint64_t pass[B];
                for (int guard = 0; guard < (1 << 20); ++guard) {
                    SH_PROF_CNT(13, 1)
                    SH_PROF_CNT(16 + (guard < 8 ? guard : 8), 1)
                    g[0] = g0;
                    pass[0] = enter;
#pragma unroll
                    for (int k = 1; k < B; k++) {
                        g[k] = g[k - 1] + gpe;
                        pass[k] = pass[k - 1] & __builtin_amdgcn_ballot_w64(g[k] <= loc[k]);
                    }
                    const float nv = mask(pass[B - 1]) ? g[B - 1] : sx_v;
                    const uint64_t ne = pass[B - 1] | sx_em;
                    const uint64_t changed = __builtin_amdgcn_ballot_w64(nv != ex_v) | (ne ^ ex_em);
                    ex_v = nv;
                    ex_em = ne;
                    if (changed == 0) break;
                    if (guard == SINA_DP_SCAN_AT) {
                        SH_PROF_CNT(14, 1)
                        // a gap runs through a whole lane: GUESS all exit states with a log-step scan, the
                        // iterations then verify the guess (mesh_dp_kernel; single fused multiply-adds
                        // stand for the cells' repeated adds -- a guess may be off at a rounding)
                        // (the guess follows the rule above too: a lane is only entered at or below its first cell's bound --
                        // without it the guess ran gaps on through the dead lanes right of the band, and seven in ten of the
                        // rows that come here paid an iteration to take them back)
                        float th = PRUNE ? min2_raw(loc[0], t_first) : loc[0];
#pragma unroll
                        for (int k = 1; k < B; k++) th = min2_raw(th, __builtin_fmaf(-(float)k, gpe, loc[k]));
                        const float th1 = lane_shr1(th);
                        th = min2_raw(th1, __builtin_fmaf(-(float)B, gpe, th));
                        if (lane <= 1) th = -__builtin_inff();
                        float cv = ex_v;
                        uint64_t cem = ex_em;
#pragma unroll
                        for (int o = 2; o < 64; o *= 2) {
                            const int src = (lane - o) << 2;
                            const float xout = cv + (mask(cem) ? gpev : gpv);
                            const float pth = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(th)));
                            const float x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(xout)));
                            const uint64_t valid = ~0ull << o;  // lane >= o
                            const uint64_t hit = valid & __builtin_amdgcn_ballot_w64(x <= th);
                            cv = mask(hit) ? __builtin_fmaf((float)(o * B - 1), gpe, x) : cv;
                            cem |= hit;
                            th = mask(valid) ? min2_raw(pth, __builtin_fmaf(-(float)(o * B), gpe, th)) : th;
                        }
                        ex_v = cv;
                        ex_em = cem;
                    }
                    left_v = lane_shr1(ex_v);
                    if (lane0) left_v = sl_v;
                    g0 = left_v + (mask((ex_em << 1) | sl_e) ? gpev : gpv);
                    enter = enters(g0);
                }
                // 3. the cells the entering gap wins (a prefix of mine) become insertion cells
#pragma unroll
                for (int k = 0; k < B; k++) {
                    fv[k] = mask(pass[k]) ? g[k] : fv[k];
                    fvm[k] = mask(pass[k]) ? kTbIns : fvm[k];
                }
            }
        }
        const bool ex_e = __builtin_amdgcn_inverse_ballot_w64(ex_em);
        SH_PROF(5)

        sload_wait(npe);
        if (have_left_strip) sload_wait(nedge);
        // ---- row skip: is any cell of this row (of this strip) at or below its bound?  Per lane the smallest of
        // its cells against the bound of its first one (the largest: bounds fall by a per column), capped by U + R(m)
        float Bm_f = 0.f, nc_f = 0.f;
        bool row_dead = false;
        if constexpr (PRUNE) {
            // (U + R(m)) - gmin * max(0, C(m) - r): every term a multiple of 1/64 below 2^18 -- exact in float32
            Bm_f = Bm_row;
            nc_f = nc_row;
            float lm = fv[0];
#pragma unroll
            for (int k = 1; k + 1 < B; k += 2) lm = min3_raw(lm, fv[k], fv[k + 1]);
            if constexpr (B % 2 == 0) lm = min2_raw(lm, fv[B - 1]);
            const float t_cols = __builtin_fmaf(-gmin_f, __builtin_fmaxf(nc_f - r_lane, 0.f), Bm_f);
            row_dead = !any_lane(lm <= min2_raw(A_lane, t_cols)) && (edge_dead || !have_left_strip);
            if (!row_dead && (cur_reach.y & 0xffffu) >= stop_at) stop_at = (cur_reach.y & 0xffffu) + 1u;  // (a row found alive: its successors may be)
        }
        // ---- publish: edge record for the strip to my right, the row for its successors
        // (row skip: is my last cell -- what the next strip's first column starts from -- at or below ITS bound)
        bool out_alive = false;
        if constexpr (PRUNE) {
            if (have_right_strip) {
                const float t_cols = __builtin_fmaf(-gmin_f, __builtin_fmaxf(nc_f - r_lane + (float)(B - 1), 0.f), Bm_f);
                out_alive = fv[B - 1] <= min2_raw(A_lane - (float)((B - 1) * (int32_t)prune_amax) * kPruneUnit, t_cols);
                if ((__builtin_amdgcn_ballot_w64(out_alive) >> 63) != 0ull) {
                    if (out_first == kNone) out_first = m;
                    out_last = m;
                }
            }
        }
        if (lane == 63 && have_right_strip) {
            EdgeRec er;
            er.bnd = fv[B - 1];
            er.xv = ex_v;
            er.xe = ex_e ? 0x80000000u : 0u;
            er.gmax = (PRUNE && out_alive) ? 1u : 0u;
            e_out[m] = er;
        }
        if (r_keep != kRowNone) {
            if (!(r_keep & kRowSpilled)) {
                unsigned char *myslot = ring + (size_t)r_keep * kSlotBytes;
                store_slot<B>(reinterpret_cast<float *>(myslot), lane, fv);
                store_slot<B>(reinterpret_cast<float *>(myslot + kValBytes), lane, gm);
                if (lane0 && have_left_strip) *reinterpret_cast<float *>(myslot + 2 * kValBytes) = edge_val;
                if constexpr (PRUNE) slot_dead &= ~(1u << r_keep);
            } else {
                float *row = spill + (size_t)(r_keep & ~kRowSpilled) * (2 * Lp);
                store_cells<B>(row + s0, fv);
                store_cells<B>(row + Lp + s0, gm);
            }
        }
        SH_PROF(6)
        // ---- trace-back cells: tag of the winner, XLast from the last predecessor's deletion tag
        {
            uint32_t tp[B / 2];
#pragma unroll
            for (int k = 0; k < B / 2; k++) {
                const uint32_t w = fvm[2 * k] | (fvm[2 * k + 1] << 16);
                const uint32_t x = tl[2 * k] | (tl[2 * k + 1] << 16);
                tp[k] = (w & ~(kXL | (kXL << 16))) | (x & (kXL | (kXL << 16)));
            }
            store_cells<B / 2>(reinterpret_cast<uint32_t *>(tb + (size_t)m * Lp + s0), tp);
        }
        if constexpr (DBG) {
            if (qi == 0) store_cells<B>(dbg_value + (size_t)m * Lp + s0, fv);
        }

        // ---- end-cell search, step 1: rows at the last query column (one lane of one strip)
        if (strip == strip_last) {
            float v = fv[0];
#pragma unroll
            for (int k = 1; k < B; k++) v = (k_last == k) ? fv[k] : v;  // (k_last is wave-uniform: scalar masks)
            if (own_last && (!lc_any || v < lc_min)) {
                lc_min = v;
                lc_arg = m;
                lc_any = true;
            }
            if constexpr (PRUNE) {
                if (own_last && m == first_sink) lc_snk0 = v;
            } else {
                if (own_last && is_sink && !sk_any) lc_snk0 = v;
            }
        }
        // step 2: sink rows x every column of this strip
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
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(bv, off);
                const uint32_t os = __shfl_xor(bs, off);
                const bool b = (ov < bv) || (ov == bv && os < bs);
                bv = b ? ov : bv;
                bs = b ? os : bs;
            }
            if (!PRUNE && !sk_any) snk0 = m;
            if (bv < sk_min) {
                sk_min = bv;
                sk_m = m;
                sk_s = bs;
            }
            sk_any = true;
        }
        SH_PROF(7)
        SH_PROF_CNT(8, 1)
        SH_PROF_CNT(15, npred >= 2 ? 1 : 0)
        SH_PROF_CNT(25, npred)
        SH_PROF_CNT(26, r_keep != kRowNone ? 1 : 0)
        SH_PROF_CNT(27, is_sink ? 1 : 0)
        cur = nrec;
        cur_edge = nedge;
        cur_pe = npe;
        reach_cur = nreach;
#pragma unroll
        for (int i = 0; i < B / 4; i++) {
            prev_v.v[i] = typename Cells<B>::V{fv[4 * i], fv[4 * i + 1], fv[4 * i + 2], fv[4 * i + 3]};
            prev_g.v[i] = typename Cells<B>::V{gm[4 * i], gm[4 * i + 1], gm[4 * i + 2], gm[4 * i + 3]};
        }
        prev_edge_val = edge_val;
        if constexpr (PRUNE) {
            pe_valid = true;
            prev_dead = false;
            dead_run = row_dead ? dead_run + 1u : 0u;
        }
    }
    if (sk_any && (!all_any || sk_min < all_min || (sk_min == all_min && sk_m < all_m))) {
        all_min = sk_min;
        all_m = sk_m;
        all_s = sk_s;
    }
    all_any = all_any || sk_any;
    if constexpr (PRUNE) {
        rows_done += rows_strip;
        cells_done += rows_strip * min((uint32_t)kStrip, L - strip * (uint32_t)kStrip);
    }
    // my edge records and spill rows must have left this CU before the next strip reads them back
    // (through the scalar cache, whose lines of this query's records -- none can be cached yet, the
    // region is 64-byte aligned and read by this wave only -- are dropped for good measure)
    if (have_right_strip) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_dcache_inv();
    }
    }  // strips

    {   // the end cell (every lane: the certificate below is the wave's decision)
        const float v1min = __shfl(lc_min, lane_last);
        const float v_snk0 = __shfl(lc_snk0, lane_last);
        const uint32_t v1arg = __shfl(lc_arg, lane_last);
        res_status = 0;
        uint32_t em = snk0, es = L - 1;
        float ev = v_snk0;
        if (v1min < v_snk0) {
            em = v1arg;
            ev = v1min;
        }
        if (all_any && all_min < ev) {
            em = all_m;
            es = all_s;
            ev = all_min;
        }
        if (!all_any) res_status = -2;
        res_m = uniform(em);
        res_s = uniform(es);
        res_v = __uint_as_float(uniform(__float_as_uint(ev)));
    }
    if constexpr (!PRUNE) break;
    // ---- the certificate: an end cell at or below U is the full sweep's, and so is its path
    // (a third attempt runs without a bound: it is the last whatever it finds -- a NaN fails every comparison)
    if (res_status != 0 || res_v <= U_f || attempt >= 3) break;
    // not certified: sweep again.  What attempt 1 found under too tight a bound is still the cost of a path (or
    // kDead-ish if everything was cut off): the optimum is at most that -- and if some quirk of the recurrence
    // (mesh.h:340: a gap is only extended where it IS the cell's value) makes even that wrong, attempt 2's own
    // certificate fails and attempt 3 sweeps everything.
    if (attempt == 1 && res_v < 100000.0f) {  // (beyond that the units leave float32's exact integers: sweep in full)
        const float up = res_v * 64.0f;
        int32_t u = (int32_t)up;
        if ((float)u < up) u++;
        // (the scout's value is a real path's cost too -- unless it was this attempt's bound and has just been refuted)
        U64 = (U64 != U_scout && U_scout < u) ? U_scout : u;
    } else {
        U64 = (attempt == 1 && U64 != U_scout) ? U_scout : kNoBound;
    }
    // (the edge records of this attempt's last strip boundary are rewritten before they are read again; the
    // scalar cache may still hold lines of them: dropped at the end of every strip that has a right neighbour)
    }  // attempts
    SH_PROF_FLUSH
#ifdef SH_PROF_TIMERS
    if (lane == 0 && blockIdx.x < 16384) g_dp_span[2 * blockIdx.x + 1] = wall_clock64();
#endif
    if (lane == 0) {
        DpResult r;
        r.status = res_status;
        r.end_m = res_m;
        r.end_s = res_s;
        r.raw = res_v;
        r.rows_done = PRUNE ? rows_done : 0u;
        r.cells_done = PRUNE ? cells_done : 0u;
        r.attempts = PRUNE ? attempt : 0u;
        r.gain0 = gain0;
        r.ubound = PRUNE ? (U64 == kNoBound ? __builtin_inff() : (float)U64 * kPruneUnit) : __builtin_inff();
        r.pad_ = 0u;
        resv[qi] = r;
    }
}

// One wave per query: the cell walk of backtrack() (mesh.h:594-721).  The walk is one logical thread
// (every lane computes the same thing) -- a chain of dependent look-ups, two per step when they go
// to HBM.  It only ever moves to smaller rows and smaller columns, so the wave keeps a WINDOW of the
// trace-back plane in LDS: the 64 rows x 32 columns whose upper right corner is the cell that missed,
// one row per lane, together with those rows' records, columns and first four predecessors.  A step
// inside the window costs LDS latency; a refill (two round trips, every ~30 steps) is paid by 64
// lanes at once.  LAZY: 16-bit cells (type code + predecessor ordinal) instead of 32-bit ones.
constexpr int kBtRows = 64, kBtCols = 32;
template <bool LAZY>
__global__ void __launch_bounds__(64) backtrack_kernel(BtArgs a) {
    using cell_t = typename std::conditional<LAZY, uint16_t, uint32_t>::type;
    constexpr uint32_t kAlign = 16 / sizeof(cell_t);  // cells per 16-byte load
    __shared__ __attribute__((aligned(16))) cell_t w_cell[kBtRows][kBtCols];
    __shared__ uint4 w_rec[kBtRows];   // row records, .w replaced by the node's column
    __shared__ uint4 w_pred[kBtRows];  // first four predecessor entries of the row
    const uint32_t q = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (q >= a.nq) return;
    const QDesc d = a.qd[q];
    const uint32_t L = d.L, N = d.N;
    const uint32_t Lp = a.Lp;
    const cell_t *tb = reinterpret_cast<const cell_t *>(a.tb) + d.tb_off;
    const uint4 *rec = a.rec + d.node_off;
    const uint32_t *node_pos = a.node_pos + d.node_off;
    const uint32_t *pred = a.pred + d.edge_off;
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
    o.assembled = o.nast_total = o.nast_longest = o.nast_last_run = 0;
    if (r.status != 0) {
        if (lane == 0) a.out[q] = o;
        return;
    }
    // ---- the window: rows [wr0, wr0 + 64), columns [wc0, wc0 + 32); empty until the first miss
    uint32_t wr0 = 0x80000000u, wc0 = 0;  // (no row is within 64 of that)
    auto refill = [&](uint32_t row, uint32_t col) {
        __syncthreads();  // (one wave: orders the LDS reads before against the writes below)
        wr0 = row + 1 >= (uint32_t)kBtRows ? row + 1 - kBtRows : 0u;
        const uint32_t ctop = (col & ~(kAlign - 1)) + kAlign;  // first column right of the window
        wc0 = ctop >= (uint32_t)kBtCols ? ctop - kBtCols : 0u;
        const uint32_t x = wr0 + lane;
        if (x < N) {
            const uint4 *src = reinterpret_cast<const uint4 *>(tb + (size_t)x * Lp + wc0);
            uint4 *dst = reinterpret_cast<uint4 *>(&w_cell[lane][0]);
#pragma unroll
            for (uint32_t i = 0; i < kBtCols / kAlign; i++) dst[i] = src[i];
            uint4 rx = rec[x];
            rx.w = node_pos[x];
            w_rec[lane] = rx;
            const uint32_t np = rx.z & 0xffu;
            uint4 pe = {0u, 0u, 0u, 0u};
            if (np > 0) pe.x = pred[rx.x];
            if (np > 1) pe.y = pred[rx.x + 1];
            if (np > 2) pe.z = pred[rx.x + 2];
            if (np > 3) pe.w = pred[rx.x + 3];
            w_pred[lane] = pe;
        }
        __syncthreads();
    };
    auto in_rows = [&](uint32_t row) -> bool { return row - wr0 < (uint32_t)kBtRows; };
    auto cell_at = [&](uint32_t row, uint32_t col) -> uint32_t {
        if (!(in_rows(row) && col - wc0 < (uint32_t)kBtCols)) refill(row, col);
        return (uint32_t)w_cell[row - wr0][col - wc0];
    };
    // row record / column / predecessor entry e of a row (through the window if the row is in it)
    auto rec_at = [&](uint32_t row) -> uint4 {  // (.w = the node's column)
        if (in_rows(row)) return w_rec[row - wr0];
        uint4 rx = rec[row];
        rx.w = node_pos[row];
        return rx;
    };
    auto pred_at = [&](uint32_t row, uint32_t pb, uint32_t e) -> uint32_t {
        if (e < 4 && in_rows(row)) {
            const uint4 pe = w_pred[row - wr0];
            return (e == 0 ? pe.x : (e == 1 ? pe.y : (e == 2 ? pe.z : pe.w))) & 0xffffu;
        }
        return pred[pb + e] & 0xffffu;
    };
    const uint32_t width = a.width;
    uint32_t m = r.end_m, s = r.end_s;
    uint32_t n = 0;
    const uint32_t send = L - 1;
    auto emit = [&](uint32_t p) {  // (lane 0 writes; the walk itself is wave-uniform)
        if (lane == 0) out[n] = p;
        n++;
    };

    // right hand overhang (:594-615)
    const int tail = (int)(send - s);
    o.cutoff_tail = tail;
    uint32_t c = cell_at(m, s);  // (fills the window around the end cell)
    if (tail && a.overhang != SINA_OVERHANG_REMOVE) {
        int pos = (a.overhang == SINA_OVERHANG_ATTACH) ? (int)(width - 1 - rec_at(m).w - (uint32_t)tail) : 0;
        for (int i = 0; i < tail; i++) {
            const int p = pos++;
            emit((uint32_t)(p > 0 ? p : 0));
        }
    }
    const uint8_t *qmb = a.qmask + d.q_off;
    auto mscore_at = [&](const uint4 &rx, uint32_t si) -> float {  // tr.s.match(sum, ab2, ab1) with comp()==true
        // (--fs-no-graph: the master copy takes the slave's base, its profile is compared with itself)
        if (a.self16 != nullptr) return a.self16[qmb[si] & 0xfu];
        const float wgt = __uint_as_float(rx.y);
        if (a.weights != nullptr) {
            const uint32_t nw1 = a.n_weights - 1;
            return a.ms * a.weights[rx.w < nw1 ? rx.w : nw1] * wgt;
        }
        return a.ms * wgt;
    };
    uint4 rm = rec_at(m);
    unsigned int pos = width - 1 - rm.w;
    float sum_weight = 0.f;
    int aligned = 0;
    emit(pos);
    aligned++;
    sum_weight = sum_weight + mscore_at(rm, s);

    // value_midx of a cell whose deletion extends the gap of predecessor x: gapm_idx[x][col]
    // (common.h, Ext / OpLast) -- follow last predecessors to the row that opened the gap
    constexpr uint32_t ext_bit = LAZY ? kTb16Ext : kTbExt;
    auto gapm_idx = [&](uint32_t x, uint32_t col) -> uint32_t {
        for (uint32_t guard = 0; guard < 65536u; ++guard) {
            const uint32_t cx = cell_at(x, col);  // (first: a refill brings the row's record along)
            const uint4 rx = rec_at(x);
            const uint32_t np = rx.z & 0xffu;
            if (np == 0) return 0u;  // an edge row keeps its initial gapm_idx
            const uint32_t lastp = pred_at(x, rx.x, np - 1);
            if (LAZY ? !(cx & kTb16XLast) : (cx & kTbOpLast) != 0) return lastp;
            x = lastp;
        }
        return 0u;
    };
    // the row a cell of row `row` points at, before Ext is resolved: stored (32-bit cells), or the
    // row itself / 0 / the predecessor with the stored ordinal (16-bit cells)
    auto midx_raw = [&](uint32_t cc, uint32_t row, uint32_t pb) -> uint32_t {
        if (!LAZY) return cc >> 16;
        const uint32_t t = cc & kTbTypeMask;
        if (t == kTbIns) return row;
        if (t == kTbNone) return 0u;
        return pred_at(row, pb, cc >> kTb16OrdShift);
    };
    // value_sidx of cell cc = (row, col): stored, or (type-code cells, common.h) what the type implies
    auto sidx_of = [&](uint32_t cc, uint32_t row, uint32_t col) -> uint32_t {
        if (!LAZY) return cc & kTbSMask;
        const uint32_t t = cc & kTbTypeMask;
        if (t == kTbNone) return 0u;
        if (t == kTbMatch) return col - 1;
        if (t == kTbDel) return col;
        uint32_t k = col - 1;  // insertion: the gap began where the run of insertion cells to the left ends
        while (k > 0 && (cell_at(row, k) & kTbTypeMask) == kTbIns) --k;
        return k;
    };
    auto is_deletion_at = [&](uint32_t cc, uint32_t col) -> bool {  // value_sidx == own column
        return LAZY ? (cc & kTbTypeMask) == kTbDel : (cc & kTbSMask) == col;
    };
    // :642-685 (a source node has no predecessors)
    uint32_t npred_m = rm.z & 0xffu;
    while (s != 0 && npred_m != 0) {
        const uint32_t snew = sidx_of(c, m, s);
        const uint32_t vm = midx_raw(c, m, rm.x);
        m = (c & ext_bit) ? gapm_idx(vm, s) : vm;
        c = cell_at(m, snew);
        if (snew != 0 && is_deletion_at(c, snew)) {  // the one-step deletion skip (:653-655)
            const uint32_t vm2 = midx_raw(c, m, rec_at(m).x);
            m = (c & ext_bit) ? gapm_idx(vm2, snew) : vm2;
            c = cell_at(m, snew);
        }
        rm = rec_at(m);
        npred_m = rm.z & 0xffu;
        pos = width - 1 - rm.w;
        while (s != snew) {
            --s;
            emit(pos);
            aligned++;
            sum_weight = sum_weight + mscore_at(rm, s);
        }
    }
    // left hand overhang (:690-721)
    if (s != 0) {
        o.cutoff_head = (int)s;
        if (a.overhang == SINA_OVERHANG_ATTACH) {
            while (s-- != 0) {
                ++pos;
                emit((width - 1 < pos) ? width - 1 : pos);
            }
        } else if (a.overhang == SINA_OVERHANG_EDGE) {
            int k = (int)s;
            while (k--) emit(width - (uint32_t)k - 1);
        }
    }
    o.sum_weight = sum_weight;
    o.aligned_bases = aligned;
    o.n_out = n;
    if (lane == 0) a.out[q] = o;
}


// The same walk with one LANE per query (launches of kBtLanesMin queries and more).  The wave-per-query kernel
// above spends a whole wave's issue slots on one logical thread -- 60 instructions per step, 3000 steps, 9216
// waves: 3 ms of a device whose other kernels (the next batch's DAG build, k-mer search and DP, running beside
// it) are bound by instruction issue as well.  Here a wave walks 64 queries: 64 times fewer instructions, every
// look-up a global load of its own (cells 2 bytes, a 64-byte sector each: a quarter of the window refills'
// traffic), and the walk's latency -- a few dependent round trips per step, the rare branches of any lane paid by
// the whole wave -- is hidden behind the kernels it runs beside instead of competing with them.
template <bool LAZY>
__global__ void __launch_bounds__(64) backtrack_lanes_kernel(BtArgs a) {
    using cell_t = typename std::conditional<LAZY, uint16_t, uint32_t>::type;
    const uint32_t q = blockIdx.x * 64u + threadIdx.x;
    if (q >= a.nq) return;
    const QDesc d = a.qd[q];
    const uint32_t L = d.L;
    const uint32_t Lp = a.Lp;
    const cell_t *tb = reinterpret_cast<const cell_t *>(a.tb) + d.tb_off;
    const uint4 *rec = a.rec + d.node_off;
    const uint32_t *node_pos = a.node_pos + d.node_off;
    const uint32_t *pred = a.pred + d.edge_off;
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
    o.assembled = o.nast_total = o.nast_longest = o.nast_last_run = 0;
    if (r.status != 0) {
        a.out[q] = o;
        return;
    }
    auto cell_at = [&](uint32_t row, uint32_t col) -> uint32_t { return (uint32_t)tb[(size_t)row * Lp + col]; };
    auto rec_at = [&](uint32_t row) -> uint4 {  // (.w = the node's column)
        uint4 rx = rec[row];
        rx.w = node_pos[row];
        return rx;
    };
    auto pred_at = [&](uint32_t pb, uint32_t e) -> uint32_t { return pred[pb + e] & 0xffffu; };
    const uint32_t width = a.width;
    uint32_t m = r.end_m, s = r.end_s;
    uint32_t n = 0;
    const uint32_t send = L - 1;
    auto emit = [&](uint32_t p) { out[n++] = p; };

    // right hand overhang (:594-615)
    const int tail = (int)(send - s);
    o.cutoff_tail = tail;
    uint32_t c = cell_at(m, s);
    uint4 rm = rec_at(m);
    if (tail && a.overhang != SINA_OVERHANG_REMOVE) {
        int pos = (a.overhang == SINA_OVERHANG_ATTACH) ? (int)(width - 1 - rm.w - (uint32_t)tail) : 0;
        for (int i = 0; i < tail; i++) {
            const int p = pos++;
            emit((uint32_t)(p > 0 ? p : 0));
        }
    }
    const uint8_t *qmb = a.qmask + d.q_off;
    auto mscore_at = [&](const uint4 &rx, uint32_t si) -> float {  // tr.s.match(sum, ab2, ab1) with comp()==true
        if (a.self16 != nullptr) return a.self16[qmb[si] & 0xfu];
        const float wgt = __uint_as_float(rx.y);
        if (a.weights != nullptr) {
            const uint32_t nw1 = a.n_weights - 1;
            return a.ms * a.weights[rx.w < nw1 ? rx.w : nw1] * wgt;
        }
        return a.ms * wgt;
    };
    unsigned int pos = width - 1 - rm.w;
    float sum_weight = 0.f;
    int aligned = 0;
    emit(pos);
    aligned++;
    sum_weight = sum_weight + mscore_at(rm, s);

    constexpr uint32_t ext_bit = LAZY ? kTb16Ext : kTbExt;
    auto gapm_idx = [&](uint32_t x, uint32_t col) -> uint32_t {  // (see backtrack_kernel)
        for (uint32_t guard = 0; guard < 65536u; ++guard) {
            const uint32_t cx = cell_at(x, col);
            const uint4 rx = rec[x];
            const uint32_t np = rx.z & 0xffu;
            if (np == 0) return 0u;
            const uint32_t lastp = pred_at(rx.x, np - 1);
            if (LAZY ? !(cx & kTb16XLast) : (cx & kTbOpLast) != 0) return lastp;
            x = lastp;
        }
        return 0u;
    };
    auto midx_raw = [&](uint32_t cc, uint32_t row, uint32_t pb) -> uint32_t {
        if (!LAZY) return cc >> 16;
        const uint32_t t = cc & kTbTypeMask;
        if (t == kTbIns) return row;
        if (t == kTbNone) return 0u;
        return pred_at(pb, cc >> kTb16OrdShift);
    };
    auto sidx_of = [&](uint32_t cc, uint32_t row, uint32_t col) -> uint32_t {
        if (!LAZY) return cc & kTbSMask;
        const uint32_t t = cc & kTbTypeMask;
        if (t == kTbNone) return 0u;
        if (t == kTbMatch) return col - 1;
        if (t == kTbDel) return col;
        uint32_t k = col - 1;  // insertion: the gap began where the run of insertion cells to the left ends
        while (k > 0 && (cell_at(row, k) & kTbTypeMask) == kTbIns) --k;
        return k;
    };
    auto is_deletion_at = [&](uint32_t cc, uint32_t col) -> bool {
        return LAZY ? (cc & kTbTypeMask) == kTbDel : (cc & kTbSMask) == col;
    };
    // :642-685 (a source node has no predecessors)
    uint32_t npred_m = rm.z & 0xffu;
    while (s != 0 && npred_m != 0) {
        const uint32_t snew = sidx_of(c, m, s);
        const uint32_t vm = midx_raw(c, m, rm.x);
        m = (c & ext_bit) ? gapm_idx(vm, s) : vm;
        c = cell_at(m, snew);
        if (snew != 0 && is_deletion_at(c, snew)) {  // the one-step deletion skip (:653-655)
            const uint32_t vm2 = midx_raw(c, m, rec[m].x);
            m = (c & ext_bit) ? gapm_idx(vm2, snew) : vm2;
            c = cell_at(m, snew);
        }
        rm = rec_at(m);
        npred_m = rm.z & 0xffu;
        pos = width - 1 - rm.w;
        while (s != snew) {
            --s;
            emit(pos);
            aligned++;
            sum_weight = sum_weight + mscore_at(rm, s);
        }
    }
    // left hand overhang (:690-721)
    if (s != 0) {
        o.cutoff_head = (int)s;
        if (a.overhang == SINA_OVERHANG_ATTACH) {
            while (s-- != 0) {
                ++pos;
                emit((width - 1 < pos) ? width - 1 : pos);
            }
        } else if (a.overhang == SINA_OVERHANG_EDGE) {
            int k = (int)s;
            while (k--) emit(width - (uint32_t)k - 1);
        }
    }
    o.sum_weight = sum_weight;
    o.aligned_bases = aligned;
    o.n_out = n;
    a.out[q] = o;
}


// The cseq container steps that follow the cell walk in backtrack() (src/mesh.h:603-726), for the
// queries where they are plain -- one wave per query, behind backtrack_kernel on the same stream:
//   * every emitted base is appended under the container rule (a column left of the sequence's
//     current width is moved up to it, src/cseq.cpp:79-95): a running maximum over the emitted columns;
//   * setWidth(width) + reverse() (:283-289): written back to front, column c -> width - 1 - c;
//   * fix_duplicate_positions (src/cseq.cpp:456-594): bases sharing a column are an
//     insertion; where the free columns up to the next base suffice they are placed right-aligned
//     there (and lower-cased with --lowercase=unaligned) -- the only case handled here.  An insertion
//     that does not fit and makes its neighbours move, a column at or beyond the alignment's width,
//     or more than kAsmMax bases leave out_pos as backtrack_kernel wrote it (assembled = 0): the host
//     finishes those with the container's own code.
// out_pos is rewritten in place: columns in, packed aligned bases (column | mask << 24) out.
// LDS: 4 bytes per base of the launch's longest query + 1 KB (16S: 7 KB -- what a resident DP kernel
// leaves free on a CU, so the wave runs beside it like backtrack_kernel does).
constexpr int kAsmMax = 4096;
__global__ void __launch_bounds__(64) assemble_kernel(BtArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char asm_lds[];
    unsigned long long *dupm = reinterpret_cast<unsigned long long *>(asm_lds);  // [kAsmMax / 64]
    unsigned long long *movedm = dupm + kAsmMax / 64;                              // [kAsmMax / 64]
    uint32_t *colm = reinterpret_cast<uint32_t *>(movedm + kAsmMax / 64);          // [a.asm_cap] column of emission i after the append rule
    __shared__ uint32_t facts[4];                                // ok, total, longest, last run
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    if (q >= a.nq) return;
    const QDesc d = a.qd[q];
    const sina_hip_align_out o = a.out[q];
    const uint32_t n = o.n_out, width = a.width;
    if (o.status != 0 || n == 0 || n > a.asm_cap) return;
    uint32_t *pos = a.out_pos + d.q_off;
    const uint8_t *qm = a.qmask + d.q_off;
    const bool keep_over = a.overhang != SINA_OVERHANG_REMOVE;
    const uint32_t tail = keep_over ? (uint32_t)o.cutoff_tail : 0u;
    const uint32_t n_aligned = (uint32_t)o.aligned_bases;
    // ---- append rule: prefix maximum of the emitted columns; equal neighbours are insertions
    uint32_t carry = 0;
    bool beyond = false;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        uint32_t v = i < n ? pos[i] : 0u;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(v, off);
            if ((int)lane >= off) v = max(v, y);
        }
        v = max(v, carry);
        uint32_t prev = __shfl_up(v, 1);
        if (lane == 0) prev = carry;
        const bool dup = i < n && i > 0 && v == prev;
        const unsigned long long dm = __builtin_amdgcn_ballot_w64(dup);
        if (lane == 0) {
            dupm[base >> 6] = dm;
            movedm[base >> 6] = 0ull;
        }
        if (i < n) {
            colm[i] = v;
            beyond = beyond || v >= width;
        }
        carry = __shfl(v, 63);
    }
    const bool any_beyond = __builtin_amdgcn_ballot_w64(beyond) != 0ull;
    __syncthreads();
    // ---- insertions that fit their gap (one lane: a handful of runs per query)
    if (lane == 0) {
        uint32_t total = 0, longest = 0, last_run = 0, ok = any_beyond ? 0u : 1u;
        bool first = true;
        uint32_t i = 1;
        while (ok && i < n) {
            const unsigned long long word = dupm[i >> 6] >> (i & 63);
            if (word == 0ull) {
                i = ((i >> 6) + 1) << 6;
                continue;
            }
            i += (uint32_t)__ffsll((long long)word) - 1u;  // emission i shares its column with i-1
            const uint32_t i0 = i - 1;                      // first emission of the run = its LAST base in sequence order
            uint32_t r = 0;                                 // bases to place: emissions i0 .. i0+r-1 (i0+r stays: the anchor)
            while (i < n && ((dupm[i >> 6] >> (i & 63)) & 1ull)) {
                r++;
                i++;
            }
            const uint32_t anchor_col = width - 1 - colm[i0];
            const uint32_t next_col = i0 > 0 ? width - 1 - colm[i0 - 1] : width;  // the next base in sequence order, or the end
            if (next_col - (anchor_col + 1) < r) {  // does not fit: neighbours would have to move
                ok = 0;
                break;
            }
            for (uint32_t t = 0; t < r; t++) {  // right-aligned in the gap: emission i0 next to next_col
                colm[i0 + t] = width - next_col + t;
                movedm[(i0 + t) >> 6] |= 1ull << ((i0 + t) & 63);
            }
            total += r;
            longest = max(longest, r);
            if (first) last_run = r;  // (the host walks the sequence left to right: its last run is the first one here)
            first = false;
        }
        facts[0] = ok;
        facts[1] = total;
        facts[2] = longest;
        facts[3] = last_run;
    }
    __syncthreads();
    if (!facts[0]) return;
    // ---- reverse + bases: sequence position n-1-i <- emission i
    const uint32_t keep_case = a.lowercase == SINA_LOWERCASE_ORIGINAL ? 0xFFu : 0xEFu;
    const uint32_t lower = a.lowercase == SINA_LOWERCASE_UNALIGNED ? 0x10u : 0u;
    const uint32_t first_q = (uint32_t)o.end_s + tail;  // query index of emission 0
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        if (i >= n) break;
        const bool overhang_base = i < tail || i >= tail + n_aligned;
        const bool moved = (movedm[i >> 6] >> (i & 63)) & 1ull;
        const uint32_t bits = ((uint32_t)qm[first_q - i] & keep_case) | ((overhang_base || moved) ? lower : 0u);
        pos[n - 1 - i] = ((width - 1 - colm[i]) & 0xFFFFFFu) | (bits << 24);
    }
    if (lane == 0) {
        sina_hip_align_out *out = a.out + q;
        out->assembled = 1;
        out->nast_total = facts[1];
        out->nast_longest = facts[2];
        out->nast_last_run = facts[3];
    }
}

template <int B>
int launch_tb(bool weighted, bool forbid, const DpArgs &a, uint32_t nq, uint32_t n_strips, size_t lds, hipStream_t s) {
#define SH_LAUNCH(WG, FB, BL)                                                                              \
    do {                                                                                                \
        auto kfn = a.dbg_value ? mesh_dp_kernel<B, WG, FB, BL, true> : mesh_dp_kernel<B, WG, FB, BL, false>;     \
        if (allow_full_lds(reinterpret_cast<const void *>(kfn))) return 1;                               \
        hipLaunchKernelGGL(kfn, dim3(nq), dim3(64), lds, s, a.qd, a.order, a.rec, a.pred, a.node_pos, a.succ_minpos, \
                           a.qmask, a.weights, a.n_weights, a.tb, a.dbg_value, a.spill, a.edge, a.edge_stride, \
                           n_strips, a.res, a.ms, a.mms, a.gp, a.gpe, a.prof16, a.dry);                  \
    } while (0)
    // the simple scheme with gap_open >= gap_extend in a launch below the initial value (every BASELINE
    // configuration): the specialised kernel; SINA_HIP_TEST=generic=1 keeps the generic one (parity tests)
    const bool generic_only = atoi(test_knob("generic").c_str()) != 0;
    if (!weighted && !forbid && a.below_init && a.gp >= a.gpe && !generic_only && a.prof16 == nullptr) {
        // (certified row skip: launches of two strips or more -- in a single strip column 0, where an alignment may
        // start at any row for free, keeps every row in play)
        const bool prune = a.prune && a.reach != nullptr && n_strips >= 2;
        auto kfn = prune ? (a.dbg_value ? mesh_dp_simple_kernel<B, true, true> : mesh_dp_simple_kernel<B, false, true>)
                         : (a.dbg_value ? mesh_dp_simple_kernel<B, true, false> : mesh_dp_simple_kernel<B, false, false>);
        if (allow_full_lds(reinterpret_cast<const void *>(kfn))) return 1;
        hipLaunchKernelGGL(kfn, dim3(nq), dim3(64), lds, s, a.qd, a.order, a.rec, a.pred, a.qmask, a.tb, a.dbg_value, a.spill,
                           a.edge, a.edge_stride, n_strips, a.res, a.ms, a.mms, a.gp, a.gpe, a.dry, a.reach, a.prune_rho,
                           a.prune_amax, (uint32_t)(lds / dp_slot_bytes(DpGeom{64 * (int)n_strips, B})), prune ? a.scout_u : nullptr, a.scout_bias);
    } else if (!weighted && !forbid && a.below_init) SH_LAUNCH(false, false, true);
    else if (!weighted && !forbid) SH_LAUNCH(false, false, false);
    else if (weighted && !forbid) SH_LAUNCH(true, false, false);
    else if (!weighted && forbid) SH_LAUNCH(false, true, false);
    else SH_LAUNCH(true, true, false);
#undef SH_LAUNCH
    SH_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

// The dynamic-LDS ceiling of a kernel is raised ONCE per kernel and device to all of a CU's 160 KB (a
// per-launch size would race between contexts launching from different host threads; the attribute belongs
// to the current device's function object, so a process with contexts on several devices sets it on each).
int allow_full_lds(const void *kernel) {
    static std::mutex mu;
    static std::vector<std::pair<int, const void *>> done;
    int dev = 0;
    SH_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    for (const auto &k : done)
        if (k.first == dev && k.second == kernel) return 0;
    hipFuncAttributes fa;
    SH_CHECK(hipFuncGetAttributes(&fa, kernel));  // (the ceiling is what the kernel's static LDS leaves of the 160 KB)
    const int room = 160 * 1024 - (int)fa.sharedSizeBytes;
    SH_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, room));
    done.emplace_back(dev, kernel);
    return 0;
}

// A geometry is (columns per lane B, strips S): one wave sweeps S strips of 64*B columns; T = 64 * S
// is kept as the geometry's "virtual thread count" (T * B columns).  Fat lanes amortise the per-row
// fixed work of a wave (row record, chain exchange, publish) over more cells, thin lanes need fewer
// registers and allow more waves per SIMD.  Measured on MI355X, 16S queries, DP kernel alone on a
// launch that fills every wave slot (tools/perf_dp_geoms.sh): B = 12 (2 waves/SIMD) 403 Gcell/s,
// B = 8 (3 waves/SIMD) 461, B = 4 (4 waves/SIMD) 226.  The geometry for a launch is the one with
// the smallest padded width / throughput.
static const int kLaneCells[] = {4, 8, 12};
static const double kLaneRate[] = {226.0, 461.0, 403.0};  // Gcell/s of the B = 4, 8, 12 kernels (above)
constexpr int kMaxStrips = 40;  // 40 x 256 columns (B = 4) cover SINA_HIP_MAX_QUERY_LEN

bool pick_geom(uint32_t maxL, DpGeom *g) {
    // test hook: SINA_HIP_TEST="geom=T,B" (used if it covers the batch's longest query)
    const std::string ov = test_knob("geom");
    if (!ov.empty()) {
        int t = 0, b = 0;
        if (sscanf(ov.c_str(), "%d,%d", &t, &b) == 2 && (uint32_t)(t * b) >= maxL && t % 64 == 0 && t >= 64 &&
            t <= 64 * kMaxStrips && (b == 4 || b == 8 || b == 12)) {
            g->T = t;
            g->B = b;
            return true;
        }
    }
    // one strip covers the query: the thinnest lanes that do (a wider lane would only idle -- V4
    // amplicons, 250 bases: B = 4 487 Gcell/s, B = 8 with half the lanes unused 260)
    for (int i = 0; i < 3; i++)
        if (maxL <= 64u * (uint32_t)kLaneCells[i]) {
            g->T = 64;
            g->B = kLaneCells[i];
            return true;
        }
    // several strips: B = 8 or 12 by padded width / throughput
    double best = 0;
    bool found = false;
    for (int i = 1; i < 3; i++) {
        const int b = kLaneCells[i];
        const uint32_t strips = (maxL + 64u * b - 1) / (64u * b);
        if (strips > (uint32_t)kMaxStrips) continue;
        const double cost = (double)(strips * 64u * b) / kLaneRate[i];
        if (!found || cost < best) {
            best = cost;
            g->T = (int)strips * 64;
            g->B = b;
            found = true;
        }
    }
    return found;
}

// one LDS row slot: value | gapm_val of the strip's columns + the value left of the strip
size_t dp_slot_bytes(const DpGeom &g) { return (size_t)64 * g.B * 8 + 16; }
size_t dp_fixed_lds_bytes(const DpGeom &) { return 0; }
int dp_max_ring(const DpGeom &) { return 8; }  // the slot allocators keep 8 slot states; deeper rings gain nothing
// LDS per workgroup (= per wave) that still lets the kernel's register budget decide the occupancy
size_t dp_default_lds_budget(const DpGeom &g) {
    // (B = 4: the simple kernel needs 87 VGPRs, five waves per SIMD -- 5120-query launches of V4 amplicons
    // run 11 % faster per query than 4096-query ones; the general B = 4 kernels stay at four)
    const int waves_per_simd = g.B <= 4 ? 5 : (g.B <= 8 ? SINA_DP_SIMPLE_WAVES8 : 2);
    return (size_t)160 * 1024 / (4 * waves_per_simd) - 64;
}

int launch_mesh_dp(const DpGeom &g, bool weighted, bool forbid, const DpArgs &a, uint32_t nq,
                   size_t lds, hipStream_t s) {
    const uint32_t n_strips = (uint32_t)g.T / 64;
    if (g.B == 4) return launch_tb<4>(weighted, forbid, a, nq, n_strips, lds, s);
    if (g.B == 8) return launch_tb<8>(weighted, forbid, a, nq, n_strips, lds, s);
    if (g.B == 12) return launch_tb<12>(weighted, forbid, a, nq, n_strips, lds, s);
    SH_FAIL("mesh_dp: unsupported geometry");
}

#ifdef SINA_DP_PROFILE
extern "C" int sina_hip_debug_dp_ablate(int mask) {
    SH_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_dp_abl), &mask, sizeof mask));
    return 0;
}
extern "C" int sina_hip_debug_dp_spans(unsigned long long *out, unsigned n_queries) {
#if SINA_DP_PROFILE == 1
    SH_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dp_span), sizeof(unsigned long long) * 2 * (n_queries < 16384u ? n_queries : 16384u)));
    return 0;
#else
    (void)out; (void)n_queries;
    return 1;
#endif
}
extern "C" int sina_hip_debug_dp_profile(unsigned long long *out32, int reset) {
    SH_CHECK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_dp_prof), sizeof(unsigned long long) * 32));
    if (reset) {
        unsigned long long z[32] = {};
        SH_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_dp_prof), z, sizeof z));
    }
    return 0;
}
#endif

int launch_assemble(const BtArgs &a0, hipStream_t s) {
    BtArgs a = a0;
    a.asm_cap = std::min<uint32_t>((a.asm_cap + 63u) & ~63u, (uint32_t)kAsmMax);  // (longer queries: the host finishes them)
    const size_t lds = 2 * (size_t)(kAsmMax / 64) * 8 + 4 * (size_t)a.asm_cap;
    hipLaunchKernelGGL(assemble_kernel, dim3(a.nq), dim3(64), lds, s, a);
    SH_CHECK(hipGetLastError());
    return 0;
}

// Launches of kBtLanesMin queries and more, none longer than kBtLanesMaxLen, walk one lane per query
// (backtrack_lanes_kernel).  The lanes' walk takes a microsecond per step whatever the launch (3.2 ms for 16S), the
// waves' walk 0.33 us of the device's instruction issue per query: at 9216 queries the two are the same 3 ms, and
// the lanes' are latency that the launches queued behind on the other FIFO stream cover (k-mer search 5.8 ms, DAG
// build 4.5) where the waves' are instructions taken from them.  A smaller launch is done sooner by waves -- with
// half the queries duplicates (4608 per launch, neighbours of 2-3 ms) lanes cost 6 %: 380 k against 406 k sequences/s,
// on the FIFO stream or off it -- and a launch of 23S-long queries (6000 steps, 14 ms) is waited for by a pipeline
// that holds two batches of them.  SINA_HIP_TEST=bt_lanes=0/1 forces one.
constexpr uint32_t kBtLanesMin = 8192, kBtLanesMaxLen = 2048;
bool backtrack_by_lanes(const BtArgs &a) {
    bool lanes = a.nq >= kBtLanesMin && a.asm_cap <= kBtLanesMaxLen;
    if (const std::string e = test_knob("bt_lanes"); !e.empty()) lanes = atoi(e.c_str()) != 0;
    return lanes;
}
int launch_backtrack(const BtArgs &a, hipStream_t s) {
    if (backtrack_by_lanes(a)) {
        if (a.lazy_sidx) hipLaunchKernelGGL(backtrack_lanes_kernel<true>, dim3((a.nq + 63u) / 64u), dim3(64), 0, s, a);
        else hipLaunchKernelGGL(backtrack_lanes_kernel<false>, dim3((a.nq + 63u) / 64u), dim3(64), 0, s, a);
    } else if (a.lazy_sidx) hipLaunchKernelGGL(backtrack_kernel<true>, dim3(a.nq), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(backtrack_kernel<false>, dim3(a.nq), dim3(64), 0, s, a);
    SH_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sina_hip
