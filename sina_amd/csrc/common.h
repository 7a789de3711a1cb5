// Shared declarations for the gfx950 kernels behind include/sina_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <ctime>

#include "sina_hip.h"

namespace sina_hip {

void set_error(const std::string &msg);

// ---- environment.  What the library reads at run time (all of it listed in INTEGRATION.md):
//   SINA_HIP_TB_GB, SINA_HIP_TB_PLANES   size / number of the device's trace-back planes (ctx.h)
//   SINA_HIP_DP_PRUNE=0                  the DP kernel sweeps every row of every strip (no certified row skip)
//   SINA_HIP_CHAIN=0|1                   chained launches off / on under a counter-collecting profiler (ctx.h)
//   SINA_HIP_NO_RUNTIME_DEFAULTS         the load-time constructor leaves the process environment alone (api.hip)
//   SINA_HIP_TRACE_ALLOC                 one line per device / pinned allocation
//   SINA_HIP_TEST="key=value;..."        test hooks (tests/ only): geom=T,B  generic=1  dense_div=N  lds_kb=N  rho=X  kmer_rows=1  bt_lanes=0/1  scout=0  scout_add=X  scout_set=X
// The experiment switches of rounds 1-4 (SINA_HIP_SERIALIZE, _DP_BURST, _GRAPH_DRY, _BT_ON_FIFO, _STREAM_PRIO,
// _SHARE_DAGS, _DP_ROUNDS, _DEBUG_SYNC) exist only in a build made with -DSINA_EXPERIMENTS
// (make -C sina_amd/csrc VARIANT=exp EXTRA=-DSINA_EXPERIMENTS): the production library does not look at them.
inline const char *experiment_env(const char *name) {
#ifdef SINA_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
// value of `key` in SINA_HIP_TEST ("" if absent); read every time: tests change it between calls
inline std::string test_knob(const char *key) {
    const char *e = getenv("SINA_HIP_TEST");
    if (!e) return std::string();
    const std::string s(e), k = std::string(key) + "=";
    size_t at = 0;
    while (at < s.size()) {
        size_t end = s.find(';', at);
        if (end == std::string::npos) end = s.size();
        if (s.compare(at, k.size(), k) == 0) return s.substr(at + k.size(), end - at - k.size());
        at = end + 1;
    }
    return std::string();
}

#define SH_CHECK(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            ::sina_hip::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));       \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

#define SH_FAIL(msg)                   \
    do {                               \
        ::sina_hip::set_error(msg);    \
        return 1;                      \
    } while (0)

// Growable device buffer (never shrinks): batches reuse their scratch.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // the largest capacity any context of the store has needed for THIS buffer so far (sina_hip_store::cap_hint;
    // nullptr: none): a buffer that has to grow goes there at once -- a context allocates only what its kind of
    // call uses, and each of those once
    const std::atomic<size_t> *hint = nullptr;
    // (hipFree / hipMalloc synchronise the whole device and stall every other context's stream:
    // grow in big steps so that batch-to-batch size jitter never reallocates in steady state)
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        if (hint) want = std::max(want, hint->load(std::memory_order_relaxed));
        trace_alloc(want);
        SH_CHECK(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    // SINA_HIP_TRACE_ALLOC=1: one line per device allocation (there should be none in steady state)
    static void trace_alloc(size_t bytes) {
        static const bool on = getenv("SINA_HIP_TRACE_ALLOC") != nullptr;
        if (on) {
            timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            fprintf(stderr, "[sina_hip] %.3f hipMalloc %.1f MB\n", ts.tv_sec % 1000 + ts.tv_nsec * 1e-9, bytes / 1048576.0);
        }
    }
    int reserve_exact(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        trace_alloc(bytes);
        SH_CHECK(hipMalloc(&p, bytes));
        cap = bytes;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Growable pinned host staging buffer.  Copies between the device and PAGEABLE host memory make the
// runtime wait for the stream inside the call, under locks that other threads' kernel launches need
// (a result copy queued behind a 5 ms backtrack kernel held up the next batch's DP launch for as
// long): results come back into pinned memory and are copied out after the stream has finished.
struct HostBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (getenv("SINA_HIP_TRACE_ALLOC")) {
            timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            fprintf(stderr, "[sina_hip] %.3f hipHostMalloc %.1f MB\n", ts.tv_sec % 1000 + ts.tv_nsec * 1e-9, want / 1048576.0);
        }
        SH_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// "My last workgroup has been dispatched": what a device-filling kernel tells the launch queued behind it
// (ctx.h, heavy_launch).  Every workgroup counts itself in when it starts; the one that completes the grid
// clears the counter for its next user and raises the flag word to this launch's sequence number, which a
// hipStreamWaitValue32 on the other heavy stream is waiting for.  counter == nullptr: nobody is listening.
struct DryArgs {
    uint32_t *counter, *flag;
    uint32_t seq;
};
#ifdef __HIPCC__
__device__ __forceinline__ void dry_signal(const DryArgs &d, uint32_t n_workgroups, bool one_thread) {
    if (d.counter != nullptr && one_thread) {
        if (atomicAdd(d.counter, 1u) == n_workgroups - 1u) {
            __hip_atomic_store(d.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(d.flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#endif

// ---------------------------------------------------------------- mesh DP

// Per-query descriptor of one DP problem inside a batch.
struct QDesc {
    uint64_t node_off;   // into the node arrays (rec, node_pos, succ_minpos)
    uint64_t edge_off;   // into pred
    uint64_t q_off;      // into qmask / out_pos
    uint64_t tb_off;     // into traceback plane (u32 cells), row stride = Lp
    uint64_t spill_off;  // into spill rows (row units)
    uint32_t N, L;
    uint32_t n_spill;
    uint32_t erec_off;   // into a strip boundary's edge records: regions start on 64-byte lines (4 records), see EdgeRec
    uint32_t first_sink; // sinks[0]: the row the end-cell search starts from (mesh.h:567); the row-skipping kernel may never visit it
    uint32_t gmin;       // row skip: the smallest "best gain of a column" of the DAG, units of kPruneUnit (what a path forfeits per column it leaves out)
};
static_assert(sizeof(QDesc) == 64, "uploaded as an array");

// Row record of one DAG node, read through the scalar cache once per row:
//   x = first predecessor (offset into this query's pred list)
//   y = node weight (float bits, mseq.cpp:113)
//   z = #pred | iupac mask << 8 | flags << 16 | (index of the first spilled predecessor + 1) << 24
//       flags bit0: sink = no successors; bit1: some successor is further than kFarLds rows away
//       (such a row is kept in a spill row: it would hold an LDS slot for hundreds of rows);
//       bits 2..7: distance to the furthest predecessor (kRecDistShift)
//   w = where the finished row {value, gapm_val} is kept for its successors:
//       0xFFFFFFFF nowhere (no successors, or the next row is the only one: the DP kernel hands the
//       row just finished to the next one in registers), kRowSpilled | spill row index, or the LDS
//       slot number.
// LDS slots are handed out by liveness (a slot is reused once the last successor of its row has
// been computed); a row that finds every slot busy, or that has a successor more than kFarLds
// rows away (kRecFence), goes to a spill row in HBM instead.
// Predecessor entries, ascending ids (= the reference's evaluation order, which decides ties):
//   id | (LDS slot or spill row index) << 16 | spilled << 31
constexpr uint32_t kRecSink = 1u << 16;
constexpr uint32_t kRecFence = 1u << 17;
// flags bits 2..7 (z bits 18..23): how far back the row's furthest predecessor is, in rows (1..62; 63: further, or
// the row has none).  The DP kernel's row skip asks "were ALL of the last d rows dead in this strip" instead of
// looking its predecessors up one by one (mesh_dp.hip PRUNE).
constexpr int kRecDistShift = 18;
constexpr uint32_t kRecDistFar = 63u;
constexpr uint32_t kRowNone = 0xFFFFFFFFu;
constexpr uint32_t kRowSpilled = 0x80000000u;
constexpr uint32_t kPredSpilled = 0x80000000u;
constexpr uint32_t kMaxSpillRows = 32768;  // 15 bits in a predecessor entry
constexpr int kFarLds = 192;   // a row with a successor further away than this never takes an LDS slot
// The slot allocators cut the rows into independent segments of this many rows (at most 16 of them):
// a row whose last successor lies in a later segment is kept in a spill row (graph_build.hip, step 7).
__host__ __device__ inline uint32_t dp_slot_segment(uint32_t n_rows) {
    const uint32_t s = (n_rows + 15u) / 16u;
    return s < 256u ? 256u : s;
}

// What a strip of the DP leaves behind per row for the strip to its right (mesh_dp.hip): the value of
// its last column (the match candidate's source for the next strip's first column) and the exit
// state of the insertion chain there.  One record per row and strip boundary, in global memory.
// They are written with vector stores and read back through the scalar cache, which is not coherent
// with them: a query's records therefore start on a 64-byte line of their own (QDesc::erec_off is a
// multiple of 4 records, dp_edge_entries() rounds every query up), so that no other wave's reads can
// pull a line holding records this wave has yet to write; the wave waits for its stores and drops its
// scalar-cache lines (s_dcache_inv) before the next strip reads them.
struct EdgeRec {
    float bnd;      // value[m][last column of the strip]
    float xv;       // chain exit state: value ...
    uint32_t xe;    // ... "gaps_val == value" << 31 | gaps_idx
    uint32_t gmax;  // ... gaps_max (--insertion=forbid)
};
static_assert(sizeof(EdgeRec) == 16, "read back with one 16-byte scalar load");
inline uint32_t dp_edge_entries(uint32_t n_rows) { return (n_rows + 3u) & ~3u; }  // records a query takes per strip boundary

// Trace-back cell: what backtrack() needs of a DP cell -- the only per-cell HBM traffic.  Two formats:
//
// * 16 bits (every scheme except --insertion=forbid): bits 1..0 type, bit 2 kTb16Ext, bit 3
//   kTb16XLast (the inverse of OpLast below), bits 11..4 the ORDINAL of the winning predecessor in the row's predecessor list.
//   value_midx is that predecessor (a deletion / match cell), the row itself (an insertion cell) or 0
//   (an untouched cell); value_sidx follows from the type: a match came from column s-1, a deletion
//   from s, an untouched cell keeps 0, and an insertion from the column where the run of insertion
//   cells to its left ends (a cell's gaps_val == value, the reference's "extend" condition, holds
//   exactly for insertion cells).  So the recurrence carries neither row nor column indices, and
//   backtrack looks the few of them on the final path up.
// * 32 bits (--insertion=forbid, where a cell that may not take a gap keeps its initial gaps_val and
//   the insertion-run rule above does not hold): bits 31..16 value_midx, bits 13..0 value_sidx,
//   kTbExt, kTbOpLast.
//
// Two bits replace carrying gapm_idx through the recurrence (it is only ever needed on the final path):
//   Ext:    the cell took a deletion that EXTENDS the gap of its predecessor `value_midx`; the
//           reference's value_midx is that row's gapm_idx at this column (mesh.h:318-330)
//   OpLast: this row's own gapm at this column was OPENED from its last predecessor, i.e.
//           gapm_idx = last predecessor; otherwise gapm_idx = the last predecessor's gapm_idx.
// backtrack resolves Ext by walking last predecessors until an OpLast cell.
constexpr uint32_t kTbSMask = 0x3FFFu;
constexpr uint32_t kTbExt = 1u << 14;
constexpr uint32_t kTbOpLast = 1u << 15;
static_assert(SINA_HIP_MAX_QUERY_LEN <= kTbSMask, "value_sidx field of the 32-bit trace-back cell");
constexpr uint32_t kTbTypeMask = 3u;
constexpr uint32_t kTbDel = 0u, kTbMatch = 1u, kTbIns = 2u, kTbNone = 3u;
constexpr uint32_t kTb16Ext = 1u << 2;
constexpr uint32_t kTb16XLast = 1u << 3;   // NOT OpLast: the row's gapm at this column EXTENDS its last predecessor's
constexpr int kTb16OrdShift = 4;  // 8 bits: a row has at most 255 predecessors (rec.z & 0xff)
// bytes per trace-back cell of a launch
inline size_t tb_cell_bytes(bool forbid) { return forbid ? 4 : 2; }

struct DpResult {
    uint32_t end_m, end_s;
    float raw;
    int32_t status;
    // certified row skip (mesh_dp.hip, PRUNE; zeros from a launch that does not prune): what the query's wave
    // actually swept -- (row, strip) pairs and real cells, summed over its attempts --, the attempts it took
    // (1: the launch's guess of the bound held; 2: the first attempt's own result served as the bound; 3: swept
    // in full), its bound on the gain any path can still collect at the first cell, and the bound U on the
    // optimum that the last attempt ran with and certified (+inf: swept in full)
    uint32_t rows_done, cells_done, attempts;
    float gain0, ubound;
    uint32_t pad_;
};
static_assert(sizeof(DpResult) == 40, "downloaded as an array");

// ---- certified row skip: the bound
// T(m, s) = U + min(a * r, R(m) - gmin * max(0, C(m) - r)), r = L-1-s the query bases still to come, bounds the value
// a cell may have and still lie on a path that ends at U or below: every query base gains at most a; every DAG
// column right of pos(m) at most its best node's gain (R(m) = their sum, C(m) = their number) -- and a path takes
// one column per base at most, so of C(m) columns it leaves C(m) - r out, each worth gmin or more; gaps cost (gap
// penalties >= 0).  All of it in units of 1/64 so that T is EXACT in float32 (the
// induction needs T(source) - gain >= T(target) to hold as computed, not just as written): a node's gain is
// rounded up to a multiple of 1/64 plus one unit of margin for the float rounding of the cell values.
constexpr float kPruneUnit = 1.0f / 64.0f;
// gain of a match at a node of weight w, in units (kappa64 = 64 * max(match gain per unit weight) * 1.0001)
__host__ __device__ inline uint32_t prune_gain_units(float w, float kappa64) {
    const float x = w * kappa64;
    uint32_t u = (uint32_t)x;
    if ((float)u < x) u++;
    return u + 1u;
}
// what a launch may prune with: non-negative gap costs and weights, sums that stay exact
struct PrunePlan {
    int on = 0;
    float kappa64 = 0.f;   // 64 * 1.0001 * max(-ms, -mms, 0): match gain per unit of node weight
    uint32_t amax = 0;     // largest gain of one match step in the launch, in units
};

// Every cell is at most its deletion candidate from any predecessor, value[p][s] + gap_open, and rows
// without predecessors start at 1: no value exceeds 1 + N * gap_open (the float sums stay within
// 0.1 % of that).  True if that is well below the 1e6 initial value of rows WITH predecessors for
// the longest DAG of a launch (constant gap costs only: not for the weighted scheme).
inline bool dp_below_init(uint32_t max_n, float gp, float gpe) {
    return gp >= 0.f && gpe >= 0.f && 1.0f + (float)max_n * gp * 1.01f + gp + gpe < 900000.0f;
}

struct DpArgs {
    const QDesc *qd;
    const uint32_t *order;      // workgroup -> query (decreasing N*L)
    const uint4 *rec;
    const uint32_t *pred;
    const uint32_t *node_pos;
    const uint32_t *succ_minpos;
    const uint8_t *qmask;
    int below_init;             // see dp_below_init()
    void *tb;                   // trace-back cells: u16 or u32 (kTb* above); QDesc::tb_off counts cells
    float *dbg_value;           // optional [N*Lp] plane of the first query
    float *spill;               // spill rows: value[Lp] | gapm_val[Lp]
    EdgeRec *edge;              // [strips - 1][edge_stride] edge records, indexed like the node arrays
    uint64_t edge_stride;       // records per strip boundary (= node array entries of the launch)
    DpResult *res;
    const float *weights;       // posvar weights (device) or nullptr
    uint32_t n_weights;
    float ms, mms, gp, gpe;     // scheme ctor args: -match, -mismatch, gap, gapext
    const float *prof16;        // --fs-no-graph: match term per node and query mask [16 * node + mask], else nullptr
    DryArgs dry;                // (heavy_launch::dry(): tells the next launch when this one's queue has run dry)
    // certified row skip (mesh_dp_simple_kernel<.., PRUNE>): per node the gain still to come right of its column
    // (units of kPruneUnit) and how far its successors reach, the launch's guess rho of optimum / first-cell
    // bound, the largest gain of one step; prune == 0: every row of every strip is swept
    const uint2 *reach;         // per node {R(m), id of its last successor (0: none) | C(m) << 16}, indexed like rec
    int prune;
    float prune_rho;
    uint32_t prune_amax;
    float scout_bias;           // test hook (SINA_HIP_TEST=scout_add=<x>): added to every scout value -- a scout forced wrong
    const float *scout_u;       // per query: the cost of a real path (scout.hip), the first attempt's bound U; nullptr: the guess rho
};

struct BtArgs {
    const QDesc *qd;
    const uint4 *rec;
    const uint32_t *pred;
    const uint32_t *node_pos;
    const void *tb;  // u16 cells if lazy_sidx, else u32
    const DpResult *res;
    const float *weights;
    uint32_t n_weights;
    sina_hip_align_out *out;
    uint32_t *out_pos;
    uint32_t nq, width, Lp;
    float ms;
    int overhang;
    int lazy_sidx;  // trace-back cells hold a type code, not value_sidx (see kTbTypeMask)
    // assemble_kernel (sina_hip_align_params::assemble): the query masks the bases come from, --lowercase
    const uint8_t *qmask;
    int lowercase;
    uint32_t asm_cap;  // bases of the launch's longest query (the kernel's LDS follows it)
    const float *self16;  // --fs-no-graph: comp(base, base) per iupac mask (sum_weight's term), else nullptr
};

// Picks the (threads, cells per thread) geometry for the longest query of a batch.
struct DpGeom {
    int T, B;
    int Lp() const { return T * B; }
};
bool pick_geom(uint32_t maxL, DpGeom *g);
size_t dp_slot_bytes(const DpGeom &g);
size_t dp_fixed_lds_bytes(const DpGeom &g);
int dp_max_ring(const DpGeom &g);  // deepest LDS ring the slot allocators support
size_t dp_default_lds_budget(const DpGeom &g);  // LDS per workgroup that keeps the register-limited occupancy
int launch_mesh_dp(const DpGeom &g, bool weighted, bool forbid, const DpArgs &a, uint32_t nq,
                   size_t lds_bytes, hipStream_t s);
// the scout pass (scout.hip): per query the cost of its banded alignment (kScoutBand columns per row) against the chain
// of its family's first member -- a real path of the mesh, the first attempt's bound U
constexpr int kScoutBand = 8;
int launch_chain_scout(const DpArgs &a, uint32_t nq, const uint32_t *ref_ab, const uint64_t *ref_off,
                       const uint32_t *chain_ref /* device: [nq] reference ids */, float *out_u /* [nq] */, hipStream_t s);
int launch_backtrack(const BtArgs &a, hipStream_t s);
bool backtrack_by_lanes(const BtArgs &a);  // one lane per query (large launches of 16S-long queries), else one wave per query
int launch_assemble(const BtArgs &a, hipStream_t s);  // (after launch_backtrack, same stream)
// raises a kernel's dynamic-LDS ceiling to a CU's 160 KB, once per kernel and process (mesh_dp.hip)
int allow_full_lds(const void *kernel);

// geometry + LDS ring depth chosen for a batch
struct DpPlan {
    DpGeom geom;
    int W;
    size_t lds;
};

}  // namespace sina_hip

struct sina_hip_ctx;

namespace sina_hip {
int plan_dp(sina_hip_ctx *c, uint32_t maxL, DpPlan *pl);
int upload_weights(sina_hip_ctx *c, const sina_hip_align_params *p);
// chain_ref: per query the reference id of its family's first member (host; nullptr: the DAGs are the caller's, no scout)
int run_dp_device(sina_hip_ctx *c, const DpPlan &pl, const QDesc *qd_host, uint32_t bq, uint64_t n_node_entries,
                  uint64_t tb_cells, uint64_t spill_rows, uint64_t cells, uint64_t nqm, const sina_hip_align_params *p, uint32_t width,
                  sina_hip_align_out *out, uint32_t *out_pos, bool want_dbg_value, const PrunePlan &pp,
                  const uint32_t *chain_ref = nullptr);
// What a launch may skip rows with (api.hip): the scoring of `p` (non-negative gap costs, the simple scheme), the
// largest and smallest node weight it will see, its longest query.  SINA_HIP_DP_PRUNE=0: never.
PrunePlan prune_plan(const sina_hip_align_params *p, float wmax, float wmin, uint32_t maxL, bool profile_batch);

}  // namespace sina_hip
