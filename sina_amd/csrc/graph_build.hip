// Family DAG build on the GPU + sina_hip_align_families.
//
// What it computes: mseq::mseq + sort + reduce_edges for every query's family
// (reference src/mseq.cpp:47-118, src/graph.h:332-357,451-488; spec SURVEY.md A.3):
//   * one node per (alignment column, IUPAC character incl. case) that occurs in the
//     family; node ids ascend with the column and, inside a column, with the family
//     index of the first reference showing that character;
//   * node weight from the number of references sharing the node (mseq.cpp:113);
//   * edge a->b whenever some reference has consecutive bases in nodes a, b; per node
//     the predecessor ids ascending and unique; sources / sinks implicit.
//
// How it maps to the hardware: one workgroup per query; the alignment is walked in tiles of
// kTC occupied columns.  Every step inside a tile is parallel over the family's BASES in the tile
// ("entries": member j's bases from its cursor on, a contiguous stretch of at most kTC; a wave takes
// 64 consecutive slots of one member, so the member's cursor / length / offset are scalars) and
// meets the other members of a column only through LDS atomics -- no thread ever walks the F members
// of a column (rounds 2-4 did: one thread per column for the nodes, one per node for the edges, 37 %
// of the kernel's time in loops of F iterations that a third of the threads sat out):
//   1. occupied-column bitmap in LDS (atomicOr per base), prefix-popcount -> dense
//      column index ("rank") of every alignment column;
//   2. per tile, per entry: tile column, character -> the column's set of characters (atomicOr);
//   3. nodes per column = popcount, block scan -> node ids of the columns; per entry atomicMin of the
//      member index into its (column, character) slot: the order of first appearance in the family
//      (exact reference order) is the rank of that minimum among the column's characters -- one
//      thread per column sorts its handful of minima;
//   4. per entry: its node, its predecessor's node (the entry to its left, or what the member carried
//      over from earlier tiles); member count and raw edge count of the node by atomicAdd; the
//      predecessor as ONE BIT of a 64-bit word per node (bit d-1: the node d ids back -- ids ascend
//      with the column, so the set bits from the top down are the predecessors ascending and unique);
//      the few predecessors further back than 64 ids (long deletions) by rounds of atomicMin;
//   5. one thread per node writes record and predecessor list; atomicMin / atomicMax collect the
//      successor minimum column (for --insertion=forbid) and the last successor;
//   6. sinks and fence flags; LDS-slot / spill-row assignment for the DP kernel by liveness
//      (sequential over the rows: 16 lanes, a segment each); predecessor entries; the row skip's bound.
// HBM traffic: the family's bases twice (bitmap, entries) and the DAG once.
// All arithmetic is integer except the node weight, which is looked up in a table the
// HOST computed with the reference's own mixed double/float expression.
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "ctx.h"

namespace sina_hip {
namespace {

#ifndef SINA_GRAPH_THREADS
#define SINA_GRAPH_THREADS 512
#endif
#ifndef SINA_GRAPH_MINWAVES
#define SINA_GRAPH_MINWAVES 8  // (4 workgroups of 8 waves per CU: <= 64 VGPRs, 38.5 KB of LDS each for families of 40)
#endif
constexpr int kGT = SINA_GRAPH_THREADS;  // threads per workgroup (the phases are latency-bound: more loads in flight per LDS byte)
constexpr uint32_t kNoPrev = 0xFFFFu;
constexpr int kMaxFam = 128;
constexpr int kSz = 8;  // u32 words the kernel reports per query (GraphArgs::sizes)
#ifndef SINA_GRAPH_KTC
#define SINA_GRAPH_KTC 128  // (a multiple of 64: a wave takes 64 consecutive entry slots of one member)
#endif
#ifndef SINA_GRAPH_TNW
#define SINA_GRAPH_TNW 512  // nodes whose per-node words are in LDS at a time (a tile with more takes them in column-aligned windows)
#endif
constexpr int kTC = SINA_GRAPH_KTC;  // occupied columns per LDS tile (< 255: tile columns are bytes, 255 = none)
constexpr int kTNW = SINA_GRAPH_TNW;
static_assert(kTC % 64 == 0 && kTC < 255 && kTNW >= 32 && kTNW % 2 == 0, "tile geometry");
static_assert((kMaxFam * (kTC / 64) + SINA_GRAPH_THREADS / 64 - 1) / (SINA_GRAPH_THREADS / 64) <= 64, "far-entry flags of a wave fit 64 bits");

#ifdef SINA_DP_PROFILE
// profiling build (make PROFILE=1): per-phase s_memtime totals of thread 0, tools/perf_graph.py
__device__ unsigned long long g_graph_prof[16];
#define GP_DECL unsigned long long gp_[16] = {}; unsigned long long gt_ = __builtin_amdgcn_s_memtime();
#define GP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); gp_[i] += t_ - gt_; gt_ = t_; }
#define GP_FLUSH if (threadIdx.x == 0) { for (int i_ = 0; i_ < 16; i_++) atomicAdd(&g_graph_prof[i_], gp_[i_]); }
#else
#define GP_DECL
#define GP(i)
#define GP_FLUSH
#endif

struct GraphArgs {
    const uint32_t *ref_ab;
    const uint64_t *ref_off;
    const uint32_t *fam_ids;   // concatenated
    const uint64_t *fam_off;   // [nq+1]
    const uint64_t *pred_off;  // [nq] offset of this query's pred area
    const float *wtab;         // [(kMaxFam+1) * (kMaxFam+1)]: weight for (F, count)
    uint4 *rec;                // [nq][ncap]
    uint32_t *node_pos;        // [nq][ncap]
    uint32_t *succ_min;        // [nq][ncap]
    uint32_t *far_mark;        // [nq][ncap] last successor row of every node (0: none)
    uint32_t *pred;            // per query area of total-family-bases entries
    uint32_t *sizes;           // [nq][kSz]: N, raw edge entries, n_spill, status (0 ok, 2 N cap, 4 spill rows), first sink row
    uint32_t width, ncap;
    uint32_t tile_bytes;       // LDS bytes of the tile tables (reused by the slot allocation) = bitmap_off
    uint32_t bitmap_off;       // LDS offset of the occupied-column bitmap (behind the tile tables)
    uint32_t member_off;       // LDS offset of the per-member records (behind the bitmap and its ranks)
    uint32_t member_cap;
    uint32_t fam_cap;          // rows of the entry table in LDS: the launch's largest family
    int W;                     // DP ring depth: edges longer than this need a spill row
    int want_smin;             // succ_min is read by somebody (--insertion=forbid, the debug entry): else it is not touched at all
    uint2 *reach;              // [nq][ncap] or nullptr: the DP kernel's row-skip bound (step 9; units: common.h): {R(m), last successor | C(m) << 16}
    float kappa64;             // ... 64 * 1.0001 * (largest match gain per unit of node weight)
    DryArgs dry;               // (ctx.h, heavy_launch: tells the launch queued behind when the last workgroup has started)
};

// per family member, in LDS
struct Member {
    uint64_t beg;            // offset of its bases in the store
    uint32_t len;
    uint32_t cur, curn;      // first base at/after the tile, ... after it
    uint32_t carry, carryn;  // node of the last base before the tile, ... of the last base in it
    uint32_t id;
};
static_assert(sizeof(Member) == 32, "LDS layout");
// LDS offsets of the tables whose size is a compile-time constant
constexpr uint32_t kOCpos = 0;
constexpr uint32_t kONbase = kOCpos + 4 * kTC;
constexpr uint32_t kOPres = kONbase + 4 * (kTC + 2);
constexpr uint32_t kOWA = kOPres + 4 * kTC;
constexpr uint32_t kOWC = kOWA + 4 * kTNW;
constexpr uint32_t kOWD = kOWC + 4 * kTNW;
constexpr uint32_t kOWB = (kOWD + 4 * kTNW + 7) & ~7u;
constexpr uint32_t kOLi = kOWB + 8 * kTNW;
constexpr uint32_t kONn = kOLi + 32 * kTC;
constexpr uint32_t kONfar = kONn + kTC;
constexpr uint32_t kOMask = kONfar + kTNW;
constexpr uint32_t kONcol = kOMask + kTNW;
constexpr uint32_t kOE = (kONcol + kTNW + 15) & ~15u;

// exclusive scan of in[0..n) into out[0..n) (may alias if same type); returns the total.
template <typename In, typename Out>
__device__ uint32_t block_exscan(const In *in, Out *out, uint32_t n, uint32_t *tmp) {
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (n + kGT - 1) / kGT;
    const uint32_t b = min(n, tid * chunk), e = min(n, b + chunk);
    uint32_t s = 0;
    for (uint32_t i = b; i < e; i++) s += (uint32_t)in[i];
    // exclusive scan of the kGT partial sums (wave shuffle + 4 wave totals)
    uint32_t x = s;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) tmp[wave] = x;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < kGT / 64; w++) {
        if (w < wave) base += tmp[w];
        total += tmp[w];
    }
    uint32_t run = base + x - s;
    __syncthreads();
    for (uint32_t i = b; i < e; i++) {
        const uint32_t v = (uint32_t)in[i];
        out[i] = (Out)run;
        run += v;
    }
    __syncthreads();
    return total;
}

// the same with the input behind a function of the index
template <typename Get, typename Out>
__device__ uint32_t block_exscan_f(uint32_t n, Get get, Out *out, uint32_t *tmp) {
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (n + kGT - 1) / kGT;
    const uint32_t b = min(n, tid * chunk), e = min(n, b + chunk);
    uint32_t s = 0;
    for (uint32_t i = b; i < e; i++) s += get(i);
    uint32_t x = s;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) tmp[wave] = x;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < kGT / 64; w++) {
        if (w < wave) base += tmp[w];
        total += tmp[w];
    }
    uint32_t run = base + x - s;
    __syncthreads();
    for (uint32_t i = b; i < e; i++) {
        const uint32_t v = get(i);
        out[i] = (Out)run;
        run += v;
    }
    __syncthreads();
    return total;
}

// Step 2 of a tile as a function of its own: the kernel runs with 78 scalar registers at eight waves per SIMD and
// over a hundred live scalars -- inlined, this loop reloaded a dozen spilled ones per task; outlined, the caller's
// scalars are put aside once per tile.
__device__ __attribute__((noinline)) void fill_tile(const uint32_t *__restrict__ ref_ab, uint32_t bitmap_off, uint32_t member_off,
                                                    uint32_t nwords, uint32_t n_tasks, uint32_t c0, uint32_t tc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *cposT = reinterpret_cast<uint32_t *>(smem + kOCpos);
    uint32_t *presT = reinterpret_cast<uint32_t *>(smem + kOPres);
    uint16_t *eE = reinterpret_cast<uint16_t *>(smem + kOE);
    const uint32_t *bitmap = reinterpret_cast<const uint32_t *>(smem + bitmap_off);
    const uint16_t *wrank = reinterpret_cast<const uint16_t *>(bitmap + nwords);
    Member *mb = reinterpret_cast<Member *>(smem + member_off);
    constexpr uint32_t kEAbsent = 0x00FFu, kEFirst = 1u << 13;
    constexpr uint32_t kCH = kTC / 64, kNW = kGT / 64;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    auto rank = [&](uint32_t pos) -> uint32_t {
        return (uint32_t)wrank[pos >> 5] + __popc(bitmap[pos >> 5] & ((1u << (pos & 31)) - 1u));
    };
        // (latency-bound: eight tasks' loads are in flight per wave before the first is used)
        for (uint32_t t4 = wave; t4 < n_tasks; t4 += 8 * kNW) {
            uint32_t abv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t task = t4 + (uint32_t)u * kNW;
                abv[u] = 0xFFFFFFFFu;  // (no packed base looks like this: the mask byte has five bits)
                if (task < n_tasks) {
                    // (the member is the same for the whole wave: its cursor, length and offset on the scalar unit)
                    const uint32_t j = task / kCH;
                    const uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb[j].cur);
                    const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb[j].len);
                    const uint64_t beg = mb[j].beg;
                    const uint64_t begs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(beg >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)beg);
                    const uint32_t i = cur + (task % kCH) * 64u + lane;
                    if (i < len) abv[u] = ref_ab[begs + i];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t task = t4 + (uint32_t)u * kNW;
                if (task >= n_tasks) continue;
                const uint32_t j = task / kCH, slot0 = (task % kCH) * 64u, slot = slot0 + lane;
                const uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb[j].cur);
                uint32_t ev = kEAbsent;
                if (abv[u] != 0xFFFFFFFFu) {
                    const uint32_t ab = abv[u];
                    const uint32_t pos = ab & 0xFFFFFFu;
                    const uint32_t c = rank(pos) - c0;
                    if (c < tc) {
                        const uint32_t m = (ab >> 24) & 0x1Fu;
                        ev = c | (m << 8) | (cur + slot == 0 ? kEFirst : 0u);
                        atomicOr(&presT[c], 1u << m);
                        cposT[c] = pos;
                    }
                }
                eE[j * (uint32_t)kTC + slot] = (uint16_t)ev;
                // the member's last base in this tile moves its cursor (its bases in the tile are the first lanes of
                // its tasks)
                const uint32_t n_in = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ev != kEAbsent));
                if (n_in && lane == 0) atomicMax(&mb[j].curn, cur + slot0 + n_in);
            }
        }
}

__device__ __forceinline__ uint32_t node_of_entry(uint32_t ev) {  // tile-local node of an entry: the column's first node + the place of its character
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t *nbaseT = reinterpret_cast<const uint32_t *>(smem + kONbase);
    const uint32_t *presT = reinterpret_cast<const uint32_t *>(smem + kOPres);
    const uint8_t *liOf = smem + kOLi;
    const uint32_t c = ev & 0xFFu, m = (ev >> 8) & 31u;
    const uint32_t b = nbaseT[c];
    return b + liOf[b + __popc(presT[c] & ((1u << m) - 1u))];
}
__global__ void __launch_bounds__(kGT) __attribute__((amdgpu_waves_per_eu(SINA_GRAPH_MINWAVES, SINA_GRAPH_MINWAVES))) family_graph_kernel(GraphArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_tmp[kGT / 64 + 8];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    dry_signal(a.dry, gridDim.x, tid == 0);
    const uint32_t nwords = (a.width + 31) / 32;
    const uint64_t f0 = a.fam_off[q];
    const uint32_t F = (uint32_t)(a.fam_off[q + 1] - f0);
    // LDS carve.  Everything whose size is known at compile time comes first: its addresses are immediates of the
    // LDS instructions, not scalars that have to stay live (the kernel runs with 100 SGPRs and spilled 109 more
    // when every table had a run-time base).
    uint32_t *cposT = reinterpret_cast<uint32_t *>(smem + kOCpos);          // [kTC] alignment column
    uint32_t *nbaseT = reinterpret_cast<uint32_t *>(smem + kONbase);        // [kTC + 2] first node of the column (tile-local); [tc] = nodes of the tile
    uint32_t *presT = reinterpret_cast<uint32_t *>(smem + kOPres);          // [kTC] the column's characters, bit m: mask value m occurs
    uint32_t *wA = reinterpret_cast<uint32_t *>(smem + kOWA);               // [kTNW] per node of the window: members | raw edges << 16
    uint32_t *wC = reinterpret_cast<uint32_t *>(smem + kOWC);               // [kTNW] first member of the node's character; then: smallest far predecessor not yet listed
    uint32_t *wD = reinterpret_cast<uint32_t *>(smem + kOWD);               // [kTNW] start of the CSR segment (raw entries before it in the window)
    unsigned long long *wB = reinterpret_cast<unsigned long long *>(smem + kOWB);  // [kTNW] predecessors at distance 1..64, a bit each
    uint8_t *liOf = smem + kOLi;                                            // [kTC * 32] by (column's first node + index of the character among the column's): place in family order
    uint8_t *nn = smem + kONn;                                              // [kTC] nodes per column
    uint8_t *nfarN = smem + kONfar;                                         // [kTNW] far predecessors listed so far
    uint8_t *maskN = smem + kOMask;                                         // [kTNW]
    uint8_t *ncolN = smem + kONcol;                                         // [kTNW] tile column of the node
    // entry (member j, slot): tile column | mask << 8 | "first base of the member" << 13, kEAbsent (column 255): not in the tile
    uint16_t *eE = reinterpret_cast<uint16_t *>(smem + kOE);                // [fam_cap][kTC]
    // ... then what is sized by the launch: the occupied-column bitmap and its ranks, the family members
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(smem + a.bitmap_off);   // [nwords]
    uint16_t *wrank = reinterpret_cast<uint16_t *>(bitmap + nwords);        // [nwords]
    Member *mb = reinterpret_cast<Member *>(smem + a.member_off);           // [member_cap >= 16: .id doubles as the slot allocator's per-segment counters]
    constexpr uint32_t kEFirst = 1u << 13;
    constexpr uint32_t kCH = kTC / 64, kNW = kGT / 64;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_tasks = F * kCH;  // (member, 64 slots) pairs, a wave each

    uint32_t *sz = a.sizes + kSz * (size_t)q;
    for (uint32_t j = tid; j < F; j += kGT) {
        const uint32_t id = a.fam_ids[f0 + j];
        mb[j].id = id;
        mb[j].beg = a.ref_off[id];
        mb[j].len = (uint32_t)(a.ref_off[id + 1] - a.ref_off[id]);
        mb[j].cur = mb[j].curn = 0;
        mb[j].carry = mb[j].carryn = kNoPrev;
    }
    for (uint32_t i = tid; i < nwords; i += kGT) bitmap[i] = 0;
    __syncthreads();
    GP_DECL

    // 1. occupied columns
    // (latency-bound: eight members' loads are in flight per thread before the first is used)
    for (uint32_t j0 = 0; j0 < F; j0 += 8) {
        uint32_t maxlen = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) maxlen = max(maxlen, j0 + u < F ? mb[j0 + u].len : 0u);
        for (uint32_t i = tid; i < maxlen; i += kGT) {
            uint32_t ab[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                ab[u] = (j0 + u < F && i < mb[j0 + u].len) ? a.ref_ab[mb[j0 + u].beg + i] : 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (ab[u] != 0xFFFFFFFFu) {
                    const uint32_t pos = ab[u] & 0xFFFFFFu;
                    atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
                }
        }
    }
    __syncthreads();
    GP(0)
    // 2. dense column index: rank(pos) = wrank[pos >> 5] + popc(bits below)
    {
        const uint32_t chunk = (nwords + kGT - 1) / kGT;
        const uint32_t b = min(nwords, tid * chunk), e = min(nwords, b + chunk);
        uint32_t s = 0;
        for (uint32_t i = b; i < e; i++) s += __popc(bitmap[i]);
        uint32_t x = s;
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) s_tmp[wave] = x;
        __syncthreads();
        uint32_t base = 0, total = 0;
        for (int w = 0; w < kGT / 64; w++) {
            if (w < wave) base += s_tmp[w];
            total += s_tmp[w];
        }
        uint32_t run = base + x - s;
        for (uint32_t i = b; i < e; i++) {
            wrank[i] = (uint16_t)run;
            run += __popc(bitmap[i]);
        }
        if (tid == 0) s_tmp[kGT / 64] = total;
    }
    __syncthreads();
    const uint32_t NC = s_tmp[kGT / 64];
    if (NC > 65535u) {  // (cannot be a valid DAG for the DP kernel anyway: more columns than row ids)
        if (tid == 0) {
            sz[0] = NC;
            sz[1] = sz[2] = 0;
            sz[3] = 2;
        }
        return;
    }
    auto rank = [&](uint32_t pos) -> uint32_t {
        return (uint32_t)wrank[pos >> 5] + __popc(bitmap[pos >> 5] & ((1u << (pos & 31)) - 1u));
    };
    uint4 *rec = a.rec + (size_t)q * a.ncap;
    uint32_t *node_pos = a.node_pos + (size_t)q * a.ncap;
    uint32_t *smin = a.succ_min + (size_t)q * a.ncap;
    uint32_t *last = a.far_mark + (size_t)q * a.ncap;
    uint32_t *pred = a.pred + a.pred_off[q];
    // (smin / last of a node are initialised by the tile that creates it, right before the first successor can
    // touch them: clearing all ncap entries up front wrote four times what a 16S DAG uses)
    const float *wt = a.wtab + (size_t)F * (kMaxFam + 1);
    uint32_t N = 0, E = 0;  // running totals (uniform)
    GP(1)

    for (uint32_t c0 = 0; c0 < NC; c0 += kTC) {
        const uint32_t tc = min((uint32_t)kTC, NC - c0);
        for (uint32_t i = tid; i < (uint32_t)kTC; i += kGT) presT[i] = 0;
        __syncthreads();
        GP(2)
        // 2. the entries of this tile: member j's bases cur[j].. as long as their column is in the tile (a
        // contiguous stretch of at most kTC bases)
        fill_tile(a.ref_ab, a.bitmap_off, a.member_off, nwords, n_tasks, c0, tc);
        __syncthreads();
        for (uint32_t c = tid; c < tc; c += kGT) nn[c] = (uint8_t)__popc(presT[c]);
        __syncthreads();
        GP(3)
        const uint32_t tn = block_exscan(nn, nbaseT, tc, s_tmp);
        if (tid == 0) nbaseT[tc] = tn;
        for (uint32_t ln = tid; ln < tn; ln += kGT)  // this tile's nodes: no successor seen yet
            if (N + ln < a.ncap) {
                last[N + ln] = 0;
                if (a.want_smin) smin[N + ln] = 0xFFFFFFFFu;
            }
        __syncthreads();
        GP(5)
        uint32_t te = 0;  // raw edge entries of the tile's windows so far
        for (uint32_t ca = 0; ca < tc;) {
            // the window: columns [ca, cb) -- as many as have their nodes' words in LDS together (nearly always the tile)
            const uint32_t w0 = nbaseT[ca];
            uint32_t cb = tc;
            if (tn - w0 > (uint32_t)kTNW) {
                uint32_t lo = ca + 1, hi = tc;  // (a column has at most 32 nodes: one always fits)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1) / 2;
                    if (nbaseT[mid] - w0 <= (uint32_t)kTNW) lo = mid;
                    else hi = mid - 1;
                }
                cb = lo;
            }
            const uint32_t wn = nbaseT[cb] - w0;
            for (uint32_t i = tid; i < wn; i += kGT) {
                wA[i] = 0;
                wB[i] = 0;
                wC[i] = 0xFFFFFFFFu;  // (first the smallest member index of the node's character, then the far predecessors)
                nfarN[i] = 0;
            }
            __syncthreads();
            // 3. first appearance of every (column, character) in family order
            for (uint32_t task = wave; task < n_tasks; task += kNW) {
                const uint32_t j = task / kCH, slot = (task % kCH) * 64u + lane;
                const uint32_t ev = eE[j * (uint32_t)kTC + slot];
                const uint32_t c = ev & 0xFFu;
                if (c >= ca && c < cb) {
                    const uint32_t m = (ev >> 8) & 31u;
                    atomicMin(&wC[nbaseT[c] + __popc(presT[c] & ((1u << m) - 1u)) - w0], j);
                }
            }
            __syncthreads();
            for (uint32_t c = ca + tid; c < cb; c += kGT) {
                const uint32_t base = nbaseT[c], k = nn[c];
                uint32_t pp = presT[c];
                for (uint32_t x = 0; x < k; x++) {
                    const uint32_t mj = wC[base - w0 + x];
                    uint32_t r = 0;
                    for (uint32_t y = 0; y < k; y++) r += (wC[base - w0 + y] < mj) ? 1u : 0u;
                    liOf[base + x] = (uint8_t)r;
                    maskN[base - w0 + r] = (uint8_t)(__ffs(pp) - 1);
                    ncolN[base - w0 + r] = (uint8_t)c;
                    pp &= pp - 1u;
                }
            }
            __syncthreads();
            GP(4)
            // 4. members and raw edges per node; the predecessor of every entry.  An entry's node: the column's first
            // node + the place of its character (liOf is complete for this window and the ones before: the entry to
            // the left is in one of them)
            auto node_of = [&](uint32_t ev) -> uint32_t { return node_of_entry(ev); };
            unsigned long long farbits = 0;  // my entries (by round of the task loop) whose predecessor is more than 64 ids back
            {
                uint32_t it = 0;
                for (uint32_t task = wave; task < n_tasks; task += kNW, it++) {
                    const uint32_t j = task / kCH, slot = (task % kCH) * 64u + lane;
                    const uint32_t ev = eE[j * (uint32_t)kTC + slot];
                    const uint32_t c = ev & 0xFFu;
                    const bool mine = c >= ca && c < cb;
                    const uint32_t ln = mine ? node_of(ev) : 0u;
                    // (the entry to the left is the lane to the left, except for the first slot of a task)
                    uint32_t lnp = __shfl_up(ln, 1);
                    const bool left_mine = __shfl_up(mine ? 1 : 0, 1) != 0;
                    if (!mine) continue;
                    const uint32_t node = N + ln;
                    const bool first = (ev & kEFirst) != 0;
                    atomicAdd(&wA[ln - w0], first ? 1u : 0x10001u);
                    if (mb[j].cur + slot + 1 == mb[j].curn) mb[j].carryn = node;
                    if (!first) {
                        if (slot && (lane == 0 || !left_mine)) lnp = node_of(eE[j * (uint32_t)kTC + slot - 1]);
                        const uint32_t pn = slot ? N + lnp : mb[j].carry;
                        const uint32_t d = node - pn;
                        if (d <= 64u) atomicOr(&wB[ln - w0], 1ull << (d - 1u));
                        else farbits |= 1ull << it;
                    }
                }
            }
            for (uint32_t i = tid; i < wn; i += kGT) wC[i] = 0xFFFFFFFFu;  // (the first members have been read: barrier above)
            __syncthreads();
            GP(6)
            const uint32_t te_w = block_exscan_f(wn, [&](uint32_t i) { return wA[i] >> 16; }, wD, s_tmp);
            // far predecessors, smallest first: every round lists one per node
            while (__syncthreads_or(farbits != 0 ? 1 : 0)) {
                uint32_t it = 0;
                for (uint32_t task = wave; task < n_tasks; task += kNW, it++) {
                    if (!((farbits >> it) & 1ull)) continue;
                    const uint32_t j = task / kCH, slot = (task % kCH) * 64u + lane;
                    const uint32_t ln = node_of(eE[j * (uint32_t)kTC + slot]);
                    const uint32_t pn = slot ? N + node_of(eE[j * (uint32_t)kTC + slot - 1]) : mb[j].carry;
                    atomicMin(&wC[ln - w0], pn);
                }
                __syncthreads();
                it = 0;
                for (uint32_t task = wave; task < n_tasks; task += kNW, it++) {
                    if (!((farbits >> it) & 1ull)) continue;
                    const uint32_t j = task / kCH, slot = (task % kCH) * 64u + lane;
                    const uint32_t ln = node_of(eE[j * (uint32_t)kTC + slot]);
                    const uint32_t pn = slot ? N + node_of(eE[j * (uint32_t)kTC + slot - 1]) : mb[j].carry;
                    if (wC[ln - w0] == pn) farbits &= ~(1ull << it);
                }
                __syncthreads();
                for (uint32_t i = tid; i < wn; i += kGT) {
                    const uint32_t v = wC[i];
                    if (v == 0xFFFFFFFFu) continue;
                    const uint32_t node = N + w0 + i;
                    if (node < a.ncap) {
                        pred[E + te + wD[i] + nfarN[i]] = v;
                        if (a.want_smin) atomicMin(&smin[v], cposT[ncolN[i]]);
                        atomicMax(&last[v], node);
                    }
                    nfarN[i]++;
                    wC[i] = 0xFFFFFFFFu;
                }
            }
            // 5. node records + predecessor lists: the far ones are in place, the near ones are the set bits from the top down
            for (uint32_t i = tid; i < wn; i += kGT) {
                const uint32_t node = N + w0 + i;
                if (node >= a.ncap) continue;
                const uint32_t seg = E + te + wD[i];
                const uint32_t pos = cposT[ncolN[i]];
                uint32_t np = nfarN[i];
                unsigned long long bits = wB[i];
                while (bits) {
                    const uint32_t hi = 63u - (uint32_t)__clzll((long long)bits);
                    bits ^= 1ull << hi;
                    const uint32_t pa = node - (hi + 1u);
                    pred[seg + np] = pa;
                    np++;
                    if (a.want_smin) atomicMin(&smin[pa], pos);
                    atomicMax(&last[pa], node);
                }
                uint4 r;
                r.x = seg;
                r.y = __float_as_uint(wt[wA[i] & 0xFFFFu]);
                r.z = (np & 0xFFu) | ((uint32_t)maskN[i] << 8);
                r.w = kRowNone;
                rec[node] = r;
                node_pos[node] = pos;
            }
            te += te_w;
            ca = cb;
            __syncthreads();  // (the window's words are cleared for the next one)
        }
        for (uint32_t j = tid; j < F; j += kGT) {
            mb[j].cur = mb[j].curn;
            mb[j].carry = mb[j].carryn;
        }
        N += tn;
        E += te;
        __syncthreads();
        GP(7)
    }
    if (N > a.ncap || N > 65535u) {
        if (tid == 0) {
            sz[0] = N;
            sz[1] = sz[2] = 0;
            sz[3] = 2;
        }
        return;
    }
    GP(8)
    // 6. sinks, successor minimum, fence flag -- and, in the same pass over the rows, the slot allocator's
    // row codes (step 7) when they fit the tile tables' space
    const uint32_t seg_len = dp_slot_segment(N);
    const uint32_t n_seg = (N + seg_len - 1) / seg_len;
    uint32_t *codeL = reinterpret_cast<uint32_t *>(smem);  // (the tile tables' space: everything in front of the bitmap)
    const bool in_lds = (size_t)N * 4 <= a.tile_bytes;
    if (tid == 0) s_tmp[kGT / 64 + 1] = 0xFFFFFFFFu;  // first sink row
    __syncthreads();
    for (uint32_t i = tid; i < N; i += kGT) {
        uint32_t z = rec[i].z;
        const uint32_t l = last[i];
        if (l == 0) {  // (a successor's id is greater than its predecessor's: never 0)
            z |= kRecSink;
            atomicMin(&s_tmp[kGT / 64 + 1], i);
            if (a.want_smin) smin[i] = 1000000u;  // "no successor" sentinel of mesh.h:480
        }
        if (l > i && l - i > (uint32_t)kFarLds) z |= kRecFence;
        rec[i].z = z;
        if (in_lds) codeL[i] = l | ((z & kRecSink) ? (1u << 30) : 0u) | ((z & kRecFence) ? (1u << 31) : 0u);
    }
    __syncthreads();
    GP(9)
    // 7. where every finished DP row is kept for its successors: LDS slots by liveness (first slot
    // whose row has seen its last successor), otherwise a spill row.  The greedy is sequential over
    // the rows, so the rows are cut into up to 16 segments that are allocated independently, one LANE
    // per segment: a row whose last successor lies in a later segment is kept in a spill row (three
    // or so per boundary), every segment starts with all slots free.  Spill rows are numbered in row
    // order (per-segment counts, prefix sum).  tests/util.py row_store_model states the same rule.
    // (the lanes walk their segments row by row: from LDS when the rows fit the tile tables' space --
    // a dependent global load per row was most of this step's time)
    if (tid < n_seg) {
        // (the walk is one dependent chain per lane in a CU that runs 31 other waves: every instruction waits its turn,
        // so the slot search is compiled for 4 slots -- the DP kernels keep 3 -- and for 8)
        auto walk = [&](auto kslots) {
            constexpr int K = decltype(kslots)::value;
            uint32_t fa[K];  // last successor of the row in slot x (0: empty)
#pragma unroll
            for (int x = 0; x < K; x++) fa[x] = (x < a.W) ? 0u : 0xFFFFFFFFu;
            uint32_t nsp = 0;
            const uint32_t b = tid * seg_len, e = min(N, b + seg_len);
            for (uint32_t m = b; m < e; m++) {
                uint32_t l, sink, fence;
                if (in_lds) {
                    const uint32_t cd = codeL[m];
                    l = cd & 0xFFFFu;
                    sink = cd & (1u << 30);
                    fence = cd & (1u << 31);
                } else {
                    const uint32_t z = rec[m].z;
                    l = last[m];
                    sink = z & kRecSink;
                    fence = z & kRecFence;
                }
                uint32_t wv = kRowNone;
                if (!sink && l != m + 1) {  // (a row only the next row reads is handed over in registers)
                    // (a row with a successor beyond kFarLds, or in a later segment, is always a spill row)
                    const bool may_slot = !fence && l < e;
                    uint32_t slot = K;
#pragma unroll
                    for (int x = K - 1; x >= 0; x--) slot = (may_slot && fa[x] <= m) ? (uint32_t)x : slot;
#pragma unroll
                    for (int x = 0; x < K; x++) fa[x] = (slot == (uint32_t)x) ? l : fa[x];
                    wv = slot < (uint32_t)K ? slot : (kRowSpilled | nsp++);
                }
                if (in_lds) codeL[m] = wv;
                else rec[m].w = wv;
            }
            mb[tid].id = nsp;  // (s_ids is free by now: spill rows of my segment)
        };
        if (a.W <= 4) walk(std::integral_constant<int, 4>());
        else walk(std::integral_constant<int, 8>());
    }
    __syncthreads();
    {   // spill rows get their final numbers: segment base + number within the segment
        uint32_t tot = 0;
        for (uint32_t g = 0; g < n_seg; g++) tot += mb[g].id;
        for (uint32_t i = tid; i < N; i += kGT) {
            uint32_t w = in_lds ? codeL[i] : rec[i].w;
            if (w != kRowNone && (w & kRowSpilled)) {
                uint32_t base = 0;
                for (uint32_t g = 0; g < i / seg_len; g++) base += mb[g].id;
                w += base;
            }
            rec[i].w = w;
            if (in_lds) codeL[i] = w;  // (read again by step 8)
        }
        if (tid == 0) {
            sz[0] = N;
            sz[1] = E;
            sz[2] = tot;
            sz[3] = (tot > kMaxSpillRows) ? 4u : 0u;
            sz[4] = s_tmp[kGT / 64 + 1];
        }
    }
    __syncthreads();
    GP(10)
    // 8. predecessor entries for the DP kernel: id | (LDS slot or spill row) << 16 | spilled << 31
    for (uint32_t i = tid; i < N; i += kGT) {
        const uint4 r = rec[i];
        const uint32_t np = r.z & 0xFFu;
        uint32_t first_far = 0;
        uint32_t dist = np ? 0u : kRecDistFar;  // (ids ascend: the first predecessor is the furthest)
        for (uint32_t x = 0; x < np; x++) {
            const uint32_t pa = pred[r.x + x];
            uint32_t pw = in_lds ? codeL[pa] : rec[pa].w;
            if (pw == kRowNone) pw = 0;  // (kept in registers for this row: the entry's slot is not read)
            const bool sp = (pw & kRowSpilled) != 0;
            if (sp && first_far == 0) first_far = x + 1;
            dist = max(dist, i - pa);
            pred[r.x + x] = pa | ((pw & 0x7FFFu) << 16) | (sp ? kPredSpilled : 0u);
        }
        rec[i].z = r.z | (first_far << 24) | (min(dist, kRecDistFar) << kRecDistShift);
    }
    GP(11)
    // 9. the DP kernel's row-skip bound (common.h "the bound", mesh_dp.hip PRUNE): R(m) = what a path can still gain
    // right of node m's column = the sum, over the occupied columns right of it, of the column's best node's gain
    // (an edge always leads to a column further right, a path takes at most one node per column).  Integer units:
    // the sums are exact and do not depend on the order of the scan.  The nodes of a column have consecutive ids:
    // the thread of a column's FIRST node finds the column's maximum; an exclusive scan of "maximum at the first
    // node, 0 elsewhere" gives every column the total of the columns left of it.
    if (a.reach != nullptr) {
        uint2 *rg = a.reach + (size_t)q * a.ncap;
        // (scratch: nobody reads succ_min in a launch that skips rows -- want_smin is --insertion=forbid; the debug
        // entry wants both and gets the last successors overwritten instead)
        // (in LDS where the rows' words fit the tile tables' space: step 8 was the last reader of the slot codes there)
        uint32_t *cw = in_lds ? codeL : (a.want_smin ? last : smin);
        if (tid == 0) s_tmp[kGT / 64 + 2] = 0xFFFFFFFFu;  // smallest column maximum
        __syncthreads();
        for (uint32_t i = tid; i < N; i += kGT) {
            const uint32_t pos = node_pos[i];
            if (i > 0 && node_pos[i - 1] == pos) continue;
            uint32_t mx = 0;
            for (uint32_t j = i; j < N && node_pos[j] == pos; j++) {
                mx = max(mx, prune_gain_units(__uint_as_float(rec[j].y), a.kappa64));
                cw[j] = 0;
            }
            cw[i] = mx;
            atomicMin(&s_tmp[kGT / 64 + 2], mx);
        }
        __syncthreads();
        if (tid == 0) sz[5] = s_tmp[kGT / 64 + 2];
        const uint32_t total = block_exscan(cw, cw, N, s_tmp);
        for (uint32_t i = tid; i < N; i += kGT) {
            const uint32_t pos = node_pos[i];
            if (i > 0 && node_pos[i - 1] == pos) continue;
            const uint32_t right = (i + 1 < N) ? total - cw[i + 1] : 0u;  // (cw[i + 1] = columns up to and including mine)
            const uint32_t cols_right = NC - 1u - rank(pos);                 // (the dense column index of step 2)
            for (uint32_t j = i; j < N && node_pos[j] == pos; j++) rg[j] = uint2{right, ((a.want_smin && !in_lds) ? 0u : last[j]) | (cols_right << 16)};
        }
    }
    GP(12)
    GP_FLUSH
}

// LDS of one workgroup: the tile tables, the bitmap + ranks (offset: graph_bitmap_off), the member records behind them
size_t graph_bitmap_off(uint32_t max_family) { return ((size_t)kOE + 2 * (size_t)max_family * kTC + 15) & ~(size_t)15; }
size_t graph_lds_bytes(uint32_t width, uint32_t max_family) {
    const size_t nwords = (width + 31) / 32;
    return graph_bitmap_off(max_family) + ((6 * nwords + 15) & ~(size_t)15);
}
size_t graph_member_cap(uint32_t max_family) { return std::max<size_t>(16, (max_family + 1) & ~(size_t)1); }
size_t graph_lds_total(uint32_t width, uint32_t max_family) {
    return graph_lds_bytes(width, max_family) + sizeof(Member) * graph_member_cap(max_family);
}

}  // namespace
}  // namespace sina_hip

using namespace sina_hip;

namespace {

struct BuiltGraphs {
    uint32_t ncap = 0;
    std::vector<uint64_t> pred_off;  // per query, into c->pred
    std::vector<uint32_t> sizes;     // per query: kSz words -- N, raw edge entries, n_spill, status, first sink row
};

// Builds the DAGs of bq families (fam_off is absolute, first family = q0) into the context's
// rec / node_pos / succ_minpos / pred buffers; grows the per-query caps and retries on overflow.
int build_family_graphs(sina_hip_ctx *c, const uint32_t *fam_ids, const uint64_t *fam_off, uint32_t q0,
                        uint32_t bq, float fs_weight, int W, BuiltGraphs *bg, bool want_smin, float kappa64) {
    hipStream_t s = c->stream;
    if (ensure_ref_off_host(c)) return 1;  // (a store that arrived by broadcast reads it back once)
    // weight table: the reference's expression (mseq.cpp:113) evaluated on the host
    if (!(c->wtab_fs_weight == fs_weight) || !c->g_wtab.p) {
        std::vector<float> wt((size_t)(kMaxFam + 1) * (kMaxFam + 1), 0.f);
        const float weight = fs_weight;
        for (unsigned F = 1; F <= (unsigned)kMaxFam; F++)
            for (unsigned cnt = 0; cnt <= F; cnt++)
                wt[(size_t)F * (kMaxFam + 1) + cnt] =
                    (float)(1.0 / (double)(weight + 1) + (double)(weight * ((float)cnt / (float)F)));
        if (c->g_wtab.reserve(wt.size() * 4)) return 1;
        SH_CHECK(hipMemcpyAsync(c->g_wtab.p, wt.data(), wt.size() * 4, hipMemcpyHostToDevice, s));
        SH_CHECK(hipStreamSynchronize(s));
        c->wtab_fs_weight = fs_weight;
    }
    std::vector<uint64_t> foff(bq + 1), elems(bq, 0);
    uint32_t max_f = 1;
    bg->pred_off.assign(bq, 0);
    for (uint32_t q = 0; q <= bq; q++) foff[q] = fam_off[q0 + q] - fam_off[q0];
    for (uint32_t q = 0; q < bq; q++) {
        max_f = std::max<uint32_t>(max_f, (uint32_t)(foff[q + 1] - foff[q]));
        for (uint64_t x = fam_off[q0 + q]; x < fam_off[q0 + q + 1]; x++) {
            const uint32_t id = fam_ids[x];
            if (id >= c->st->n_refs) SH_FAIL("align_families: reference id out of range");
            elems[q] += c->st->ref_off_host[id + 1] - c->st->ref_off_host[id];
        }
    }
    uint32_t ncap = std::min<uint32_t>(65535, 12288);
    for (int attempt = 0;; attempt++) {
        uint64_t pred_total = 0;
        for (uint32_t q = 0; q < bq; q++) {
            bg->pred_off[q] = pred_total;
            pred_total += elems[q] + 8;  // +8: slack behind every list
        }
        const size_t glds_tables = graph_lds_bytes(c->st->width, max_f);
        const size_t glds = graph_lds_total(c->st->width, max_f);
        if (glds > 160 * 1024) SH_FAIL("align_families: family too wide for the device DAG build");
        if (c->g_fam_ids.reserve(4 * std::max<uint64_t>(foff[bq], 1)) || c->g_fam_off.reserve(8 * ((uint64_t)bq + 1)) ||
            c->g_tmp1.reserve(8 * (uint64_t)bq) ||
            c->rec.reserve(sizeof(uint4) * (uint64_t)bq * ncap) || c->node_pos.reserve(4 * (uint64_t)bq * ncap) ||
            c->succ_minpos.reserve(4 * (uint64_t)bq * ncap) || c->g_tmp3.reserve(4 * (uint64_t)bq * ncap) ||
            c->pred.reserve(4 * pred_total) || c->g_sizes.reserve(4 * kSz * (uint64_t)bq) ||
            (kappa64 > 0.f && c->rgain.reserve(8 * (uint64_t)bq * ncap)))
            return 1;
        if (upload(c, 1, c->g_fam_ids.p, fam_ids + fam_off[q0], 4 * foff[bq], s) ||
            upload(c, 2, c->g_fam_off.p, foff.data(), 8 * ((uint64_t)bq + 1), s) ||
            upload(c, 3, c->g_tmp1.p, bg->pred_off.data(), 8 * (uint64_t)bq, s))
            return 1;
        GraphArgs ga;
        ga.ref_ab = c->st->ref_ab.as<uint32_t>();
        ga.ref_off = c->st->ref_off.as<uint64_t>();
        ga.fam_ids = c->g_fam_ids.as<uint32_t>();
        ga.fam_off = c->g_fam_off.as<uint64_t>();
        ga.pred_off = c->g_tmp1.as<uint64_t>();
        ga.wtab = c->g_wtab.as<float>();
        ga.rec = c->rec.as<uint4>();
        ga.node_pos = c->node_pos.as<uint32_t>();
        ga.succ_min = c->succ_minpos.as<uint32_t>();
        ga.far_mark = c->g_tmp3.as<uint32_t>();
        ga.pred = c->pred.as<uint32_t>();
        ga.sizes = c->g_sizes.as<uint32_t>();
        ga.width = c->st->width;
        ga.ncap = ncap;
        ga.bitmap_off = (uint32_t)graph_bitmap_off(max_f);
        ga.tile_bytes = ga.bitmap_off;
        ga.member_off = (uint32_t)glds_tables;
        ga.member_cap = (uint32_t)graph_member_cap(max_f);
        ga.fam_cap = max_f;
        ga.W = W;
        ga.want_smin = want_smin ? 1 : 0;
        ga.reach = kappa64 > 0.f ? c->rgain.as<uint2>() : nullptr;
        ga.kappa64 = kappa64;
        if (allow_full_lds(reinterpret_cast<const void *>(family_graph_kernel))) return 1;
        bg->sizes.resize(kSz * (size_t)bq);
        {
            heavy_launch hl(c, s, kHeavyGraph);  // (a device-filling kernel: ctx.h)
            SH_CHECK(hipEventRecord(c->ev[6], hl.stream()));
            // The launch behind a DAG build waits for its END: the build's own drain is a millisecond, and a DP launch
            // that starts in it begins with a stagger it carries to its own end (measured: 142.1 k sequences/s with
            // the signal, 143.1 k without, DP 49.5 against 48.3 ms per launch; profiles/r04_chain_ab.txt).
            // SINA_HIP_GRAPH_DRY=1 (experiments): the build tells its follower when its queue has run dry, like a DP launch.
            static const bool graph_dry = experiment_env("SINA_HIP_GRAPH_DRY") && experiment_env("SINA_HIP_GRAPH_DRY")[0] == '1';
            ga.dry = graph_dry ? hl.dry() : DryArgs{nullptr, nullptr, 0};
            hipLaunchKernelGGL(family_graph_kernel, dim3(bq), dim3(kGT), glds, hl.stream(), ga);
            SH_CHECK(hipGetLastError());
            SH_CHECK(hipEventRecord(c->ev[7], hl.stream()));
            if (hl.done()) return 1;
            if (experiment_env("SINA_HIP_DEBUG_SYNC")) fprintf(stderr, "[sina_hip] DAG build kernel done: %u families, ncap %u\n", bq, ncap);
        }
        if (download(c, 4, c->g_sizes.p, 4 * kSz * (uint64_t)bq, s)) return 1;
        SH_CHECK(wait_stream(c, s));
        memcpy(bg->sizes.data(), c->h_stage[4].p, 4 * kSz * (uint64_t)bq);
        float gms = 0;
        SH_CHECK(hipEventElapsedTime(&gms, c->ev[6], c->ev[7]));
        {
            uint64_t bytes = 0;  // algorithmic: the families' bases in, the DAGs out
            for (uint32_t q = 0; q < bq; q++)
                bytes += 4 * elems[q] + (uint64_t)bg->sizes[kSz * q] * (16 + 4 + (kappa64 > 0.f ? 8 : 0)) + 4 * (uint64_t)bg->sizes[kSz * q + 1];
            std::lock_guard<std::mutex> slk(c->st->stats_mu);
            c->st->stats.graph_ms += gms;
            c->st->stats.graph_bytes += bytes;
            c->st->stats.graph_launches++;
        }
        uint32_t need_n = 0;
        for (uint32_t q = 0; q < bq; q++) {
            if (bg->sizes[kSz * q + 3] == 2) need_n = std::max(need_n, bg->sizes[kSz * q]);
            if (bg->sizes[kSz * q + 3] == 4) SH_FAIL("align_families: too many spill rows for one query");
        }
        if (!need_n) break;
        if (attempt >= 3 || need_n > 65535u) SH_FAIL("align_families: family DAG exceeds device limits");
        ncap = std::min<uint32_t>(65535, need_n + need_n / 8 + 16);
    }
    bg->ncap = ncap;
    return 0;
}

}  // namespace

extern "C" {

int sina_hip_align_families(sina_hip_ctx *c, const uint32_t *fam_ids, const uint64_t *fam_off, uint32_t nq,
                            const uint8_t *qmask, const uint64_t *qoff, const sina_hip_align_params *p,
                            sina_hip_align_out *out, uint32_t *out_pos) {
    if (!c || !fam_ids || !fam_off || !qmask || !qoff || !p || !out)
        SH_FAIL("align_families: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    sina_hip_hint_guard hints(c);
    if (!c->st->have_refs) SH_FAIL("align_families: upload references first");
    if (nq == 0) return 0;
    SH_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (c->st->width > 524288u) SH_FAIL("align_families: alignment wider than 524288 columns (use align_graphs)");
    uint32_t maxL = 0;
    for (uint32_t q = 0; q < nq; q++) {
        const uint64_t L = qoff[q + 1] - qoff[q], F = fam_off[q + 1] - fam_off[q];
        if (L == 0 || L > 65535) SH_FAIL("align_families: query length must be in 1..65535");
        if (F == 0 || F > (uint64_t)kMaxFam) SH_FAIL("align_families: family size must be in 1..128");
        maxL = std::max<uint32_t>(maxL, (uint32_t)L);
    }
    DpPlan pl;
    if (plan_dp(c, maxL, &pl)) return 1;
    const int Lp = pl.geom.Lp();
    if (upload_weights(c, p)) return 1;
    if (c->h_out_pos.reserve(4 * std::max<uint64_t>(qoff[nq] - qoff[0], 1))) return 1;

    const uint64_t tb_budget_cells = tb_plane_budget(c) / tb_cell_bytes(p->insertion == SINA_INSERTION_FORBID);
    // queries per DAG build and DP launch: up to three rounds of DP wave slots (one DP wave per query) -- a DP
    // launch ends with ~4.4 ms of draining device whatever its size, so a third round makes it 3 % faster per
    // query than two (a fourth adds 2 % and another 22 GB per trace-back plane; SINA_HIP_DP_ROUNDS); the DP
    // launches below are whole rounds where the trace-back budget cuts a chunk
    const uint32_t slots = dp_wave_slots(c, pl.geom.B);
    static const uint32_t rounds = experiment_env("SINA_HIP_DP_ROUNDS") ? (uint32_t)std::max(1, atoi(experiment_env("SINA_HIP_DP_ROUNDS"))) : 3u;
    const uint32_t chunk_q = rounds * slots;
    BuiltGraphs bg;
    std::vector<uint32_t> dag_of;      // per query of the chunk: which of the chunk's distinct DAGs is its family's
    std::vector<uint32_t> ufam_ids;    // the distinct families, concatenated
    std::vector<uint64_t> ufam_off;
    for (uint32_t q0 = 0; q0 < nq; q0 += chunk_q) {
        const uint32_t bq = std::min(chunk_q, nq - q0);
        // Queries with the same ORDERED family share one DAG (node order, weights, predecessor lists and the DP's
        // row-slot assignment depend on nothing else): amplicons of one region against one reference clade.  The
        // DAG is built once per distinct family of the chunk; every query keeps its own trace-back cells, spill
        // rows and edge records.  (SINA_HIP_SHARE_DAGS=0: one build per query.)
        static const bool share = !(experiment_env("SINA_HIP_SHARE_DAGS") && experiment_env("SINA_HIP_SHARE_DAGS")[0] == '0');
        dag_of.assign(bq, 0);
        uint32_t n_dags = bq;
        const uint32_t *b_ids = fam_ids;
        const uint64_t *b_off = fam_off;
        uint32_t b_q0 = q0;
        if (share && bq > 1) {
            auto fam_hash = [&](uint32_t q) {
                uint64_t h = 0xcbf29ce484222325ull ^ (fam_off[q + 1] - fam_off[q]);
                for (uint64_t x = fam_off[q]; x < fam_off[q + 1]; x++) {
                    h = (h ^ fam_ids[x]) * 0x100000001b3ull;
                    h ^= h >> 31;
                }
                return h;
            };
            size_t cap = 16;
            while (cap < 2 * (size_t)bq) cap <<= 1;
            std::vector<uint32_t> slot(cap, 0xFFFFFFFFu), first;  // first[u] = first query (in the chunk) of DAG u
            std::vector<uint64_t> hq(bq);
            for (uint32_t q = 0; q < bq; q++) {
                hq[q] = fam_hash(q0 + q);
                size_t at = (size_t)(hq[q] >> 17) & (cap - 1);
                for (;;) {
                    const uint32_t u = slot[at];
                    if (u == 0xFFFFFFFFu) {
                        slot[at] = (uint32_t)first.size();
                        dag_of[q] = (uint32_t)first.size();
                        first.push_back(q);
                        break;
                    }
                    const uint32_t f = first[u];
                    const uint64_t la = fam_off[q0 + q + 1] - fam_off[q0 + q], lb = fam_off[q0 + f + 1] - fam_off[q0 + f];
                    if (hq[f] == hq[q] && la == lb && memcmp(fam_ids + fam_off[q0 + q], fam_ids + fam_off[q0 + f], 4 * la) == 0) {
                        dag_of[q] = u;
                        break;
                    }
                    at = (at + 1) & (cap - 1);
                }
            }
            n_dags = (uint32_t)first.size();
            if (n_dags < bq) {  // the distinct families, packed for the build
                ufam_off.assign((size_t)n_dags + 1, 0);
                for (uint32_t u = 0; u < n_dags; u++)
                    ufam_off[u + 1] = ufam_off[u] + (fam_off[q0 + first[u] + 1] - fam_off[q0 + first[u]]);
                ufam_ids.resize(ufam_off[n_dags]);
                for (uint32_t u = 0; u < n_dags; u++)
                    memcpy(ufam_ids.data() + ufam_off[u], fam_ids + fam_off[q0 + first[u]], 4 * (ufam_off[u + 1] - ufam_off[u]));
                b_ids = ufam_ids.data();
                b_off = ufam_off.data();
                b_q0 = 0;
            } else {
                for (uint32_t q = 0; q < bq; q++) dag_of[q] = q;
            }
        } else {
            for (uint32_t q = 0; q < bq; q++) dag_of[q] = q;
        }
        // (certified row skip of the DP kernel: the DAG build adds every node's bound on the gain still to come)
        PrunePlan pp = prune_plan(p, (float)(1.0 / (double)(p->fs_weight + 1) + (double)p->fs_weight), p->fs_weight >= 0.f ? 0.f : -1.f, maxL, false);
        if (build_family_graphs(c, b_ids, b_off, b_q0, n_dags, p->fs_weight, pl.W, &bg, p->insertion == SINA_INSERTION_FORBID,
                                pp.on ? pp.kappa64 : 0.f))
            return 1;
        {
            std::lock_guard<std::mutex> slk(c->st->stats_mu);
            c->st->stats.dags_built += n_dags;
            c->st->stats.dags_used += bq;
        }
        // DP in sub-ranges that fit the trace-back budget
        uint32_t r0 = 0;
        while (r0 < bq) {
            uint32_t r1 = r0;
            uint64_t tbc = 0, sprows = 0, cells = 0;
            uint32_t erec_cursor = 0;
            std::vector<QDesc> qd;
            std::vector<uint32_t> chain_ref;  // per query: its family's first member (the scout's chain, scout.hip)
            while (r1 < bq) {
                const uint32_t u = dag_of[r1];  // (this query's DAG among the chunk's distinct ones)
                const uint32_t N = bg.sizes[kSz * u];
                if (r1 > r0 && tbc + (uint64_t)N * Lp > tb_budget_cells) break;
                QDesc d;
                d.node_off = (uint64_t)u * bg.ncap;
                d.edge_off = bg.pred_off[u];
                d.q_off = qoff[q0 + r1] - qoff[q0 + r0];
                d.tb_off = tbc;
                d.spill_off = sprows;
                d.N = N;
                d.L = (uint32_t)(qoff[q0 + r1 + 1] - qoff[q0 + r1]);
                d.n_spill = bg.sizes[kSz * u + 2];
                d.first_sink = bg.sizes[kSz * u + 4];
                d.gmin = bg.sizes[kSz * u + 5];
                d.erec_off = erec_cursor;
                erec_cursor += dp_edge_entries(N);
                qd.push_back(d);
                chain_ref.push_back(fam_ids[fam_off[q0 + r1]]);
                tbc += (uint64_t)N * Lp;
                sprows += d.n_spill;
                cells += (uint64_t)N * d.L;
                r1++;
            }
            {
                const uint32_t r1r = dp_round_range(r0, r1, bq, slots);
                for (uint32_t r = r1r; r < r1; r++) {  // (the queries handed back to the next launch)
                    tbc -= (uint64_t)qd[r - r0].N * Lp;
                    sprows -= qd[r - r0].n_spill;
                    cells -= (uint64_t)qd[r - r0].N * qd[r - r0].L;
                }
                qd.resize(r1r - r0);
                chain_ref.resize(r1r - r0);
                r1 = r1r;
            }
            const uint32_t rq = r1 - r0;
            const uint64_t qbase = qoff[q0 + r0], nqm = qoff[q0 + r1] - qbase;
            if (c->qd.reserve(sizeof(QDesc) * rq) || c->qmask.reserve(std::max<uint64_t>(nqm, 1))) return 1;
            if (upload(c, 5, c->qd.p, qd.data(), sizeof(QDesc) * rq, s) || upload(c, 6, c->qmask.p, qmask + qbase, nqm, s))
                return 1;
            c->profile_batch = false;  // (device-built DAGs: never a profile)
            c->out_pos_base = qbase - qoff[0];
            if (run_dp_device(c, pl, qd.data(), rq, (uint64_t)n_dags * bg.ncap, tbc, sprows, cells, nqm, p, c->st->width,
                              out + q0 + r0, out_pos ? out_pos + qbase : nullptr, false, pp, chain_ref.data()))
                return 1;
            r0 = r1;
        }
    }
    return 0;
}

#ifdef SINA_DP_PROFILE
int sina_hip_debug_graph_profile(unsigned long long *out16, int reset) {
    SH_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_graph_prof), sizeof(unsigned long long) * 16));
    if (reset) {
        unsigned long long z[16] = {};
        SH_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_graph_prof), z, sizeof z));
    }
    return 0;
}
#endif

int sina_hip_debug_family_graph(sina_hip_ctx *c, const uint32_t *fam_ids, uint32_t F, float fs_weight,
                                uint32_t ring_depth, uint32_t *n_nodes, uint32_t *n_edges, uint32_t *pos,
                                uint8_t *mask, float *weight, uint32_t *pred_off, uint32_t *pred,
                                uint32_t *succ_minpos, uint8_t *sink, uint32_t *spill_idx, uint32_t cap_nodes,
                                uint32_t cap_edges) {
    if (!c || !fam_ids || !n_nodes || !n_edges) SH_FAIL("debug_family_graph: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->st->have_refs) SH_FAIL("debug_family_graph: upload references first");
    if (F == 0 || F > (uint32_t)kMaxFam) SH_FAIL("debug_family_graph: family size must be in 1..128");
    SH_CHECK(hipSetDevice(c->device));
    const uint64_t foff[2] = {0, F};
    BuiltGraphs bg;
    // (with the row-skip bound for the default scoring: sina_hip_debug_rgain reads it back)
    if (build_family_graphs(c, fam_ids, foff, 0, 1, fs_weight, (int)ring_depth, &bg, true, fs_weight >= 0.f ? 64.0f * 1.0001f * 2.0f : 0.f)) return 1;
    const uint32_t N = bg.sizes[0];
    std::vector<uint4> rec(N);
    std::vector<uint32_t> pr(bg.sizes[1] + 8);
    SH_CHECK(hipMemcpy(rec.data(), c->rec.p, sizeof(uint4) * N, hipMemcpyDeviceToHost));
    SH_CHECK(hipMemcpy(pr.data(), c->pred.p, 4 * pr.size(), hipMemcpyDeviceToHost));
    uint32_t E = 0;
    for (uint32_t m = 0; m < N; m++) E += rec[m].z & 0xffu;
    *n_nodes = N;
    *n_edges = E;
    if (N > cap_nodes || E > cap_edges) SH_FAIL("debug_family_graph: output buffers too small");
    SH_CHECK(hipMemcpy(pos, c->node_pos.p, 4 * N, hipMemcpyDeviceToHost));
    SH_CHECK(hipMemcpy(succ_minpos, c->succ_minpos.p, 4 * N, hipMemcpyDeviceToHost));
    uint32_t e = 0;
    for (uint32_t m = 0; m < N; m++) {
        mask[m] = (uint8_t)((rec[m].z >> 8) & 0xffu);
        memcpy(&weight[m], &rec[m].y, 4);
        sink[m] = (rec[m].z & kRecSink) ? 1 : 0;
        spill_idx[m] = rec[m].w;
        pred_off[m] = e;
        for (uint32_t x = 0; x < (rec[m].z & 0xffu); x++) pred[e++] = pr[rec[m].x + x] & 0xffffu;
    }
    pred_off[N] = e;
    return 0;
}

}  // extern "C"
