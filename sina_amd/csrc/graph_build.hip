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
// kTC occupied columns so that the [column][family member] tables stay in LDS:
//   1. occupied-column bitmap in LDS (atomicOr per base), prefix-popcount -> dense
//      column index ("rank") of every alignment column;
//   2. per tile: every member's bases that fall into the tile (a contiguous stretch of its
//      sequence, found with a per-member cursor) drop their mask into tabm[column][member];
//   3. one thread per column scans its F entries in family order: first appearance of
//      a mask opens a node (exact reference order), later ones count; block scans
//      turn per-column node / raw-edge counts into node ids and CSR segments (running totals
//      carry across tiles);
//   4. per base again: the node of the member's PREVIOUS base (same tile: table lookup; earlier
//      tile: carried per member) goes into tabp[column][member];
//   5. one thread per column emits node records and, per node, inserts the
//      predecessor ids into a sorted unique list in its CSR segment; atomicMin / atomicMax
//      collect the successor minimum column (for --insertion=forbid) and the last successor;
//   6. sinks and fence flags; LDS-slot / spill-row assignment for the DP kernel by liveness
//      (sequential over the rows: one wave, bookkeeping on the scalar unit); predecessor entries.
// HBM traffic: the family's bases three times (bitmap, masks, previous nodes) and the DAG once.
// All arithmetic is integer except the node weight, which is looked up in a table the
// HOST computed with the reference's own mixed double/float expression.
#include <algorithm>
#include <cstring>

#include "common.h"
#include "ctx.h"

namespace sina_hip {
namespace {

#ifndef SINA_GRAPH_THREADS
#define SINA_GRAPH_THREADS 512
#endif
#ifndef SINA_GRAPH_MINWAVES
#define SINA_GRAPH_MINWAVES 6  // (3 workgroups of 8 waves per CU: <= 80 VGPRs)
#endif
constexpr int kGT = SINA_GRAPH_THREADS;  // threads per workgroup (the phases are latency-bound: more loads in flight per LDS byte)
constexpr uint32_t kNoPrev = 0xFFFFu;
constexpr int kMaxFam = 128;
constexpr int kSz = 8;  // u32 words the kernel reports per query (GraphArgs::sizes)
#ifndef SINA_GRAPH_KTC
#define SINA_GRAPH_KTC 160  // (measured, 3072 families of 40: 96 -> 3.05 ms, 128 -> 2.54, 160 -> 2.40, 192 -> 3.12: a third workgroup per CU no longer fits)
#endif
constexpr int kTC = SINA_GRAPH_KTC;  // occupied columns per LDS tile (< 255: tile columns are bytes, 255 = none)

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
    uint32_t tile_bytes;       // LDS bytes of the tile tables (reused by the slot allocation)
    uint32_t member_off;       // LDS offset of the per-member arrays (behind the tile tables), entries each
    uint32_t member_cap;
    int W;                     // DP ring depth: edges longer than this need a spill row
    int want_smin;             // succ_min is read by somebody (--insertion=forbid, the debug entry): else it is not touched at all
    uint2 *reach;              // [nq][ncap] or nullptr: the DP kernel's row-skip bound (step 9; units: common.h): {R(m), last successor | C(m) << 16}
    float kappa64;             // ... 64 * 1.0001 * (largest match gain per unit of node weight)
    DryArgs dry;               // (ctx.h, heavy_launch: tells the launch queued behind when the last workgroup has started)
};

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

__global__ void __launch_bounds__(kGT, SINA_GRAPH_MINWAVES) family_graph_kernel(GraphArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // per family member (sized by the launch's largest family, at least 16: s_ids doubles as the
    // slot allocator's per-segment counters) -- dynamic, so that small families leave the LDS to a
    // fourth workgroup per CU
    uint64_t *s_beg = reinterpret_cast<uint64_t *>(smem + a.member_off);
    uint32_t *s_ids = reinterpret_cast<uint32_t *>(s_beg + a.member_cap);
    uint32_t *s_len = s_ids + a.member_cap;
    uint32_t *s_cur = s_len + a.member_cap, *s_curn = s_cur + a.member_cap;      // first base of member j at/after the tile
    uint32_t *s_carry = s_curn + a.member_cap, *s_carryn = s_carry + a.member_cap;  // node of member j's last base before the tile
    __shared__ uint32_t s_tmp[kGT / 64 + 8];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    dry_signal(a.dry, gridDim.x, tid == 0);
    const uint32_t nwords = (a.width + 31) / 32;
    const uint64_t f0 = a.fam_off[q];
    const uint32_t F = (uint32_t)(a.fam_off[q + 1] - f0);
    const uint32_t FS = (F + 1) | 1u;  // table row stride (u16 units)
    // LDS carve
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(smem);                 // [nwords]
    uint16_t *wrank = reinterpret_cast<uint16_t *>(bitmap + nwords);        // [nwords]
    unsigned char *tile = smem + ((6 * (size_t)nwords + 15) & ~(size_t)15);
    uint32_t *cposT = reinterpret_cast<uint32_t *>(tile);                   // [kTC] alignment column
    uint32_t *nbaseT = cposT + kTC;                                         // [kTC] first node id
    uint32_t *ebaseT = nbaseT + kTC;                                        // [kTC] first raw edge slot
    uint8_t *nn = reinterpret_cast<uint8_t *>(ebaseT + kTC);                // [kTC] nodes per column
    uint8_t *rc = nn + kTC;                                                 // [kTC] raw edges per column
    // tabm[c][j]: mask (8) | local node index (5) << 8 | first base of the member << 13 ; 0 = absent
    uint16_t *tabm = reinterpret_cast<uint16_t *>(rc + kTC);                // [kTC][FS]
    uint16_t *tabp = tabm + (size_t)kTC * FS;                               // [kTC][FS] previous node, 0xFFFF none
    uint8_t *ncolT = reinterpret_cast<uint8_t *>(tabp + (size_t)kTC * FS);  // [kTC * 32] tile column of every node of the tile
    // [F][kTC] tile column of member j's (s_cur[j] + slot)-th base, 255: not in the tile -- the same
    // LDS as ncolT: cl8 is read for the last time in step 4, ncolT is filled after it
    uint8_t *cl8 = ncolT;

    uint32_t *sz = a.sizes + kSz * (size_t)q;
    for (uint32_t j = tid; j < F; j += kGT) {
        const uint32_t id = a.fam_ids[f0 + j];
        s_ids[j] = id;
        s_beg[j] = a.ref_off[id];
        s_len[j] = (uint32_t)(a.ref_off[id + 1] - a.ref_off[id]);
        s_cur[j] = s_curn[j] = 0;
        s_carry[j] = s_carryn[j] = kNoPrev;
    }
    for (uint32_t i = tid; i < nwords; i += kGT) bitmap[i] = 0;
    __syncthreads();
    GP_DECL

    // 1. occupied columns
    // (latency-bound: four members' loads are in flight per thread before the first is used)
    for (uint32_t j0 = 0; j0 < F; j0 += 4) {
        const uint32_t maxlen = max(max(s_len[j0], j0 + 1 < F ? s_len[j0 + 1] : 0u),
                                    max(j0 + 2 < F ? s_len[j0 + 2] : 0u, j0 + 3 < F ? s_len[j0 + 3] : 0u));
        for (uint32_t i = tid; i < maxlen; i += kGT) {
            uint32_t ab[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                ab[u] = (j0 + u < F && i < s_len[j0 + u]) ? a.ref_ab[s_beg[j0 + u] + i] : 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < 4; u++)
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
        {   // clear the mask table (u32 stores; FS odd, kTC even: the element count is even)
            uint32_t *z = reinterpret_cast<uint32_t *>(tabm);
            for (uint32_t i = tid; i < (uint32_t)kTC * FS / 2; i += kGT) z[i] = 0;
        }
        __syncthreads();
        GP(2)
        // 3a. masks of this tile: member j's bases cur[j].. as long as their column is in the tile (a
        // contiguous stretch of at most kTC bases); the tile column of every such base is kept in
        // cl8 so that nothing below has to rank a position again
        for (uint32_t idx0 = tid; idx0 < F * (uint32_t)kTC; idx0 += 4 * kGT) {
            // (latency-bound: four loads in flight per thread)
            uint32_t abv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t idx = idx0 + (uint32_t)u * kGT;
                abv[u] = 0xFFFFFFFFu;  // (no packed base looks like this: the mask byte has five bits)
                if (idx < F * (uint32_t)kTC) {
                    const uint32_t j = idx / kTC, i = s_cur[j] + idx % kTC;
                    if (i < s_len[j]) abv[u] = a.ref_ab[s_beg[j] + i];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t idx = idx0 + (uint32_t)u * kGT;
                if (idx >= F * (uint32_t)kTC) continue;
                const uint32_t j = idx / kTC, slot = idx % kTC, i = s_cur[j] + slot;
                uint32_t cl = 255;
                if (abv[u] != 0xFFFFFFFFu) {
                    const uint32_t ab = abv[u];
                    const uint32_t pos = ab & 0xFFFFFFu;
                    const uint32_t c = rank(pos) - c0;
                    if (c < tc) {
                        cl = c;
                        tabm[c * FS + j] = (uint16_t)(((ab >> 24) & 0xFFu) | (i == 0 ? (1u << 13) : 0u));
                        cposT[c] = pos;
                    }
                }
                cl8[j * kTC + slot] = (uint8_t)cl;
            }
        }
        __syncthreads();
        // the member's last base in this tile moves its cursor
        for (uint32_t idx = tid; idx < F * (uint32_t)kTC; idx += kGT) {
            const uint32_t j = idx / kTC, slot = idx % kTC;
            if (cl8[j * kTC + slot] != 255 && (slot + 1 == (uint32_t)kTC || cl8[j * kTC + slot + 1] == 255))
                s_curn[j] = s_cur[j] + slot + 1;
        }
        __syncthreads();
        GP(3)
        // 3b. per column: nodes in order of first appearance (family order), raw edge count
        for (uint32_t c = tid; c < tc; c += kGT) {
            uint32_t seen = 0;          // bit m: mask value m already has a node in this column
            uint64_t idx0 = 0, idx1 = 0, idx2 = 0;  // local node index of mask m, 5 bits each (12/12/8 masks)
            uint32_t k = 0, raw = 0;
            uint16_t *row = tabm + c * FS;
            for (uint32_t j0 = 0; j0 < F; j0 += 8) {  // (eight LDS reads in flight, as in step 5)
                uint32_t tv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) tv[u] = (j0 + u < F) ? (uint32_t)row[j0 + u] : 0u;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t t = tv[u];
                    const uint32_t m = t & 0x1Fu;
                    if ((t & 0xFFu) == 0) continue;
                    uint32_t li;
                    if (!((seen >> m) & 1u)) {
                        seen |= 1u << m;
                        li = k++;
                        if (m < 12) idx0 |= (uint64_t)li << (5 * m);
                        else if (m < 24) idx1 |= (uint64_t)li << (5 * (m - 12));
                        else idx2 |= (uint64_t)li << (5 * (m - 24));
                    } else {
                        li = (m < 12) ? (uint32_t)(idx0 >> (5 * m)) & 31u
                             : (m < 24) ? (uint32_t)(idx1 >> (5 * (m - 12))) & 31u
                                        : (uint32_t)(idx2 >> (5 * (m - 24))) & 31u;
                    }
                    row[j0 + u] = (uint16_t)(t | (li << 8));
                    if (!(t & (1u << 13))) raw++;
                }
            }
            nn[c] = (uint8_t)k;
            rc[c] = (uint8_t)raw;
        }
        __syncthreads();
        GP(4)
        const uint32_t tn = block_exscan(nn, nbaseT, tc, s_tmp);
        const uint32_t te = block_exscan(rc, ebaseT, tc, s_tmp);
        GP(5)
        // 4. node of every base's predecessor base (the member's previous base: one slot to the left,
        // or -- for its first base in the tile -- what the previous tiles carried over)
        for (uint32_t idx = tid; idx < F * (uint32_t)kTC; idx += kGT) {
            const uint32_t j = idx / kTC, slot = idx % kTC;
            const uint32_t cl = cl8[j * kTC + slot];
            if (cl == 255) continue;
            const uint32_t i = s_cur[j] + slot;
            const uint32_t node = N + nbaseT[cl] + ((tabm[cl * FS + j] >> 8) & 31u);
            uint32_t pn = kNoPrev;
            if (slot > 0) {
                const uint32_t pl = cl8[j * kTC + slot - 1];
                pn = N + nbaseT[pl] + ((tabm[pl * FS + j] >> 8) & 31u);
            } else if (i > 0) {
                pn = s_carry[j];
            }
            tabp[cl * FS + j] = (uint16_t)pn;
            if (i + 1 == s_curn[j]) s_carryn[j] = node;
        }
        __syncthreads();
        for (uint32_t c = tid; c < tc; c += kGT)  // (cl8 is dead: its LDS becomes the node -> column map)
            for (uint32_t k = 0; k < nn[c]; k++) ncolT[nbaseT[c] + k] = (uint8_t)c;
        for (uint32_t ln = tid; ln < tn; ln += kGT)  // this tile's nodes: no successor seen yet
            if (N + ln < a.ncap) {
                last[N + ln] = 0;
                if (a.want_smin) smin[N + ln] = 0xFFFFFFFFu;
            }
        __syncthreads();
        GP(6)
        // 5. node records + sorted unique predecessor lists: one thread per NODE of the tile (a column
        // has 1.7 nodes on average and up to five or so: threads per column would wait for the widest
        // column of their wave).  One pass over the column's F members gives the node its member
        // count, mask, the raw edge entries of the column's earlier nodes (= where its CSR segment
        // starts) and its predecessors.
        for (uint32_t ln = tid; ln < tn; ln += kGT) {
            const uint32_t c = ncolT[ln];
            const uint32_t k = ln - nbaseT[c];
            const uint16_t *rowm = tabm + c * FS;
            const uint16_t *rowp = tabp + c * FS;
            const uint32_t pos = cposT[c];
            const uint32_t node = N + ln;
            uint32_t cnt = 0, np = 0, rawk = 0, raw_before = 0, mask = 0;
            // sorted unique predecessor ids, first in registers (8 cover all but freak
            // columns): no read-modify-write round trips to the CSR segment in HBM
            uint32_t pl[8];
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) pl[t8] = 0xFFFFFFFFu;
            bool overflow = false;
            uint32_t recent = 0xFFFFFFFFu;  // (most members of a node come from the same previous node)
            // (the column's entries eight at a time: the LDS reads of a batch are in flight together
            // instead of one round trip per family member)
            for (uint32_t j0 = 0; j0 < F; j0 += 8) {
                uint32_t tv[8], pv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const bool in = j0 + u < F;
                    tv[u] = in ? (uint32_t)rowm[j0 + u] : 0u;
                    pv[u] = in ? (uint32_t)rowp[j0 + u] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t t = tv[u];
                    if ((t & 0xFFu) == 0) continue;
                    const uint32_t lj = (t >> 8) & 31u;
                    const bool has_prev = !(t & (1u << 13));
                    if (lj < k) raw_before += has_prev ? 1u : 0u;
                    if (lj != k) continue;
                    mask = t & 0xFFu;
                    cnt++;
                    if (!has_prev) continue;
                    rawk++;
                    uint32_t x = pv[u];
                    if (x == recent) continue;
                    recent = x;
#pragma unroll
                    for (int t8 = 0; t8 < 8; t8++) {  // bubble x into place; a duplicate turns into the pad value
                        const uint32_t y = pl[t8];
                        if (x == y) x = 0xFFFFFFFFu;
                        const bool sw = x < y;
                        pl[t8] = sw ? x : y;
                        x = sw ? y : x;
                    }
                    overflow = overflow || (x != 0xFFFFFFFFu);
                }
            }
            const uint32_t seg = E + ebaseT[c] + raw_before;
            if (node < a.ncap) {
                if (!overflow) {
#pragma unroll
                    for (int t8 = 0; t8 < 8; t8++) {
                        const uint32_t pa = pl[t8];
                        if (pa != 0xFFFFFFFFu) {
                            pred[seg + np] = pa;
                            np++;
                            if (a.want_smin) atomicMin(&smin[pa], pos);
                            atomicMax(&last[pa], node);
                        }
                    }
                } else {  // more than 8 distinct predecessors: insertion sort in the segment itself
                    for (uint32_t j = 0; j < F; j++) {
                        const uint32_t t = rowm[j];
                        if ((t & 0xFFu) == 0 || ((t >> 8) & 31u) != k || (t & (1u << 13))) continue;
                        const uint32_t pa = rowp[j];
                        uint32_t x = 0;
                        while (x < np && pred[seg + x] < pa) x++;
                        if (x < np && pred[seg + x] == pa) continue;
                        for (uint32_t y = np; y > x; y--) pred[seg + y] = pred[seg + y - 1];
                        pred[seg + x] = pa;
                        np++;
                        if (a.want_smin) atomicMin(&smin[pa], pos);
                        atomicMax(&last[pa], node);
                    }
                }
                uint4 r;
                r.x = seg;
                r.y = __float_as_uint(wt[cnt]);
                r.z = (np & 0xFFu) | (mask << 8);
                r.w = kRowNone;
                rec[node] = r;
                node_pos[node] = pos;
            }
            (void)rawk;
        }
        __syncthreads();
        for (uint32_t j = tid; j < F; j += kGT) {
            s_cur[j] = s_curn[j];
            s_carry[j] = s_carryn[j];
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
    uint32_t *codeL = reinterpret_cast<uint32_t *>(tile);
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
        uint32_t fa[8];  // last successor of the row in slot x (0: empty)
#pragma unroll
        for (int x = 0; x < 8; x++) fa[x] = (x < a.W) ? 0u : 0xFFFFFFFFu;
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
                uint32_t slot = 8;
#pragma unroll
                for (int x = 7; x >= 0; x--) slot = (may_slot && fa[x] <= m) ? (uint32_t)x : slot;
#pragma unroll
                for (int x = 0; x < 8; x++) fa[x] = (slot == (uint32_t)x) ? l : fa[x];
                wv = slot < 8 ? slot : (kRowSpilled | nsp++);
            }
            if (in_lds) codeL[m] = wv;
            else rec[m].w = wv;
        }
        s_ids[tid] = nsp;  // (s_ids is free by now: spill rows of my segment)
    }
    __syncthreads();
    {   // spill rows get their final numbers: segment base + number within the segment
        uint32_t tot = 0;
        for (uint32_t g = 0; g < n_seg; g++) tot += s_ids[g];
        for (uint32_t i = tid; i < N; i += kGT) {
            uint32_t w = in_lds ? codeL[i] : rec[i].w;
            if (w != kRowNone && (w & kRowSpilled)) {
                uint32_t base = 0;
                for (uint32_t g = 0; g < i / seg_len; g++) base += s_ids[g];
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

size_t graph_lds_bytes(uint32_t width, uint32_t max_family) {
    const size_t nwords = (width + 31) / 32;
    const size_t fs = (max_family + 1) | 1u;
    return ((6 * nwords + 15) & ~(size_t)15) + (size_t)kTC * (4 + 4 + 4 + 1 + 1) + 2 * 2 * (size_t)kTC * fs +
           (size_t)kTC * std::max<size_t>(32, fs) + 64;
}
// the per-member arrays behind that: offset (8-byte aligned) and total
size_t graph_member_cap(uint32_t max_family) { return std::max<size_t>(16, (max_family + 1) & ~(size_t)1); }
size_t graph_lds_total(uint32_t width, uint32_t max_family) {
    return ((graph_lds_bytes(width, max_family) + 7) & ~(size_t)7) + 32 * graph_member_cap(max_family);
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
        ga.tile_bytes = (uint32_t)(glds_tables - ((6 * (size_t)((c->st->width + 31) / 32) + 15) & ~(size_t)15) - 64);
        ga.member_off = (uint32_t)((glds_tables + 7) & ~(size_t)7);
        ga.member_cap = (uint32_t)graph_member_cap(max_f);
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
                              out + q0 + r0, out_pos ? out_pos + qbase : nullptr, false, pp))
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
