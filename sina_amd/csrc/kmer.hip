// K-mer reference search on gfx950: device index build, shared-k-mer counting,
// top-k selection.
//
// What it computes (reference src/kmer.h, src/idset.h, src/kmer_search.cpp:
// 152-276,366-420; spec in SURVEY.md Appendix A.1):
//   K(seq)  = k-mers over windows of k unambiguous bases ending at base index
//             e, k-1 <= e <= len-2 (the window ending on the last base is never
//             produced, kmer.h:188-201); "fast" keeps windows starting with A.
//   index   : reference r is in postings(v) iff v in set(K(r)); ids ascending.
//   score   = sum over K(query) WITH multiplicity of [r in postings(v)], int16.
//   top-k   : (score desc, id desc), i.e. std::greater<pair<int16,int>>.
//
// How it maps to the hardware: integer work on data that sits in L2 / Infinity Cache.
//   count : one 1024-thread workgroup per query.  Every query k-mer keeps a cursor into its
//           ascending posting list; the references are processed in tiles of 32768 whose int16
//           counters live in LDS (two per 32-bit word); a wave streams a list from its cursor with
//           2 x 1 KiB loads in flight into LDS atomics -- no search -- and the tile is written out
//           once.  Lists longer than 1/64 of the references are kept a second time as bitmaps and
//           counted bit-sliced in registers (carry-save adders, no atomics).
//   select: one workgroup per query: the cut score by an 8-way search on "how many scores are >= t"
//           (reductions over 16-byte loads; an LDS histogram serialises on the few low bins every
//           lane hits) -> ordered compaction (ties keep the LARGEST ids) -> bitonic sort of the
//           <= 4096 survivors.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "ctx.h"

namespace sina_hip {
namespace {

constexpr int kCountThreads = 1024;
constexpr int kTileRefs = 32768;            // refs per LDS histogram tile (64 KiB)
constexpr int kWide = 2;                    // 1 KiB loads a wave of the count kernel keeps in flight
constexpr int kMaxQueryLen = (int)SINA_HIP_MAX_QUERY_LEN;  // k-mer list capacity in LDS: 64 KiB tile + 9 B per base <= 160 KiB
static_assert((size_t)kTileRefs * 2 + ((size_t)kMaxQueryLen + 63) / 64 * 64 * 9 + 64 <= 160 * 1024, "LDS of the count kernel");
constexpr int kSelThreads = 256;
constexpr int kSelMax = 4096;               // candidates sortable in LDS

__device__ __forceinline__ bool ambig(uint32_t m) { return __popc(m & 0xfu) > 1; }

// k-mer ending at base e of a sequence of iupac masks (bases must be unambiguous
// and non-gap).  Returns false if the window is not a valid k-mer.
__device__ __forceinline__ bool kmer_at(const uint8_t *m, uint32_t len, uint32_t e, unsigned k, bool fast,
                                        uint32_t *out) {
    if (e + 1 < k || e + 2 > len) return false;  // need k bases and e <= len-2
    uint32_t v = 0;
    for (unsigned i = 0; i < k; i++) {
        const uint32_t b = m[e + 1 - k + i] & 0xfu;
        if (__popc(b) != 1) return false;
        v = (v << 2) | (uint32_t)(__ffs(b) - 1);
    }
    if (fast && (v >> (2 * (k - 1))) != 0) return false;  // prefix_filter(k, 1, BASE_A)
    *out = v;
    return true;
}

// ---------------------------------------------------------------- index build

// one (kmer << 32 | ref) key per valid window of every reference; invalid
// windows emit ~0 (sorted to the end, dropped).
__global__ void ref_kmer_keys(const uint32_t *ref_ab, const uint64_t *ref_off, uint32_t n_refs, unsigned k,
                              bool fast, uint64_t *keys) {
    const uint32_t r = blockIdx.x;
    const uint64_t b = ref_off[r], e = ref_off[r + 1];
    const uint32_t len = (uint32_t)(e - b);
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
        uint64_t key = ~0ull;
        if (i + 1 >= k && i + 2 <= len) {
            uint32_t v = 0;
            bool ok = true;
            for (unsigned x = 0; x < k; x++) {
                const uint32_t m = (ref_ab[b + i + 1 - k + x] >> 24) & 0xfu;
                if (__popc(m) != 1) {
                    ok = false;
                    break;
                }
                v = (v << 2) | (uint32_t)(__ffs(m) - 1);
            }
            if (ok && fast && (v >> (2 * (k - 1))) != 0) ok = false;
            if (ok) key = ((uint64_t)v << 32) | r;
        }
        keys[b + i] = key;
    }
}

// after sorting: flag first occurrence of each (kmer, ref) key
__global__ void mark_unique(const uint64_t *keys, uint64_t n, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t kx = keys[i];
    flag[i] = (kx != ~0ull && (i == 0 || keys[i - 1] != kx)) ? 1u : 0u;
}

__global__ void scatter_unique(const uint64_t *keys, const uint32_t *flag, const uint32_t *pos, uint64_t n,
                               uint32_t *ids, uint32_t *counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint64_t kx = keys[i];
    ids[pos[i]] = (uint32_t)kx;
    atomicAdd(&counts[(uint32_t)(kx >> 32)], 1u);
}

// ---------------------------------------------------------------- count

struct CountArgs {
    const uint8_t *qmask;
    const uint64_t *qoff;
    const uint32_t *idx_off;
    const uint32_t *idx_ids;
    int16_t *scores;  // [nq][stride]
    uint32_t *nkq;    // [nq] number of query k-mers (upper bound of any score)
    unsigned long long *postings;
    uint32_t n_refs, stride, kmax;
    unsigned k;
    int fast;
    // dense posting lists as bitmaps (see ensure_dense): bitmap number per k-mer or kNoDense, words per bitmap
    const uint32_t *dense_id;
    const uint32_t *dense_bits;
    uint32_t dense_words;
    // CAND (kmer_count_kernel<true>): instead of the score row, the references that can still be among the top
    // `topm` -- keys (score + 32768) << 32 | id, at most cand_cap per query; cand_n[q] = how many there were
    unsigned long long *cand;
    uint32_t *cand_n;
    uint32_t cand_cap, topm;
};

constexpr uint32_t kNoDense = 0xFFFFFFFFu;
constexpr uint32_t kMaxDenseQ = 1023;  // dense k-mers of one query counted bit-sliced in 10 planes
static_assert(kTileRefs / 32 == kCountThreads, "one bitmap word of the tile per thread");

// ---- dense lists: which k-mers, and their bitmaps
__global__ void mark_dense(const uint32_t *idx_off, uint32_t n_kmers, uint32_t thresh, uint32_t *dense_id,
                           uint32_t *counter) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_kmers) return;
    const uint32_t len = idx_off[v + 1] - idx_off[v];
    dense_id[v] = (len > thresh) ? atomicAdd(counter, 1u) : kNoDense;
}
__global__ void fill_dense(const uint32_t *idx_off, const uint32_t *idx_ids, const uint32_t *dense_id,
                           uint32_t n_kmers, uint32_t *bits, uint32_t words) {
    // one wave per k-mer (most leave at once)
    const uint32_t v = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (v >= n_kmers) return;
    const uint32_t d = dense_id[v];
    if (d == kNoDense) return;
    uint32_t *b = bits + (size_t)d * words;
    for (uint32_t x = idx_off[v] + lane; x < idx_off[v + 1]; x += 64) {
        const uint32_t r = idx_ids[x];
        atomicOr(&b[r >> 5], 1u << (r & 31));
    }
}

// One workgroup (16 waves) per query.  Every query k-mer keeps a cursor into its (ascending)
// posting list; the reference range is processed in tiles of kTileRefs whose int16 counters
// live in LDS (two per 32-bit word).  For a tile, a wave takes a k-mer, streams postings from
// its cursor (kWide loads of 1 KiB in flight, four consecutive postings per lane), bumps the LDS
// counters of those below the tile end (a prefix, the lists being sorted) and advances the
// cursor: there is no search, and the tile is written out once.  (64 VGPRs: two workgroups per CU.)
// CAND: the score row never leaves the chip.  Top-M by (score desc, id desc) only needs the references whose score
// reaches the M-th largest score -- and tile 0 alone already holds M references that reach t0 = the M-th largest of
// its 1024 per-thread maxima (each maximum is the score of a different reference), so the final cut is t0 or more:
// every tile hands the references with score >= t0 (a few dozen per tile; ties included) to a candidate list, and
// kmer_select_cand_kernel sorts that list.  Saves the row's write and the select kernel's two passes over it
// (2 + 4 bytes per reference and query: at 500 000 references 3 MB per query, and the 2 GiB score matrix that cut
// a 9216-query search into five launches).  More candidates than the list holds (a giant group of equal scores):
// cand_n says so and the host repeats the launch with the score rows.
template <bool CAND>
__global__ void __launch_bounds__(kCountThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) kmer_count_kernel(CountArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t n_kmers, next_kmer, n_dense_q;
    __shared__ uint32_t c_cnt[8], c_ncand;
    __shared__ int c_t0;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);                 // [kTileRefs/2]
    uint32_t *cur = hist + kTileRefs / 2;                                // [kmax]
    uint32_t *end = cur + a.kmax;                                        // [kmax]
    uint8_t *qb = reinterpret_cast<uint8_t *>(end + a.kmax);             // [kmax] query masks
    // bitmap numbers of the query's dense k-mers: from the top of cur[] downwards (cursor slots grow
    // from the bottom; together they are at most kmax k-mers) -- a third array would cost the second
    // workgroup per CU
    uint32_t *dtop = cur + a.kmax - 1;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63;
    const uint8_t *qm = a.qmask + a.qoff[q];
    const uint32_t len = (uint32_t)(a.qoff[q + 1] - a.qoff[q]);
    if (tid == 0) {
        n_kmers = n_dense_q = 0;
        c_ncand = 0;
        c_t0 = 0;
    }
    for (uint32_t i = tid; i < len; i += kCountThreads) qb[i] = qm[i];
    __syncthreads();
    unsigned long long mine = 0;
    for (uint32_t e = tid; e < len; e += kCountThreads) {
        uint32_t v;
        if (kmer_at(qb, len, e, a.k, a.fast != 0, &v)) {
            const uint32_t lo = a.idx_off[v], hi = a.idx_off[v + 1];
            if (lo != hi) {
                // a k-mer of a conserved region is in a large share of the references: its list comes
                // as a bitmap and is counted without atomics (below); everything else by cursor
                const uint32_t did = a.dense_id ? a.dense_id[v] : kNoDense;
                uint32_t dslot = kMaxDenseQ;
                if (did != kNoDense) dslot = atomicAdd(&n_dense_q, 1u);
                if (dslot < kMaxDenseQ) {
                    *(dtop - dslot) = did;
                } else {
                    const uint32_t slot = atomicAdd(&n_kmers, 1u);
                    cur[slot] = lo;
                    end[slot] = hi;
                }
                mine += hi - lo;
            }
        }
    }
    __syncthreads();
    const uint32_t nk = n_kmers;
    const uint32_t nd = min(n_dense_q, kMaxDenseQ);
    const uint32_t ntiles = (a.n_refs + kTileRefs - 1) / kTileRefs;
    int16_t *row = a.scores + (size_t)q * a.stride;
    for (uint32_t t = 0; t < ntiles; t++) {
        const uint32_t tile_lo = t * kTileRefs;
        const uint32_t tile_hi = min(tile_lo + (uint32_t)kTileRefs, a.n_refs);
        for (uint32_t i = tid; i < kTileRefs / 8; i += kCountThreads) reinterpret_cast<uint4 *>(hist)[i] = uint4{0u, 0u, 0u, 0u};
        if (tid == 0) next_kmer = 0;
        __syncthreads();
        for (uint32_t guard = 0; guard < (1u << 22); guard++) {
            uint32_t i = 0;
            if (lane == 0) i = atomicAdd(&next_kmer, 1u);
            i = __builtin_amdgcn_readfirstlane(i);
            if (i >= nk) break;
            uint32_t c = cur[i];
            const uint32_t e = end[i];
            // First a probe of 64 postings, one per lane.  Most visits end here: 1250 of a query's ~1330 cursor lists
            // are short -- ~130 postings at 500 000 references, eight or so per tile -- and the wide loop below costs
            // such a visit a hundred instructions (the kernel is bound by instruction issue, not by the round trips:
            // 32 waves per CU hide those).  The lists are ascending: the postings of this tile are a prefix.
            {
                const uint32_t x = c + (uint32_t)lane;
                const uint32_t id = x < e ? a.idx_ids[x] : 0xFFFFFFFFu;
                const bool in = id < tile_hi;
                if (in) {
                    const uint32_t r = id - tile_lo;
                    atomicAdd(&hist[r >> 1], 1u << (16 * (r & 1)));
                }
                const uint32_t cnt = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in));
                c += cnt;
                if (cnt < 64u) {
                    if (lane == 0) cur[i] = c;
                    continue;
                }
            }
            // kWide x 1 KiB in flight per wave: every lane reads four consecutive postings per load
            // (most postings sit in a few hundred long lists -- k-mers of conserved regions -- and
            // one wave streams each of them: bytes in flight are what bounds it)
            for (uint32_t g2 = 0; c < e && g2 < (1u << 22); g2++) {
                uint32_t id[kWide][4];
#pragma unroll
                for (int u = 0; u < kWide; u++) {
                    const uint32_t x = c + 4u * (uint32_t)lane + 256u * u;
                    if (x + 4u <= e) {
                        const uint32_t *src = a.idx_ids + x;  // (4-byte aligned: three dwords + one, or one 16-byte load)
                        id[u][0] = src[0];
                        id[u][1] = src[1];
                        id[u][2] = src[2];
                        id[u][3] = src[3];
                    } else {
#pragma unroll
                        for (int v = 0; v < 4; v++) id[u][v] = (x + v < e) ? a.idx_ids[x + v] : 0xFFFFFFFFu;
                    }
                }
                uint32_t cnt = 0;
#pragma unroll
                for (int u = 0; u < kWide; u++) {
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        if (id[u][v] < tile_hi) {
                            const uint32_t r = id[u][v] - tile_lo;
                            atomicAdd(&hist[r >> 1], 1u << (16 * (r & 1)));
                            cnt++;
                        }
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
                c += cnt;
                if (cnt < 256u * kWide) break;
            }
            if (lane == 0) cur[i] = c;
        }
        __syncthreads();
        // Dense k-mers: thread t owns the 32 references of bitmap word t of this tile and counts, for
        // each of them, in how many of the query's dense bitmaps its bit is set -- bit-sliced: the
        // planes hold one bit of all 32 counters each; eight bitmap words go in with seven carry-save
        // adders (ones / twos / fours) and one ripple of the resulting eights (7 operations per word,
        // no atomics, nothing but registers).
        if (nd) {
            // (the planes above the fours: as many as the query's number of dense k-mers has bits beyond three -- a
            // hundred bitmaps need four of the seven; the ripple and the unpacking below are compiled for each count)
            auto dense_path = [&](auto nhi_c) {
                constexpr int NHI = decltype(nhi_c)::value;
                const uint32_t *bw = a.dense_bits + (size_t)(tile_lo >> 5) + tid;
                uint32_t ones = 0, twos = 0, fours = 0, hi[NHI > 0 ? NHI : 1] = {0};  // hi[p]: weight 8 << p
                auto csa = [](uint32_t &h, uint32_t &l, uint32_t x, uint32_t y, uint32_t z) {
                    const uint32_t u = x ^ y;
                    h = (x & y) | (u & z);
                    l = u ^ z;
                };
                for (uint32_t i = 0; i < nd; i += 8) {
                    uint32_t w[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) w[u] = (i + u < nd) ? bw[(size_t)*(dtop - (i + u)) * a.dense_words] : 0u;
                    uint32_t twosA, twosB, foursA, foursB, eights;
                    csa(twosA, ones, ones, w[0], w[1]);
                    csa(twosB, ones, ones, w[2], w[3]);
                    csa(foursA, twos, twos, twosA, twosB);
                    csa(twosA, ones, ones, w[4], w[5]);
                    csa(twosB, ones, ones, w[6], w[7]);
                    csa(foursB, twos, twos, twosA, twosB);
                    csa(eights, fours, fours, foursA, foursB);
                    uint32_t carry = eights;
#pragma unroll
                    for (int p = 0; p < NHI; p++) {
                        const uint32_t t2 = hi[p] & carry;
                        hi[p] ^= carry;
                        carry = t2;
                    }
                }
                // add my 32 counts to the tile's counters: words 16 t .. 16 t + 15 are mine alone now
                if constexpr (NHI <= 5) {
                    // counts below 256: four references at a time -- their bits of a plane are a nibble, one
                    // multiplication spreads the nibble's bits over the four bytes of a word (bit i to bit 8 i), the
                    // planes are or-ed in at their weights; two byte shuffles make the counters' two 16-bit pairs.
                    // (group g of lane l in step (g - l) mod 8, a 64-bit access each: the lanes of a wave spread over
                    // the LDS banks)
#pragma unroll 2
                    for (int gg = 0; gg < 8; gg++) {
                        const int g = (gg + lane) & 7;
                        const int b = 4 * g;
                        auto spread = [&](uint32_t plane) -> uint32_t { return (((plane >> b) & 0xFu) * 0x00204081u) & 0x01010101u; };
                        uint32_t acc = spread(ones) | (spread(twos) << 1) | (spread(fours) << 2);
#pragma unroll
                        for (int p = 0; p < NHI; p++) acc |= spread(hi[p]) << (3 + p);
                        uint2 *hw = reinterpret_cast<uint2 *>(&hist[16 * tid + 2 * g]);
                        uint2 v = *hw;
                        v.x += (acc & 0xFFu) | ((acc & 0xFF00u) << 8);
                        v.y += ((acc >> 16) & 0xFFu) | ((acc >> 24) << 16);
                        *hw = v;
                    }
                } else {
                    // (word j of lane l in step (j - l) mod 16: the lanes of a wave spread over the LDS banks)
#pragma unroll 4
                    for (int jj = 0; jj < 16; jj++) {
                        const int j = (jj + lane) & 15;
                        const int b0 = 2 * j, b1 = 2 * j + 1;
                        uint32_t cl = ((ones >> b0) & 1u) | (((twos >> b0) & 1u) << 1) | (((fours >> b0) & 1u) << 2);
                        uint32_t ch = ((ones >> b1) & 1u) | (((twos >> b1) & 1u) << 1) | (((fours >> b1) & 1u) << 2);
#pragma unroll
                        for (int p = 0; p < NHI; p++) {
                            cl |= ((hi[p] >> b0) & 1u) << (3 + p);
                            ch |= ((hi[p] >> b1) & 1u) << (3 + p);
                        }
                        hist[16 * tid + j] += cl | (ch << 16);
                    }
                }
            };
            // (counts up to nd: 32 - clz(nd) bits, three of them in ones / twos / fours)
            const int bits = 32 - __builtin_clz(nd);
            switch (bits > 3 ? bits - 3 : 0) {
            case 0: dense_path(std::integral_constant<int, 0>()); break;
            case 1: dense_path(std::integral_constant<int, 1>()); break;
            case 2: dense_path(std::integral_constant<int, 2>()); break;
            case 3: dense_path(std::integral_constant<int, 3>()); break;
            case 4: dense_path(std::integral_constant<int, 4>()); break;
            case 5: dense_path(std::integral_constant<int, 5>()); break;
            case 6: dense_path(std::integral_constant<int, 6>()); break;
            default: dense_path(std::integral_constant<int, 7>()); break;
            }
            __syncthreads();
        }
        if constexpr (!CAND) {
            // tile scores out: two int16 per 32-bit store (row stride is even)
            uint32_t *dst = reinterpret_cast<uint32_t *>(row + tile_lo);
            const uint32_t words = (tile_hi - tile_lo + 1) / 2;
            for (uint32_t i = tid; i < words; i += kCountThreads) dst[i] = hist[i];
            __syncthreads();
        } else {
            // my 32 references of the tile: words 16 tid .. 16 tid + 15, taken as they are read (pair g of lane l in
            // step (g - l) mod 8, a 64-bit access each: the lanes of a wave spread over the LDS banks) -- kept in an
            // array they would be indexed by the lane: sixteen selects per word
            const uint32_t my_lo = tile_lo + 32u * tid;
            if (t == 0) {
                // t0: the topm-th largest of the 1024 per-thread maxima (8-way search: seven ballots per pass)
                int mx = -1;
#pragma unroll
                for (int gg = 0; gg < 8; gg++) {
                    const int g = (gg + lane) & 7;
                    const uint2 wv = *reinterpret_cast<const uint2 *>(&hist[16 * tid + 2 * g]);
                    const uint32_t id0 = my_lo + 4u * (uint32_t)g;
                    if (id0 < a.n_refs) mx = max(mx, (int)(wv.x & 0xffffu));
                    if (id0 + 1 < a.n_refs) mx = max(mx, (int)(wv.x >> 16));
                    if (id0 + 2 < a.n_refs) mx = max(mx, (int)(wv.y & 0xffffu));
                    if (id0 + 3 < a.n_refs) mx = max(mx, (int)(wv.y >> 16));
                }
                // (no score exceeds the number of the query's k-mers with a posting: every one adds at most 1 to a reference)
                int lo = 0, hi = (int)(nk + nd) + 1;  // count(max >= lo) >= topm > count(max >= hi)
                for (int guard = 0; guard < 8 && hi - lo > 1; ++guard) {
                    if (tid < 8) c_cnt[tid] = 0;
                    __syncthreads();
                    int th[7];
#pragma unroll
                    for (int x = 0; x < 7; x++) {
                        const int tx = lo + (int)(((long long)(hi - lo) * (x + 1)) / 8);
                        th[x] = tx > lo ? tx : lo + 1;
                        const uint32_t c = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(mx >= th[x]));
                        if (lane == 0 && c) atomicAdd(&c_cnt[x], c);
                    }
                    __syncthreads();
                    int nlo = lo, nhi = hi;
                    bool hi_set = false;
#pragma unroll
                    for (int x = 0; x < 7; x++) {
                        if (th[x] >= hi) continue;
                        if (c_cnt[x] >= a.topm) nlo = th[x];
                        else if (!hi_set) {
                            nhi = th[x];
                            hi_set = true;
                        }
                    }
                    lo = nlo;
                    hi = nhi;
                    __syncthreads();
                }
                if (tid == 0) c_t0 = lo;
                __syncthreads();
            }
            const int t0 = c_t0;
#pragma unroll
            for (int gg = 0; gg < 8; gg++) {
                const int g = (gg + lane) & 7;
                const uint2 wv = *reinterpret_cast<const uint2 *>(&hist[16 * tid + 2 * g]);
                // (nearly always none of the four reaches t0: one comparison of the largest decides)
                const int v4[4] = {(int)(wv.x & 0xffffu), (int)(wv.x >> 16), (int)(wv.y & 0xffffu), (int)(wv.y >> 16)};
                if (max(max(v4[0], v4[1]), max(v4[2], v4[3])) < t0) continue;
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const uint32_t id = my_lo + 4u * (uint32_t)g + (uint32_t)h;
                    if (id < a.n_refs && v4[h] >= t0) {
                        const uint32_t slot = atomicAdd(&c_ncand, 1u);
                        if (slot < a.cand_cap)
                            a.cand[(size_t)q * a.cand_cap + slot] = ((unsigned long long)(uint32_t)(v4[h] + 32768) << 32) | id;
                    }
                }
            }
            __syncthreads();
        }
    }
    if constexpr (CAND) {
        if (tid == 0) a.cand_n[q] = c_ncand;
    }
    // postings visited = sum of the list lengths of the query's k-mers (each read once)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0 && mine) atomicAdd(a.postings, mine);
    if (tid == 0) a.nkq[q] = (uint32_t)((len > a.k) ? len - a.k : 0);
}

// ---------------------------------------------------------------- select

struct SelectArgs {
    const int16_t *scores;  // [nq][stride]
    const uint32_t *nkq;    // [nq] upper bound of the scores of query q
    uint32_t *out_ids;      // [nq][max]
    float *out_scores;
    uint32_t *out_n;
    uint32_t n_refs, stride, max;
};

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum, uint32_t *total) {
    // exclusive scan over the workgroup (kSelThreads), wave64 ballot-free version
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < kSelThreads / 64; w++) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// Top-M of one query's score row, M = min(max, n_refs), order (score desc, id desc)
// (kmer_search.cpp:405-418 partial_sort on pair<score, id>).  The cut score is found by bisection
// on "how many scores are >= t" -- pure reductions over 16-byte vector loads of a row that sits in
// L2 (200 KB for 100k references), no atomics on a histogram whose low bins every lane hits --
// then one ordered pass emits everything above the cut plus the ties with the LARGEST ids.
__global__ void __launch_bounds__(kSelThreads) kmer_select_kernel(SelectArgs a) {
    __shared__ unsigned long long cand[kSelMax];
    __shared__ uint32_t wsum[kSelThreads / 64];
    __shared__ uint32_t sh_slot;
    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t n_refs = a.n_refs;
    const int16_t *sc = a.scores + (size_t)q * a.stride;  // stride % 8 == 0: rows are 16-byte aligned
    const uint4 *sc8 = reinterpret_cast<const uint4 *>(sc);
    const uint32_t nvec = (n_refs + 7) / 8;
    const uint32_t M = min(a.max, n_refs);
    const int top = (int)min(a.nkq[q], (uint32_t)kMaxQueryLen);  // no score can exceed the k-mer count

    // f(value, index) over the 8 scores of vector i; entries past n_refs are skipped
    auto for8 = [&](uint32_t i, auto &&f) {
        const uint4 v = sc8[i];
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
        const uint32_t base = 8 * i;
        if (base + 8 <= n_refs) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f((int)(int16_t)(wds[j] & 0xffffu), base + 2 * j);
                f((int)(int16_t)(wds[j] >> 16), base + 2 * j + 1);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (base + 2 * j < n_refs) f((int)(int16_t)(wds[j] & 0xffffu), base + 2 * j);
                if (base + 2 * j + 1 < n_refs) f((int)(int16_t)(wds[j] >> 16), base + 2 * j + 1);
            }
        }
    };
    auto block_sum = [&](uint32_t x) -> uint32_t {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        __syncthreads();  // wsum free again
        if ((tid & 63) == 0) wsum[tid >> 6] = x;
        __syncthreads();
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kSelThreads / 64; w++) t += wsum[w];
        return t;
    };
    // invariant: count(>= lo) >= M > count(>= hi).  Each pass over the row (it sits in L2) counts
    // against kWays - 1 thresholds at once: log_8 instead of log_2 passes.
    constexpr int kWays = 8;
    int lo = 0, hi = top + 1;
    uint32_t c_hi = 0;
    uint32_t n_ge_cut = 0;  // how many scores reach the cut, if the short cut below found it (else 0)
    // Long rows first try a short cut: a histogram of every 16th vector places a threshold T0 that
    // about 2 M + 128 scores of the whole row should reach -- few enough that a histogram of the
    // scores >= T0 over the WHOLE row has no hot bins (the bulk of a row is small scores, which is
    // what rules a plain LDS histogram out).  If at least M scores turn out to reach T0, the cut and
    // the number of scores above it come out of that histogram: two passes instead of four or five;
    // otherwise the search below starts from [0, T0).  The histogram borrows cand[] (unused so far).
    constexpr uint32_t kSample = 16;
    if (nvec >= 2048 && top < 2 * kSelMax) {  // (a query of more k-mers than bins: the general search below)
        uint32_t *hist = reinterpret_cast<uint32_t *>(cand);  // bins 0..top
        __shared__ int f_t;
        __shared__ uint32_t f_above, f_at;
        const uint32_t nb = (uint32_t)top + 1;
        // largest bin t whose suffix count (bins >= t) reaches `target`: f_t (-1: there is none),
        // f_above = count in the bins above t, f_at = count in bin t
        auto suffix_find = [&](uint32_t target) {
            const uint32_t chunk = (nb + kSelThreads - 1) / kSelThreads;
            const uint32_t rb = min(nb, (uint32_t)tid * chunk), re = min(nb, rb + chunk);  // reversed bin index
            uint32_t sum = 0;
            for (uint32_t r = rb; r < re; r++) sum += hist[nb - 1 - r];
            if (tid == 0) f_t = -1;
            uint32_t total;
            const uint32_t excl = block_excl_scan(sum, wsum, &total);  // (synchronises)
            if (excl < target && excl + sum >= target) {
                uint32_t run = excl;
                for (uint32_t r = rb; r < re; r++) {
                    const uint32_t h = hist[nb - 1 - r];
                    if (run + h >= target) {
                        f_t = (int)(nb - 1 - r);
                        f_above = run;
                        f_at = h;
                        break;
                    }
                    run += h;
                }
            }
            __syncthreads();
        };
        for (uint32_t i = tid; i < nb; i += kSelThreads) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = (uint32_t)tid * kSample; i < nvec; i += kSelThreads * kSample)
            for8(i, [&](int v, uint32_t) {
                if (v > 0) atomicAdd(&hist[min(v, top)], 1u);
            });
        __syncthreads();
        const uint32_t target_s = 2 * M / kSample + 8;
        suffix_find(target_s);
        const int T0 = f_t;
        const bool usable = T0 >= 1 && f_above + f_at <= 8 * target_s;  // (not one giant group of equal scores)
        __syncthreads();
        if (usable) {
            for (uint32_t i = tid; i < nb; i += kSelThreads) hist[i] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < nvec; i += kSelThreads)
                for8(i, [&](int v, uint32_t) {
                    if (v >= T0) atomicAdd(&hist[min(v, top)], 1u);
                });
            __syncthreads();
            suffix_find(M);
            if (f_t >= 0) {  // at least M scores reach T0: the cut is among them
                lo = f_t;
                hi = f_t + 1;
                c_hi = f_above;
                n_ge_cut = f_above + f_at;
            } else {  // fewer: count(>= T0) < M, the cut is below T0
                uint32_t part = 0;
                for (uint32_t i = tid; i < nb; i += kSelThreads) part += hist[i];
                hi = T0;
                c_hi = block_sum(part);
            }
            __syncthreads();
        }
    }
    for (int guard = 0; guard < 20 && hi - lo > 1; ++guard) {
        int th[kWays - 1];
        uint32_t c[kWays - 1];
#pragma unroll
        for (int x = 0; x < kWays - 1; x++) {
            // ascending thresholds strictly inside (lo, hi); duplicates at the top when the gap is small
            const int t = lo + (int)(((long long)(hi - lo) * (x + 1)) / kWays);
            th[x] = t > lo ? t : lo + 1;
            c[x] = 0;
        }
        for (uint32_t i = tid; i < nvec; i += kSelThreads)
            for8(i, [&](int v, uint32_t) {
#pragma unroll
                for (int x = 0; x < kWays - 1; x++) c[x] += (v >= th[x]) ? 1u : 0u;
            });
#pragma unroll
        for (int x = 0; x < kWays - 1; x++) c[x] = block_sum(c[x]);
        // the largest threshold that still has M scores at or above it becomes lo, the next one hi
        int nlo = lo, nhi = hi;
        uint32_t nc_hi = c_hi;
        bool hi_set = false;
#pragma unroll
        for (int x = 0; x < kWays - 1; x++) {
            if (th[x] >= hi) continue;
            if (c[x] >= M) {
                nlo = th[x];
            } else if (!hi_set) {
                nhi = th[x];
                nc_hi = c[x];
                hi_set = true;
            }
        }
        lo = nlo;
        hi = nhi;
        c_hi = nc_hi;
    }
    const int cut = lo;
    const uint32_t acc = c_hi;  // scores > cut: all taken

    // Few enough scores at or above the cut to sort them all: take every one of them -- the sort below
    // orders by (score, id) descending, so the first M are the ones above the cut plus the ties with
    // the LARGEST ids -- in one unordered pass.
    const bool take_all = n_ge_cut != 0 && n_ge_cut <= (uint32_t)kSelMax;
    if (take_all) {
        __syncthreads();
        if (tid == 0) sh_slot = 0;
        __syncthreads();
        for (uint32_t i = tid; i < nvec; i += kSelThreads)
            for8(i, [&](int v, uint32_t id) {
                if (v >= cut) {
                    const uint32_t slot = atomicAdd(&sh_slot, 1u);
                    cand[slot] = ((unsigned long long)(uint32_t)(v + 32768) << 32) | id;
                }
            });
    } else {
    // Ordered pass: which ties (score == cut) to take -- those with the largest ids.  A wave owns
    // a contiguous range of vectors and reads it 64 vectors (1 KiB, coalesced) at a time; the rank
    // of a tie in id order is (ties in earlier waves) + (earlier iterations) + (lower lanes).
    constexpr int kSelWaves = kSelThreads / 64;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t vw = ((nvec + kSelWaves - 1) / kSelWaves + 63) / 64 * 64;  // vectors per wave
    const uint32_t w0 = min(nvec, (uint32_t)wave * vw), w1 = min(nvec, w0 + vw);
    uint32_t my_eq = 0;
    for (uint32_t i = w0 + lane; i < w1; i += 64) for8(i, [&](int v, uint32_t) { my_eq += (v == cut) ? 1u : 0u; });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_eq += __shfl_xor(my_eq, off);  // ties in my wave's range
    __syncthreads();
    if (lane == 0) wsum[wave] = my_eq;
    if (tid == 0) sh_slot = 0;
    __syncthreads();
    uint32_t tot_eq = 0, run = 0;  // run: ties before the vectors of this iteration
    for (int w = 0; w < kSelWaves; w++) {
        if (w < wave) run += wsum[w];
        tot_eq += wsum[w];
    }
    const uint32_t skip_eq = tot_eq - (M - acc);  // ties to skip: the smallest ids
    for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
        const uint32_t i = i0 + lane;
        uint32_t c = 0;
        if (i < w1) for8(i, [&](int v, uint32_t) { c += (v == cut) ? 1u : 0u; });
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        uint32_t eq_rank = run + incl - c;
        run += __shfl(incl, 63);
        if (i < w1)
            for8(i, [&](int v, uint32_t id) {
                bool take = v > cut;
                if (v == cut) {
                    take = eq_rank >= skip_eq;
                    eq_rank++;
                }
                if (take) {
                    const uint32_t slot = atomicAdd(&sh_slot, 1u);
                    // sort key: score (biased) high, id low -> descending order = (score desc, id desc)
                    if (slot < kSelMax) cand[slot] = ((unsigned long long)(uint32_t)(v + 32768) << 32) | id;
                }
            });
    }
    }
    __syncthreads();
    const uint32_t out_base = sh_slot;
    const uint32_t n = min(out_base, (uint32_t)kSelMax);
    uint32_t P = 1;
    while (P < n) P <<= 1;
    for (uint32_t i = n + threadIdx.x; i < P; i += kSelThreads) cand[i] = 0ull;
    __syncthreads();
    // bitonic sort, descending
    for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
        for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += kSelThreads) {
                const uint32_t ixj = i ^ j2;
                if (ixj > i) {
                    const unsigned long long x = cand[i], y = cand[ixj];
                    const bool desc = ((i & k2) == 0);
                    if (desc ? (x < y) : (x > y)) {
                        cand[i] = y;
                        cand[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < M; i += kSelThreads) {
        if (i < n) {
            const unsigned long long c = cand[i];
            a.out_ids[(size_t)q * a.max + i] = (uint32_t)c;
            a.out_scores[(size_t)q * a.max + i] = (float)((int)(uint32_t)(c >> 32) - 32768);
        }
    }
    if (threadIdx.x == 0) a.out_n[q] = n < M ? n : M;
}

// Top-M out of the candidate list kmer_count_kernel<true> left for the query: a bitonic sort of its keys (score,
// id) descending.  out_n = 0xFFFFFFFF: the list overflowed -- the host repeats the launch with the score rows.
struct SelectCandArgs {
    const unsigned long long *cand;
    const uint32_t *cand_n;
    uint32_t cand_cap;
    uint32_t *out_ids;
    float *out_scores;
    uint32_t *out_n;
    uint32_t max;
};
__global__ void __launch_bounds__(kSelThreads) kmer_select_cand_kernel(SelectCandArgs a) {
    __shared__ unsigned long long cand[kSelMax];
    const uint32_t q = blockIdx.x;
    const uint32_t have = a.cand_n[q];
    if (have > a.cand_cap) {
        if (threadIdx.x == 0) a.out_n[q] = 0xFFFFFFFFu;
        return;
    }
    const uint32_t n = have, M = a.max;
    uint32_t P = 1;
    while (P < n) P <<= 1;
    for (uint32_t i = threadIdx.x; i < P; i += kSelThreads) cand[i] = i < n ? a.cand[(size_t)q * a.cand_cap + i] : 0ull;
    __syncthreads();
    for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
        for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += kSelThreads) {
                const uint32_t ixj = i ^ j2;
                if (ixj > i) {
                    const unsigned long long x = cand[i], y = cand[ixj];
                    const bool desc = ((i & k2) == 0);
                    if (desc ? (x < y) : (x > y)) {
                        cand[i] = y;
                        cand[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < M; i += kSelThreads) {
        if (i < n) {
            const unsigned long long c = cand[i];
            a.out_ids[(size_t)q * a.max + i] = (uint32_t)c;
            a.out_scores[(size_t)q * a.max + i] = (float)((int)(uint32_t)(c >> 32) - 32768);
        }
    }
    if (threadIdx.x == 0) a.out_n[q] = n < M ? n : M;
}

}  // namespace

static int index_ready(sina_hip_ctx *c) {
    if (!c->st->have_refs) SH_FAIL("k-mer search: no references uploaded");
    if (!c->st->have_index) SH_FAIL("k-mer search: no index (call sina_hip_build_index / upload_index)");
    return 0;
}

// Posting lists longer than 1/64 of the references (k-mers of conserved regions: a few hundred per
// query hold 97 % of its postings) as bitmaps over the references, zero-padded to whole tiles, for
// the count kernel's atomic-free path.  (By bytes a bitmap breaks even with a list at 1/32; the bitmaps of all
// queries in flight are the same few thousand and come out of L2 / MALL, the lists out of HBM: 1/64 .. 1/128
// measured 4 % faster at 100 000 references, 9 % at 500 000 -- tools/perf_kmer_dense.py.)  Built from the CSR index by the first search after the
// index changed (also when the index arrived by broadcast), under the store's mutex.
static int ensure_dense(sina_hip_ctx *c) {
    sina_hip_store *st = c->st;
    std::lock_guard<std::mutex> lk(st->aux_mu);
    if (st->dense_ready) return 0;
    hipStream_t s = c->stream;
    const uint32_t nk = 1u << (2 * st->k);
    uint32_t div_ = 64;
    if (const std::string e_ = test_knob("dense_div"); !e_.empty()) div_ = (uint32_t)std::max(1, atoi(e_.c_str()));
    const uint32_t thresh = std::max<uint32_t>(256u, st->n_refs / div_);
    const uint32_t ntiles = (st->n_refs + kTileRefs - 1) / kTileRefs;
    st->dense_words = ntiles * (uint32_t)(kTileRefs / 32);
    sina_hip::DevBuf counter;
    if (st->dense_id.reserve_exact(4 * (size_t)nk) || counter.reserve_exact(4)) return 1;
    SH_CHECK(hipMemsetAsync(counter.p, 0, 4, s));
    hipLaunchKernelGGL(mark_dense, dim3((nk + 255) / 256), dim3(256), 0, s, st->idx_off.as<uint32_t>(), nk, thresh,
                       st->dense_id.as<uint32_t>(), counter.as<uint32_t>());
    SH_CHECK(hipGetLastError());
    SH_CHECK(hipStreamSynchronize(s));
    uint32_t nd = 0;
    SH_CHECK(hipMemcpy(&nd, counter.p, 4, hipMemcpyDeviceToHost));
    counter.release();
    st->n_dense = nd;
    if (nd) {
        const size_t bytes = 4 * (size_t)nd * st->dense_words;
        if (st->dense_bits.reserve_exact(bytes)) return 1;
        SH_CHECK(hipMemsetAsync(st->dense_bits.p, 0, bytes, s));
        hipLaunchKernelGGL(fill_dense, dim3((unsigned)(((uint64_t)nk * 64 + 255) / 256)), dim3(256), 0, s,
                           st->idx_off.as<uint32_t>(), st->idx_ids.as<uint32_t>(), st->dense_id.as<uint32_t>(), nk,
                           st->dense_bits.as<uint32_t>(), st->dense_words);
        SH_CHECK(hipGetLastError());
    }
    SH_CHECK(hipStreamSynchronize(s));  // (other contexts' streams read it next)
    st->dense_ready = true;
    return 0;
}

// counts + selects for nq queries whose masks are already on the device
// (the candidate-list path, kmer_count_kernel<true>: at least two tiles of references -- tile 0 must hold `max`
// of them --, a top-M small enough for the list; SINA_HIP_TEST=kmer_rows=1 keeps the score rows)
constexpr uint32_t kCandCap = 4096, kCandMaxM = 128;
static bool kmer_cand_path(const sina_hip_ctx *c, uint32_t max) {
    return max <= kCandMaxM && c->st->n_refs >= 2u * (uint32_t)kTileRefs && atoi(test_knob("kmer_rows").c_str()) == 0;
}
static int kmer_topk_device(sina_hip_ctx *c, const uint8_t *d_qmask, const uint64_t *d_qoff, uint32_t nq,
                            uint32_t max, uint32_t max_qlen, bool want_scores_only, bool cand_path) {
    hipStream_t s = c->stream;
    const uint32_t stride = (c->st->n_refs + 7u) & ~7u;  // rows 16-byte aligned (vector loads in the select kernel)
    if ((!cand_path && c->k_scores.reserve((size_t)nq * stride * 2 + 64)) || c->k_tmp2.reserve(8) || c->k_tmp0.reserve(4 * (size_t)nq) ||
        (cand_path && (c->k_tmp1.reserve(8 * (size_t)nq * kCandCap) || c->k_out_n.reserve((size_t)nq * 4))))
        return 1;
    SH_CHECK(hipMemsetAsync(c->k_tmp2.p, 0, 8, s));
    CountArgs ca;
    ca.qmask = d_qmask;
    ca.qoff = d_qoff;
    ca.idx_off = c->st->idx_off.as<uint32_t>();
    ca.idx_ids = c->st->idx_ids.as<uint32_t>();
    ca.scores = c->k_scores.as<int16_t>();
    ca.nkq = c->k_tmp0.as<uint32_t>();
    ca.postings = c->k_tmp2.as<unsigned long long>();
    ca.n_refs = c->st->n_refs;
    ca.stride = stride;
    ca.kmax = (max_qlen + 63u) & ~63u;
    ca.k = c->st->k;
    ca.fast = c->st->nofast ? 0 : 1;
    if (ensure_dense(c)) return 1;
    ca.dense_id = c->st->n_dense ? c->st->dense_id.as<uint32_t>() : nullptr;
    ca.dense_bits = c->st->dense_bits.as<uint32_t>();
    ca.dense_words = c->st->dense_words;
    ca.cand = cand_path ? c->k_tmp1.as<unsigned long long>() : nullptr;
    ca.cand_n = cand_path ? c->k_out_n.as<uint32_t>() : nullptr;  // (kmer_select_cand_kernel turns it into out_n in place)
    ca.cand_cap = kCandCap;
    ca.topm = max;
    const size_t clds = (size_t)kTileRefs * 2 + (size_t)ca.kmax * 9 + 64;
    if (allow_full_lds(reinterpret_cast<const void *>(kmer_count_kernel<false>)) ||
        allow_full_lds(reinterpret_cast<const void *>(kmer_count_kernel<true>)))
        return 1;
    heavy_launch hl(c, s, kHeavyKmer);  // (count + select: device-filling kernels, ctx.h)
    const hipStream_t hs = hl.stream();
    SH_CHECK(hipEventRecord(c->ev[3], hs));
    if (cand_path) hipLaunchKernelGGL(kmer_count_kernel<true>, dim3(nq), dim3(kCountThreads), clds, hs, ca);
    else hipLaunchKernelGGL(kmer_count_kernel<false>, dim3(nq), dim3(kCountThreads), clds, hs, ca);
    SH_CHECK(hipGetLastError());
    SH_CHECK(hipEventRecord(c->ev[4], hs));
    if (cand_path) {
        if (c->k_out_ids.reserve((size_t)nq * max * 4) || c->k_out_scores.reserve((size_t)nq * max * 4)) return 1;
        SelectCandArgs sc;
        sc.cand = ca.cand;
        sc.cand_n = ca.cand_n;
        sc.cand_cap = kCandCap;
        sc.out_ids = c->k_out_ids.as<uint32_t>();
        sc.out_scores = c->k_out_scores.as<float>();
        sc.out_n = c->k_out_n.as<uint32_t>();
        sc.max = max;
        hipLaunchKernelGGL(kmer_select_cand_kernel, dim3(nq), dim3(kSelThreads), 0, hs, sc);
        SH_CHECK(hipGetLastError());
    } else if (!want_scores_only) {
        if (c->k_out_ids.reserve((size_t)nq * max * 4) || c->k_out_scores.reserve((size_t)nq * max * 4) ||
            c->k_out_n.reserve((size_t)nq * 4))
            return 1;
        SelectArgs sa;
        sa.scores = ca.scores;
        sa.nkq = ca.nkq;
        sa.stride = stride;
        sa.out_ids = c->k_out_ids.as<uint32_t>();
        sa.out_scores = c->k_out_scores.as<float>();
        sa.out_n = c->k_out_n.as<uint32_t>();
        sa.n_refs = c->st->n_refs;
        sa.max = max;
        hipLaunchKernelGGL(kmer_select_kernel, dim3(nq), dim3(kSelThreads), 0, hs, sa);
        SH_CHECK(hipGetLastError());
    }
    SH_CHECK(hipEventRecord(c->ev[5], hs));
    return hl.done();
}

}  // namespace sina_hip

using namespace sina_hip;

extern "C" {

int sina_hip_upload_index(sina_hip_ctx *c, unsigned k, int nofast, const uint32_t *offsets, const uint32_t *ids,
                          uint64_t n_postings) {
    if (!c || !offsets || (!ids && n_postings)) SH_FAIL("upload_index: null argument");
    if (k < 1 || k > 12) SH_FAIL("upload_index: k must be in 1..12");
    if (!c->owns_store) SH_FAIL("upload_index: a forked context cannot change the reference store");
    std::lock_guard<std::mutex> lk(c->mu);
    SH_CHECK(hipSetDevice(c->device));
    const uint64_t nk = 1ull << (2 * k);
    if (c->st->idx_off.reserve(4 * (nk + 1)) || c->st->idx_ids.reserve(4 * std::max<uint64_t>(n_postings, 1))) return 1;
    SH_CHECK(hipMemcpyAsync(c->st->idx_off.p, offsets, 4 * (nk + 1), hipMemcpyHostToDevice, c->stream));
    if (n_postings)
        SH_CHECK(hipMemcpyAsync(c->st->idx_ids.p, ids, 4 * n_postings, hipMemcpyHostToDevice, c->stream));
    SH_CHECK(hipStreamSynchronize(c->stream));
    c->st->k = k;
    c->st->nofast = nofast ? 1 : 0;
    c->st->n_postings = n_postings;
    c->st->have_index = true;
    c->st->dense_ready = false;
    return 0;
}

int sina_hip_download_index(sina_hip_ctx *c, uint32_t *offsets, uint32_t *ids) {
    if (!c || !offsets) SH_FAIL("download_index: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->st->have_index) SH_FAIL("download_index: no index");
    if (!ids && c->st->n_postings) SH_FAIL("download_index: null ids");
    SH_CHECK(hipSetDevice(c->device));
    const uint64_t nk = 1ull << (2 * c->st->k);
    SH_CHECK(hipMemcpy(offsets, c->st->idx_off.p, 4 * (nk + 1), hipMemcpyDeviceToHost));
    if (c->st->n_postings) SH_CHECK(hipMemcpy(ids, c->st->idx_ids.p, 4 * c->st->n_postings, hipMemcpyDeviceToHost));
    return 0;
}

int sina_hip_build_index(sina_hip_ctx *c, unsigned k, int nofast) {
    if (!c) SH_FAIL("build_index: null ctx");
    if (k < 1 || k > 12) SH_FAIL("build_index: k must be in 1..12");
    if (!c->owns_store) SH_FAIL("build_index: a forked context cannot change the reference store");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->st->have_refs) SH_FAIL("build_index: upload references first");
    SH_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint64_t n = c->st->total_bases;
    const uint64_t nk = 1ull << (2 * k);
    if (n >= (1ull << 32)) SH_FAIL("build_index: more than 2^32 reference bases");
    DevBuf keys_in, keys_out, flag, pos, tmp;
    int rc = 1;
    do {
        if (keys_in.reserve(8 * std::max<uint64_t>(n, 1)) || keys_out.reserve(8 * std::max<uint64_t>(n, 1)) ||
            flag.reserve(4 * std::max<uint64_t>(n, 1)) || pos.reserve(4 * std::max<uint64_t>(n, 1)))
            break;
        if (c->st->n_refs)
            hipLaunchKernelGGL(ref_kmer_keys, dim3(c->st->n_refs), dim3(256), 0, s, c->st->ref_ab.as<uint32_t>(),
                               c->st->ref_off.as<uint64_t>(), c->st->n_refs, k, nofast == 0, keys_in.as<uint64_t>());
        size_t tb = 0;
        const int end_bit = 32 + 2 * (int)k;
        // (rocPRIM directly; only the bits a key can have: k-mer << 32 | reference id)
        if (rocprim::radix_sort_keys(nullptr, tb, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), (size_t)n, 0u,
                                     (unsigned)end_bit, s) != hipSuccess)
            break;
        if (tmp.reserve(tb + 16)) break;
        if (n && rocprim::radix_sort_keys(tmp.p, tb, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), (size_t)n, 0u,
                                          (unsigned)end_bit, s) != hipSuccess)
            break;
        const unsigned blocks = (unsigned)((n + 255) / 256);
        if (n) hipLaunchKernelGGL(mark_unique, dim3(blocks), dim3(256), 0, s, keys_out.as<uint64_t>(), n,
                                  flag.as<uint32_t>());
        size_t tb2 = 0;
        if (rocprim::exclusive_scan(nullptr, tb2, flag.as<uint32_t>(), pos.as<uint32_t>(), 0u, (size_t)n,
                                    rocprim::plus<uint32_t>(), s) != hipSuccess)
            break;
        if (tmp.reserve(tb2 + 16)) break;
        if (n && rocprim::exclusive_scan(tmp.p, tb2, flag.as<uint32_t>(), pos.as<uint32_t>(), 0u, (size_t)n,
                                         rocprim::plus<uint32_t>(), s) != hipSuccess)
            break;
        uint32_t last_flag = 0, last_pos = 0;
        if (n) {
            if (hipMemcpyAsync(&last_flag, flag.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(&last_pos, pos.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                break;
        }
        const uint64_t np = (uint64_t)last_flag + last_pos;
        DevBuf counts;
        if (counts.reserve(4 * (nk + 1)) || c->st->idx_off.reserve(4 * (nk + 1)) ||
            c->st->idx_ids.reserve(4 * std::max<uint64_t>(np, 1))) {
            counts.release();
            break;
        }
        bool ok = hipMemsetAsync(counts.p, 0, 4 * (nk + 1), s) == hipSuccess;
        if (ok && n)
            hipLaunchKernelGGL(scatter_unique, dim3(blocks), dim3(256), 0, s, keys_out.as<uint64_t>(),
                               flag.as<uint32_t>(), pos.as<uint32_t>(), n, c->st->idx_ids.as<uint32_t>(),
                               counts.as<uint32_t>());
        size_t tb3 = 0;
        ok = ok && rocprim::exclusive_scan(nullptr, tb3, counts.as<uint32_t>(), c->st->idx_off.as<uint32_t>(), 0u,
                                           (size_t)(nk + 1), rocprim::plus<uint32_t>(), s) == hipSuccess;
        ok = ok && tmp.reserve(tb3 + 16) == 0;
        ok = ok && rocprim::exclusive_scan(tmp.p, tb3, counts.as<uint32_t>(), c->st->idx_off.as<uint32_t>(), 0u,
                                           (size_t)(nk + 1), rocprim::plus<uint32_t>(), s) == hipSuccess;
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
        counts.release();
        if (!ok) break;
        c->st->k = k;
        c->st->nofast = nofast ? 1 : 0;
        c->st->n_postings = np;
        c->st->have_index = true;
        c->st->dense_ready = false;
        rc = 0;
    } while (0);
    keys_in.release();
    keys_out.release();
    flag.release();
    pos.release();
    tmp.release();
    if (rc) {
        hipError_t e = hipGetLastError();
        set_error(std::string("build_index failed: ") + hipGetErrorString(e));
    }
    return rc;
}

int sina_hip_kmer_topk(sina_hip_ctx *c, const uint8_t *qmask, const uint64_t *qoff, uint32_t nq, uint32_t max,
                       uint32_t *out_ids, float *out_scores, uint32_t *out_n) {
    if (!c || !qmask || !qoff || !out_ids || !out_scores || !out_n) SH_FAIL("kmer_topk: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    sina_hip_hint_guard hints(c);
    if (index_ready(c)) return 1;
    if (nq == 0) return 0;
    SH_CHECK(hipSetDevice(c->device));
    if (max > c->st->n_refs) max = c->st->n_refs;
    if (max == 0) {
        memset(out_n, 0, sizeof(uint32_t) * nq);
        return 0;
    }
    if (max > (uint32_t)kSelMax) SH_FAIL("kmer_topk: max > 4096 not supported by the LDS select kernel");
    uint32_t max_qlen = 1;
    for (uint32_t q = 0; q < nq; q++) {
        if (qoff[q + 1] - qoff[q] > (uint64_t)kMaxQueryLen) SH_FAIL("kmer_topk: query longer than SINA_HIP_MAX_QUERY_LEN bases");
        max_qlen = std::max<uint32_t>(max_qlen, (uint32_t)(qoff[q + 1] - qoff[q]));
    }
    hipStream_t s = c->stream;
    const uint64_t nqm = qoff[nq] - qoff[0];
    // sub-batches bound the [nq][n_refs] int16 score matrix to ~2 GiB
    const uint32_t per = (uint32_t)std::max<uint64_t>(1, ((uint64_t)2 << 30) / (2ull * std::max<uint32_t>(c->st->n_refs, 1)));
    if (c->qmask.reserve(std::max<uint64_t>(nqm, 1)) || c->k_qoff.reserve(8 * ((uint64_t)nq + 1))) return 1;
    std::vector<uint64_t> rel(nq + 1);
    for (uint32_t q = 0; q <= nq; q++) rel[q] = qoff[q] - qoff[0];
    if (upload(c, 7, c->qmask.p, qmask + qoff[0], nqm, s) || upload(c, 8, c->k_qoff.p, rel.data(), 8 * ((uint64_t)nq + 1), s))
        return 1;
    // one launch range: kernels, results back through pinned staging, statistics; *overflow = a candidate list of the
    // range did not hold its query's candidates (nothing was copied out then: the caller repeats the range with rows)
    auto run_range = [&](uint32_t q0, uint32_t bq, bool cand_path, bool *overflow) -> int {
        if (kmer_topk_device(c, c->qmask.as<uint8_t>(), c->k_qoff.as<uint64_t>() + q0, bq, max, max_qlen, false, cand_path)) return 1;
        // (the kernels have finished: kmer_topk_device waits for the heavy stream)
        if (download(c, 9, c->k_out_ids.p, (size_t)bq * max * 4, s) || download(c, 10, c->k_out_scores.p, (size_t)bq * max * 4, s) ||
            download(c, 11, c->k_out_n.p, (size_t)bq * 4, s) || download(c, 0, c->k_tmp2.p, 8, s))
            return 1;
        SH_CHECK(wait_stream(c, s));
        *overflow = false;
        if (cand_path) {
            const uint32_t *n = static_cast<const uint32_t *>(c->h_stage[11].p);
            for (uint32_t q = 0; q < bq && !*overflow; q++) *overflow = n[q] == 0xFFFFFFFFu;
        }
        unsigned long long visited = 0;
        memcpy(&visited, c->h_stage[0].p, 8);
        float ms = 0, ms2 = 0;
        SH_CHECK(hipEventElapsedTime(&ms, c->ev[3], c->ev[4]));
        SH_CHECK(hipEventElapsedTime(&ms2, c->ev[4], c->ev[5]));
        {
            std::lock_guard<std::mutex> slk(c->st->stats_mu);
            c->st->stats.kmer_count_ms += ms;
            c->st->stats.kmer_select_ms += ms2;
            c->st->stats.kmer_launches++;
            if (!*overflow) {
                c->st->stats.postings += visited;
                c->st->stats.kmer_queries += bq;
            }
        }
        if (*overflow) return 0;
        memcpy(out_ids + (size_t)q0 * max, c->h_stage[9].p, (size_t)bq * max * 4);
        memcpy(out_scores + (size_t)q0 * max, c->h_stage[10].p, (size_t)bq * max * 4);
        memcpy(out_n + q0, c->h_stage[11].p, (size_t)bq * 4);
        return 0;
    };
    const bool cand_path = kmer_cand_path(c, max);
    const uint32_t per_cand = 16384;  // (candidate lists: 32 KB per query)
    for (uint32_t q0 = 0; q0 < nq; q0 += cand_path ? per_cand : per) {
        const uint32_t bq = std::min(cand_path ? per_cand : per, nq - q0);
        bool overflow = false;
        if (run_range(q0, bq, cand_path, &overflow)) return 1;
        if (overflow)  // a giant group of equal scores somewhere in the range: the score rows and the full select
            for (uint32_t r0 = q0; r0 < q0 + bq; r0 += per) {
                bool dummy = false;
                if (run_range(r0, std::min(per, q0 + bq - r0), false, &dummy)) return 1;
            }
    }
    return 0;
}

int sina_hip_kmer_scores(sina_hip_ctx *c, const uint8_t *qmask, uint32_t qlen, int16_t *scores) {
    if (!c || !qmask || !scores) SH_FAIL("kmer_scores: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (index_ready(c)) return 1;
    if (qlen > (uint32_t)kMaxQueryLen) SH_FAIL("kmer_scores: query longer than SINA_HIP_MAX_QUERY_LEN bases");
    SH_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint64_t rel[2] = {0, qlen};
    if (c->qmask.reserve(std::max<uint32_t>(qlen, 1)) || c->k_qoff.reserve(16)) return 1;
    SH_CHECK(hipMemcpyAsync(c->qmask.p, qmask, qlen, hipMemcpyHostToDevice, s));
    SH_CHECK(hipMemcpyAsync(c->k_qoff.p, rel, 16, hipMemcpyHostToDevice, s));
    if (kmer_topk_device(c, c->qmask.as<uint8_t>(), c->k_qoff.as<uint64_t>(), 1, 1, std::max<uint32_t>(qlen, 1), true, false)) return 1;
    SH_CHECK(hipMemcpyAsync(scores, c->k_scores.p, (size_t)c->st->n_refs * 2, hipMemcpyDeviceToHost, s));
    SH_CHECK(hipStreamSynchronize(s));
    return 0;
}

int sina_hip_store_view_get(sina_hip_ctx *c, sina_hip_store_view *v) {
    if (!c || !v) SH_FAIL("store_view_get: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->st->have_refs) SH_FAIL("store_view_get: no references uploaded");
    memset(v, 0, sizeof(*v));
    v->ref_ab = c->st->ref_ab.p;
    v->ref_ab_bytes = 4 * c->st->total_bases;
    v->ref_off = c->st->ref_off.p;
    v->ref_off_bytes = 8 * ((uint64_t)c->st->n_refs + 1);
    v->n_refs = c->st->n_refs;
    v->width = c->st->width;
    v->total_bases = c->st->total_bases;
    if (c->st->have_index) {
        v->idx_offsets = c->st->idx_off.p;
        v->idx_offsets_bytes = 4 * ((1ull << (2 * c->st->k)) + 1);
        v->idx_ids = c->st->idx_ids.p;
        v->idx_ids_bytes = 4 * c->st->n_postings;
        v->k = c->st->k;
        v->nofast = c->st->nofast;
        v->n_postings = c->st->n_postings;
    }
    return 0;
}

int sina_hip_store_alloc_like(sina_hip_ctx *c, sina_hip_store_view *v) {
    if (!c || !v) SH_FAIL("store_alloc_like: null argument");
    if (v->k < 1 || v->k > 12) SH_FAIL("store_alloc_like: k must be in 1..12");
    if (!c->owns_store) SH_FAIL("store_alloc_like: a forked context cannot change the reference store");
    std::lock_guard<std::mutex> lk(c->mu);
    SH_CHECK(hipSetDevice(c->device));
    const uint64_t nk = 1ull << (2 * v->k);
    if (c->st->ref_ab.reserve(4 * std::max<uint64_t>(v->total_bases, 1)) ||
        c->st->ref_off.reserve(8 * ((uint64_t)v->n_refs + 1)) || c->st->idx_off.reserve(4 * (nk + 1)) ||
        c->st->idx_ids.reserve(4 * std::max<uint64_t>(v->n_postings, 1)))
        return 1;
    c->st->n_refs = v->n_refs;
    c->st->width = v->width;
    c->st->total_bases = v->total_bases;
    c->st->k = v->k;
    c->st->nofast = v->nofast;
    c->st->n_postings = v->n_postings;
    c->st->have_refs = c->st->have_index = true;
    c->st->dense_ready = false;  // (the index arrives by broadcast after this call: built by the first search)
    {   // re-read from the device, once, after the broadcast filled it (ensure_ref_off_host)
        std::lock_guard<std::mutex> alk(c->st->aux_mu);
        c->st->ref_off_host_ready.store(false, std::memory_order_release);
        c->st->ref_off_host.clear();
    }
    v->ref_ab = c->st->ref_ab.p;
    v->ref_ab_bytes = 4 * v->total_bases;
    v->ref_off = c->st->ref_off.p;
    v->ref_off_bytes = 8 * ((uint64_t)v->n_refs + 1);
    v->idx_offsets = c->st->idx_off.p;
    v->idx_offsets_bytes = 4 * (nk + 1);
    v->idx_ids = c->st->idx_ids.p;
    v->idx_ids_bytes = 4 * v->n_postings;
    return 0;
}

}  // extern "C"
