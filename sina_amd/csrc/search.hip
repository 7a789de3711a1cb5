// Search-stage sequence comparison on the GPU + sina_hip_compare.
//
// What it computes: for every (query, candidate reference) pair the six counters of
// match_counter as traverse() fills them (reference src/cseq_comparator.cpp:56-111,146-206):
// match / mismatch on columns where both sequences have an unfiltered base, only_a / only_b for
// bases inside the other sequence's column range without a partner, and the overhang counters for
// bases outside that range.  search_filter::operator() calls this once per k-mer candidate --
// 1000 times per query (src/search_filter.cpp:311-313).
//
// How it maps to the hardware: the reference walks both base lists in lock-step, but every
// counter is a function of column membership only:
//   * a base of B (candidate) at column p is an overhang if p lies outside [first, last] unfiltered
//     column of A (query), a match/mismatch if A has an unfiltered base at p, only_b otherwise;
//   * for A the same with roles swapped, and those three numbers follow from counts:
//     in-range(A) = rankA(lastB + 1) - rankA(firstB); only_a = in-range(A) - (match + mismatch);
//     overhang(A) = |A| - in-range(A);
//   * filtered (lower-case) bases behave as absent: in the lock-step walk a filtered base never
//     increments a counter, and its partner, if any, is counted exactly as an unpartnered base.
// One workgroup per query keeps A in LDS as a column bitmap + per-word popcount prefix (rank) +
// the base masks in rank order; each wave then streams one candidate at a time from HBM
// (coalesced 4-byte reads, every base read once) and reduces its counters with wave shuffles.
// HBM-bound: algorithmic bytes = 4 B x sum of candidate lengths (6 MB per query at 1000
// candidates x 1500 bases).
#include <algorithm>
#include <cstring>

#include "common.h"
#include "ctx.h"

namespace sina_hip {
namespace {

constexpr int kCT = 256;  // threads per workgroup

struct CompareArgs {
    const uint32_t *ref_ab;
    const uint64_t *ref_off;
    const uint32_t *q_ab;
    const uint64_t *q_off;
    const uint32_t *cand_ids;
    const uint64_t *cand_off;
    sina_hip_match_counts *out;
    uint32_t width, n_refs;
    int iupac, filter_lc;
};

__global__ void __launch_bounds__(kCT) compare_kernel(CompareArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_tmp[8];
    __shared__ uint32_t s_first, s_last, s_na;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t nwords = (a.width + 31) / 32;
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(smem);            // [nwords] unfiltered columns of A
    uint16_t *wrank = reinterpret_cast<uint16_t *>(bitmap + nwords);   // [nwords + 1] bases before the word
    uint8_t *amask = reinterpret_cast<uint8_t *>(wrank + nwords + 2);  // [|A|] iupac mask by rank

    const uint32_t *A = a.q_ab + a.q_off[q];
    const uint32_t la = (uint32_t)(a.q_off[q + 1] - a.q_off[q]);
    const uint32_t lc_bit = a.filter_lc ? 0x10u : 0u;  // filtered <=> (mask byte & lc_bit) != 0
    for (uint32_t i = tid; i < nwords; i += kCT) bitmap[i] = 0;
    if (tid == 0) {
        s_first = 0xFFFFFFFFu;
        s_last = 0;
    }
    __syncthreads();
    for (uint32_t i = tid; i < la; i += kCT) {
        const uint32_t ab = A[i];
        if ((ab >> 24) & lc_bit) continue;
        const uint32_t pos = ab & 0xFFFFFFu;
        if (pos >= a.width) continue;  // (cannot happen for a sequence of this alignment)
        atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
        atomicMin(&s_first, pos);
        atomicMax(&s_last, pos);
    }
    __syncthreads();
    {   // exclusive prefix popcount over the bitmap words
        const uint32_t chunk = (nwords + kCT - 1) / kCT;
        const uint32_t b = min(nwords, tid * chunk), e = min(nwords, b + chunk);
        uint32_t s = 0;
        for (uint32_t i = b; i < e; i++) s += __popc(bitmap[i]);
        uint32_t x = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) s_tmp[wave] = x;
        __syncthreads();
        uint32_t base = 0, total = 0;
        for (int w = 0; w < kCT / 64; w++) {
            if (w < wave) base += s_tmp[w];
            total += s_tmp[w];
        }
        uint32_t run = base + x - s;
        for (uint32_t i = b; i < e; i++) {
            wrank[i] = (uint16_t)run;
            run += __popc(bitmap[i]);
        }
        if (tid == 0) {
            wrank[nwords] = (uint16_t)total;
            s_na = total;
        }
    }
    __syncthreads();
    // number of unfiltered A bases in columns < p (p <= width)
    auto rank = [&](uint32_t p) -> uint32_t {
        const uint32_t wd = p >> 5;
        if (wd >= nwords) return wrank[nwords];
        return (uint32_t)wrank[wd] + __popc(bitmap[wd] & ((1u << (p & 31)) - 1u));
    };
    for (uint32_t i = tid; i < la; i += kCT) {
        const uint32_t ab = A[i];
        if ((ab >> 24) & lc_bit) continue;
        const uint32_t pos = ab & 0xFFFFFFu;
        if (pos >= a.width) continue;
        amask[rank(pos)] = (uint8_t)((ab >> 24) & 0xFu);
    }
    __syncthreads();
    const uint32_t aF = s_first, aL = s_last, nA = s_na;

    const uint64_t c0 = a.cand_off[q], c1 = a.cand_off[q + 1];
    for (uint64_t c = c0 + wave; c < c1; c += kCT / 64) {
        const uint32_t id = a.cand_ids[c];
        int32_t n_match = 0, n_mis = 0, n_onlyb = 0, n_ovb = 0;
        uint32_t bF = 0xFFFFFFFFu, bL = 0;
        if (id < a.n_refs && nA != 0) {
            const uint32_t *Bp = a.ref_ab + a.ref_off[id];
            const uint32_t lb = (uint32_t)(a.ref_off[id + 1] - a.ref_off[id]);
            for (uint32_t i = lane; i < lb; i += 64) {
                const uint32_t ab = Bp[i];
                if ((ab >> 24) & lc_bit) continue;
                const uint32_t pos = ab & 0xFFFFFFu;
                bF = min(bF, pos);
                bL = max(bL, pos);
                if (pos < aF || pos > aL) {
                    n_ovb++;
                } else if ((bitmap[pos >> 5] >> (pos & 31)) & 1u) {
                    const uint32_t ma = amask[rank(pos)], mb = (ab >> 24) & 0xFu;
                    bool eq;
                    if (a.iupac == SINA_CMP_IUPAC_OPTIMISTIC) eq = (ma & mb) != 0;       // aligned_base.h:153-155
                    else if (a.iupac == SINA_CMP_IUPAC_PESSIMISTIC) eq = (__popc(ma) <= 1) && ma == mb;  // :163-165
                    else eq = ma == mb;                                                  // :167-169
                    if (eq) n_match++;
                    else n_mis++;
                } else {
                    n_onlyb++;
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            n_match += __shfl_xor(n_match, off);
            n_mis += __shfl_xor(n_mis, off);
            n_onlyb += __shfl_xor(n_onlyb, off);
            n_ovb += __shfl_xor(n_ovb, off);
            bF = min(bF, (uint32_t)__shfl_xor((int)bF, off));
            bL = max(bL, (uint32_t)__shfl_xor((int)bL, off));
        }
        if (lane == 0) {
            sina_hip_match_counts m;
            if (bF == 0xFFFFFFFFu || nA == 0) {  // one side has no unfiltered base
                m.only_a_overhang = m.only_b_overhang = m.only_a = m.only_b = m.match = m.mismatch = 0;
            } else {
                const int32_t in_a = (int32_t)(rank(bL + 1) - rank(bF));
                m.match = n_match;
                m.mismatch = n_mis;
                m.only_b = n_onlyb;
                m.only_b_overhang = n_ovb;
                m.only_a = in_a - (n_match + n_mis);
                m.only_a_overhang = (int32_t)nA - in_a;
            }
            a.out[c] = m;
        }
    }
}

}  // namespace
}  // namespace sina_hip

using namespace sina_hip;

extern "C" int sina_hip_compare(sina_hip_ctx *c, const uint32_t *q_ab, const uint64_t *q_off, uint32_t nq,
                                const uint32_t *cand_ids, const uint64_t *cand_off, int iupac_rule,
                                int filter_lowercase, sina_hip_match_counts *out) {
    if (!c || !q_ab || !q_off || !cand_off || !out) SH_FAIL("compare: null argument");
    if (iupac_rule < 0 || iupac_rule > 2) SH_FAIL("compare: unknown iupac rule");
    std::lock_guard<std::mutex> lk(c->mu);
    sina_hip_hint_guard hints(c);
    if (!c->st->have_refs) SH_FAIL("compare: upload references first");
    if (nq == 0) return 0;
    SH_CHECK(hipSetDevice(c->device));
    const uint64_t nqa = q_off[nq] - q_off[0], ncand = cand_off[nq] - cand_off[0];
    if (ncand == 0) return 0;
    if (!cand_ids) SH_FAIL("compare: null candidate ids");
    uint32_t max_la = 0;
    std::vector<uint64_t> qrel(nq + 1), crel(nq + 1);
    for (uint32_t q = 0; q <= nq; q++) {
        qrel[q] = q_off[q] - q_off[0];
        crel[q] = cand_off[q] - cand_off[0];
        if (q < nq) max_la = std::max<uint32_t>(max_la, (uint32_t)(q_off[q + 1] - q_off[q]));
    }
    if (max_la > 65535) SH_FAIL("compare: query longer than 65535 bases");
    for (uint64_t i = 0; i < ncand; i++)
        if (cand_ids[cand_off[0] + i] >= c->st->n_refs) SH_FAIL("compare: reference id out of range");
    const size_t nwords = ((size_t)c->st->width + 31) / 32;
    const size_t lds = 4 * nwords + 2 * (nwords + 2) + ((size_t)max_la + 15) + 16;
    if (lds > 150 * 1024) SH_FAIL("compare: alignment too wide for the device comparison");
    hipStream_t s = c->stream;
    if (c->s_qab.reserve(4 * std::max<uint64_t>(nqa, 1)) || c->s_qoff.reserve(8 * ((uint64_t)nq + 1)) ||
        c->s_cand.reserve(4 * ncand) || c->s_coff.reserve(8 * ((uint64_t)nq + 1)) ||
        c->s_out.reserve(sizeof(sina_hip_match_counts) * ncand))
        return 1;
    SH_CHECK(hipMemcpyAsync(c->s_qab.p, q_ab + q_off[0], 4 * nqa, hipMemcpyHostToDevice, s));
    SH_CHECK(hipMemcpyAsync(c->s_qoff.p, qrel.data(), 8 * ((uint64_t)nq + 1), hipMemcpyHostToDevice, s));
    SH_CHECK(hipMemcpyAsync(c->s_cand.p, cand_ids + cand_off[0], 4 * ncand, hipMemcpyHostToDevice, s));
    SH_CHECK(hipMemcpyAsync(c->s_coff.p, crel.data(), 8 * ((uint64_t)nq + 1), hipMemcpyHostToDevice, s));
    CompareArgs a;
    a.ref_ab = c->st->ref_ab.as<uint32_t>();
    a.ref_off = c->st->ref_off.as<uint64_t>();
    a.q_ab = c->s_qab.as<uint32_t>();
    a.q_off = c->s_qoff.as<uint64_t>();
    a.cand_ids = c->s_cand.as<uint32_t>();
    a.cand_off = c->s_coff.as<uint64_t>();
    a.out = c->s_out.as<sina_hip_match_counts>();
    a.width = c->st->width;
    a.n_refs = c->st->n_refs;
    a.iupac = iupac_rule;
    a.filter_lc = filter_lowercase ? 1 : 0;
    if (allow_full_lds(reinterpret_cast<const void *>(compare_kernel))) return 1;
    SH_CHECK(hipEventRecord(c->ev[3], s));
    hipLaunchKernelGGL(compare_kernel, dim3(nq), dim3(kCT), lds, s, a);
    SH_CHECK(hipGetLastError());
    SH_CHECK(hipEventRecord(c->ev[4], s));
    // (wait for the kernel first: see HostBuf in common.h)
    SH_CHECK(hipStreamSynchronize(s));
    SH_CHECK(hipMemcpyAsync(out, c->s_out.p, sizeof(sina_hip_match_counts) * ncand, hipMemcpyDeviceToHost, s));
    SH_CHECK(hipStreamSynchronize(s));
    float ms = 0;
    SH_CHECK(hipEventElapsedTime(&ms, c->ev[3], c->ev[4]));
    uint64_t bases = 0;
    if (ensure_ref_off_host(c) == 0)
        for (uint64_t i = 0; i < ncand; i++) {
            const uint32_t id = cand_ids[cand_off[0] + i];
            bases += c->st->ref_off_host[id + 1] - c->st->ref_off_host[id];
        }
    std::lock_guard<std::mutex> slk(c->st->stats_mu);
    c->st->stats.compare_ms += ms;
    c->st->stats.compare_bases += bases;
    c->st->stats.compare_launches++;
    return 0;
}
