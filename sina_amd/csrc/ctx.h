// The context object behind the opaque sina_hip_ctx handle.
#pragma once

#include <atomic>
#include <algorithm>
#include <condition_variable>
#include <cmath>
#include <cstring>

#include "common.h"

// Reference store + k-mer index (HBM-resident) and the cumulative counters.  Owned by the context
// sina_hip_init() made; contexts made by sina_hip_fork() point at their parent's.
struct sina_hip_store {
    sina_hip::DevBuf ref_ab, ref_off, idx_off, idx_ids;
    // host copy of the offsets (sizing of DAG-build scratch).  Written once per store content, under
    // aux_mu, BEFORE ref_off_host_ready is set (release); every reader goes through
    // sina_hip::ensure_ref_off_host() and sees either "not ready" or the completed vector.  Forked
    // contexts share the store and call in concurrently (per-context mutexes do not cover it).
    std::vector<uint64_t> ref_off_host;
    std::atomic<bool> ref_off_host_ready{false};
    uint32_t n_refs = 0, width = 0, k = 0, nofast = 0;
    uint64_t n_postings = 0, total_bases = 0;
    bool have_refs = false, have_index = false;
    // Dense posting lists as bitmaps over the references (kmer.hip, ensure_dense): built lazily from
    // the CSR index by the first search after the index changed
    sina_hip::DevBuf dense_id, dense_bits;  // u32 [4^k]: bitmap number or ~0; u32 [n_dense][dense_words]
    uint32_t n_dense = 0, dense_words = 0;
    std::atomic<bool> dense_ready{false};
    std::mutex aux_mu;
    std::mutex stats_mu;
    // One DP kernel at a time per device: a DP launch fills every CU by itself, and contexts that
    // all reach their DP phase together would otherwise run in lock-step (GPU idle while all of
    // them are in their host phases).  Holding this while the DP kernel runs staggers them.
    std::mutex dp_token;
    // The device-filling kernels of all contexts (k-mer count/select, DAG build, DP) go through ONE
    // stream, in the order the host threads reach them: see sina_hip::heavy_launch below.
    hipStream_t heavy = nullptr;
    std::mutex heavy_mu;
    // Round 4: the FIFO is kept on TWO streams taking turns, so that a kernel can start when the kernel
    // queued before it has DISPATCHED its last workgroup ("its queue has run dry") instead of when it has
    // ended -- see sina_hip::heavy_launch.  dry_mem: [0] the flag word (sequence number of the last launch
    // that ran dry), [1 + seq % kDryCounters] that launch's count of started workgroups.
    hipStream_t heavy2 = nullptr;
    hipEvent_t heavy_done[2] = {nullptr, nullptr};  // end of the last launch queued on heavy / heavy2
    // dry_flag: the word hipStreamWaitValue32 waits on -- SIGNAL memory (hipExtMallocWithFlags(hipMallocSignalMemory),
    // what the HIP header documents for that call); dry_mem: the launches' counters of started workgroups, plain memory
    uint32_t *dry_flag = nullptr;
    uint32_t *dry_mem = nullptr;
    // a launch did not end within kChainTimeoutS: its kernels and wait-value operations are still queued and still
    // refer to the contexts' buffers.  From then on every launch fails at once, and nothing of the store or its
    // contexts is freed (hipFree would block on the wedged queue, or a stale kernel would run over reused memory)
    std::atomic<bool> wedged{false};
    std::atomic<bool> chain_broken{false};  // a chained wait timed out once (heavy_launch::done): launches wait for ends from now on
    static constexpr uint32_t kDryCounters = 64;
    uint32_t heavy_seq = 0;          // launches queued so far (guarded by heavy_mu, like everything below)
    int heavy_turn = 0;              // the stream the next launch goes to
    bool heavy_prev_dry = false;     // the last launch signals "dry" (flag word reaches heavy_prev_seq) ...
    uint32_t heavy_prev_seq = 0;     // (its number: NOT heavy_seq -- a launch that failed half-way took a number too)
    bool heavy_prev_any = false;     // ... there has been one at all
    // admission (heavy_launch): launches handed to the GPU and not yet known to have ended, the kind of the
    // last one handed over, and who is waiting to be
    std::condition_variable heavy_cv;
    int heavy_outstanding = 0;
    int heavy_last_kind = -1;
    uint64_t heavy_ticket = 0;
    struct heavy_waiter {
        uint64_t ticket;
        int kind;
    };
    std::vector<heavy_waiter> heavy_waiting;
    // DP launches overlap now (the next one starts in the drain of the one before): stats.dp_busy_ms is the time
    // during which ANY DP kernel was resident -- the sum of the launches' durations minus their overlaps, which
    // are measured against the end event of the launch before (a ring: launch k records dp_end[k % 8])
    hipEvent_t dp_end[8] = {};
    std::atomic<uint64_t> dp_end_no[8] = {};  // which launch's end slot k holds (written under heavy_mu when it is recorded)
    uint64_t dp_seq = 0;
    // largest capacity any context has needed for each scratch buffer so far: a new fork reserves
    // these at once (hipMalloc / hipFree synchronise the device; never in steady state)
    std::atomic<size_t> cap_hint[64] = {};  // (indexed like sina_hip_ctx::scratch(): kNumScratch entries; read without a lock)
    // certified row skip of the DP kernel (mesh_dp.hip PRUNE): the guess rho of "optimum / bound on the whole gain"
    // the next launch starts its queries with, learnt from the queries aligned so far (run_dp_device; under stats_mu).
    // Results never depend on it: a guess that is too bold costs the queries it fails a second sweep, a timid one
    // a wider band.
    float prune_rho = 0.80f;        // the guess a launch WITHOUT a scout pass starts from (below the smallest ratio seen)
    float prune_rho_guard = 0.80f;  // ... and the guard of the scout's values (the 2 % point of the ratios seen)
    sina_hip_stats stats;
    // Trace-back planes -- tens of GB each, the one allocation whose size follows the launch -- belong
    // to the DEVICE, not to a context: only one DP kernel runs at a time, so two planes serve any number
    // of contexts (one being written by the DP kernel, one being walked by the previous launch's
    // backtrack).  A context borrows a plane from its DP launch until its results are on the host.
    // Size per plane: SINA_HIP_TB_GB if set, else 42 % of the memory that is free when the first DP
    // launch is planned (hipMemGetInfo): 288 GB parts run 3072-query 23S launches (107 GB) unprompted.
    struct tb_pool_t {
        static constexpr int kMax = 4;
        std::mutex mu;
        std::condition_variable cv;
        sina_hip::DevBuf plane[kMax];
        bool busy[kMax] = {false, false, false, false};
        int n = 2;            // SINA_HIP_TB_PLANES
        uint64_t budget = 0;  // bytes per plane; 0 = not decided yet
        uint64_t now_budget = 0;  // ... capped by what the device could give when `now_made` planes existed
        int now_made = -1;
    } tb_pool;
};

struct sina_hip_ctx {
    int device = 0;
    int n_cu = 256;  // compute units of the device (4 SIMDs each)
    hipStream_t stream = nullptr;     // everything but the DP: highest priority
    hipStream_t stream_dp = nullptr;  // DP kernel + backtrack: lowest priority (see make_streams)
    std::mutex mu;
    hipEvent_t ev[12];  // [10], [11]: hand-over to / from the store's heavy stream

    sina_hip_store *st = nullptr;
    bool owns_store = false;

    // per-batch scratch, grown on demand and reused
    sina_hip::DevBuf qd, order, rec, node_pos, pred, succ_minpos, qmask, spill, edge, res, weights, out, out_pos, dbg;
    sina_hip::DevBuf prof16, self16;  // --fs-no-graph: match-term tables of a profile batch (sina_hip_graph_batch)
    sina_hip::DevBuf rgain;           // per DAG node: bound on the gain still to come (the DP kernel's row skip, common.h)
    sina_hip::DevBuf scout, scout_u;  // the scout pass (scout.hip): its band rows, and its result -- a bound U per query
    sina_hip::HostBuf h_res;          // pinned copy of a launch's DpResults (row-skip statistics, the next launch's guess)
    bool profile_batch = false;       // the launch being prepared is one (set by sina_hip_align_graphs)
    void *last_tb = nullptr;  // the plane of the last launch (debug read-back: sina_hip_debug_mesh)
    bool dbg_planes = false;  // the launch being prepared is sina_hip_debug_mesh's: its planes are unpacked cell by cell
    bool no_prune = false;    // ... and it asked for a sweep of every row (sina_hip_debug_mesh, prune = 0)
    bool last_scout = false;  // the last launch's bounds came from the scout pass
    uint32_t last_bq = 0, last_prune_step = 0;  // queries / assumed largest step gain of the last launch (sina_hip_debug_dp_info)
    sina_hip::DevBuf k_qoff, k_scores, k_out_ids, k_out_scores, k_out_n, k_tmp0, k_tmp1, k_tmp2;
    sina_hip::DevBuf g_fam_ids, g_fam_off, g_tmp0, g_tmp1, g_tmp2, g_tmp3, g_sizes, g_wtab;
    sina_hip::DevBuf s_qab, s_qoff, s_cand, s_coff, s_out;  // search-stage comparison
    sina_hip::HostBuf h_out, h_out_pos;  // pinned staging for the DP results
    // h_out_pos holds the aligned columns of a whole align call (laid out like the caller's out_pos): a launch
    // range writes at out_pos_base, callers that pass no out_pos read them here (sina_hip_staged_out_pos)
    uint64_t out_pos_base = 0;
    sina_hip::HostBuf h_stage[12];       // pinned staging of the per-batch uploads / downloads (sina_hip::upload)
    float wtab_fs_weight = NAN;  // fs_weight the device weight table was computed for

    size_t lds_budget = 0;  // LDS per DP workgroup; 0 = what keeps the register-limited occupancy (dp_default_lds_budget)

    static constexpr int kNumScratch = 39;
    static_assert(kNumScratch <= 64, "sina_hip_store::cap_hint is too short");
    void scratch(sina_hip::DevBuf **all) {
        sina_hip::DevBuf *list[kNumScratch] = {&qd, &rec, &node_pos, &pred, &succ_minpos, &qmask, &spill, &res,
                                               &weights, &out, &out_pos, &k_qoff, &k_scores, &k_out_ids,
                                               &k_out_scores, &k_out_n, &k_tmp0, &k_tmp1, &k_tmp2, &g_fam_ids,
                                               &g_fam_off, &g_tmp0, &g_tmp1, &g_tmp2, &g_tmp3, &g_sizes, &g_wtab, &order,
                                               &s_qab, &s_qoff, &s_cand, &s_coff, &s_out, &edge, &prof16, &self16, &rgain,
                                               &scout, &scout_u};
        for (int i = 0; i < kNumScratch; i++) all[i] = list[i];
    }
    void publish_hints() {  // after a call: remember how big my buffers had to be
        sina_hip::DevBuf *all[kNumScratch];
        scratch(all);
        for (int i = 0; i < kNumScratch; i++) {  // (a maximum: compare-exchange until mine is in or beaten)
            size_t seen = st->cap_hint[i].load(std::memory_order_relaxed);
            while (all[i]->cap > seen && !st->cap_hint[i].compare_exchange_weak(seen, all[i]->cap, std::memory_order_relaxed)) {
            }
        }
    }
    // Every scratch buffer knows where the store keeps the largest capacity any context needed for it: a buffer
    // that grows goes there in one step (DevBuf::reserve).  Until round 4 a new fork allocated ALL its buffers at
    // those sizes at once -- a context that only ever searches carried 10 GB of aligner scratch, and a fork made
    // in the middle of a run cost a dozen hipMallocs (13 GB; usually a millisecond, sometimes 250: one default
    // bench run in four lost a third of its rate to it).
    void bind_hints() {
        sina_hip::DevBuf *all[kNumScratch];
        scratch(all);
        for (int i = 0; i < kNumScratch; i++) all[i]->hint = &st->cap_hint[i];
    }
    // What one KIND of call uses, brought to the hinted sizes now (sina_hip_prewarm): 0 k-mer search,
    // 1 alignment (DAG build, DP, walk), 2 search-stage comparison.
    int prewarm(int kind) {
        sina_hip::DevBuf *search[] = {&k_qoff, &k_scores, &k_out_ids, &k_out_scores, &k_out_n, &k_tmp0, &k_tmp1, &k_tmp2, &qmask};
        sina_hip::DevBuf *align[] = {&qd, &order, &rec, &node_pos, &pred, &succ_minpos, &qmask, &spill, &edge, &res, &weights, &out,
                                     &out_pos, &g_fam_ids, &g_fam_off, &g_tmp0, &g_tmp1, &g_tmp2, &g_tmp3, &g_sizes, &g_wtab, &rgain, &scout, &scout_u};
        sina_hip::DevBuf *compare[] = {&s_qab, &s_qoff, &s_cand, &s_coff, &s_out};
        sina_hip::DevBuf **list = kind == 0 ? search : (kind == 1 ? align : compare);
        const size_t n = kind == 0 ? sizeof search / sizeof *search : (kind == 1 ? sizeof align / sizeof *align : sizeof compare / sizeof *compare);
        for (size_t i = 0; i < n; i++) {
            const size_t want = list[i]->hint ? list[i]->hint->load(std::memory_order_relaxed) : 0;
            if (want > list[i]->cap && list[i]->reserve_exact(want)) return 1;
        }
        return 0;
    }
    void free_all() {
        if (st && st->wedged.load(std::memory_order_relaxed)) return;  // (see sina_hip_store::wedged: leaked on purpose)
        sina_hip::DevBuf *all[kNumScratch];
        scratch(all);
        for (auto *b : all) b->release();
        dbg.release();
        h_out.release();
        h_out_pos.release();
        h_res.release();
        for (auto &h : h_stage) h.release();
        if (owns_store && st) {
            if (st->heavy) (void)hipStreamDestroy(st->heavy);
            if (st->heavy2) (void)hipStreamDestroy(st->heavy2);
            for (auto &e : st->heavy_done)
                if (e) (void)hipEventDestroy(e);
            if (st->dry_mem) (void)hipFree(st->dry_mem);
            if (st->dry_flag) (void)hipFree(st->dry_flag);
            for (auto &e : st->dp_end)
                if (e) (void)hipEventDestroy(e);
            st->ref_ab.release();
            st->ref_off.release();
            st->idx_off.release();
            st->idx_ids.release();
            st->dense_id.release();
            st->dense_bits.release();
            for (auto &pl : st->tb_pool.plane) pl.release();
            delete st;
        }
        st = nullptr;
    }
};

namespace sina_hip {
// bytes one trace-back plane may have (decided once per store, see sina_hip_store::tb_pool)
inline uint64_t tb_plane_budget(sina_hip_ctx *c) {
    auto &tp = c->st->tb_pool;
    std::lock_guard<std::mutex> lk(tp.mu);
    if (tp.budget == 0) {
        if (const char *e = getenv("SINA_HIP_TB_PLANES")) tp.n = std::max(1, std::min((int)sina_hip_store::tb_pool_t::kMax, atoi(e)));
        const char *gb = getenv("SINA_HIP_TB_GB");
        size_t free_b = 0, total_b = 0;
        if (gb && atof(gb) > 0) {
            tp.budget = (uint64_t)(atof(gb) * 1073741824.0);
        } else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
            tp.budget = (uint64_t)((double)free_b * 0.84 / tp.n);
        } else {
            (void)hipGetLastError();
            tp.budget = (uint64_t)32 << 30;
        }
        if (tp.budget < ((uint64_t)1 << 28)) tp.budget = (uint64_t)1 << 28;
    }
    // ... and never more than the device can give NOW: what is free plus what this pool already holds (a plane that
    // has to grow is freed first).  Two stores on one device, or two ranks sharing a GPU, each decided their budget
    // when most of the memory was free; the one that comes second splits its launches to what is left instead of
    // failing in hipMalloc.  (SINA_HIP_TB_GB still wins: an explicit size is taken as given.)
    // (asked only while the pool is still being built: once every plane exists launches repeat their sizes, and
    // hipMemGetInfo is not a call to make per batch)
    // (asked while the pool is still being built -- and latched per number of planes made so far: with one context in
    // flight the second plane is never made, and a budget that moved with the free memory on every call made the
    // launch sizes stop repeating)
    int made = 0;
    for (int i = 0; i < tp.n; i++) made += tp.plane[i].cap != 0 ? 1 : 0;
    if (made < tp.n && !(getenv("SINA_HIP_TB_GB") && atof(getenv("SINA_HIP_TB_GB")) > 0)) {
        if (tp.now_made != made || tp.now_budget == 0) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                uint64_t held = 0;
                for (int i = 0; i < tp.n; i++) held += tp.plane[i].cap;
                const uint64_t now = (uint64_t)((double)(free_b + held) * 0.84 / tp.n);
                tp.now_budget = std::max<uint64_t>((uint64_t)1 << 28, std::min<uint64_t>(tp.budget, now));
                tp.now_made = made;
            } else {
                (void)hipGetLastError();
                return tp.budget;
            }
        }
        return tp.now_budget;
    }
    return tp.budget;
}
// a free plane of at least `bytes` (waits for one; grows it -- rarely: sizes repeat -- up to the budget)
struct tb_plane_lease {
    sina_hip_store *st = nullptr;
    int idx = -1;
    void *ptr = nullptr;
    int acquire(sina_hip_ctx *c, uint64_t bytes) {
        const uint64_t budget = tb_plane_budget(c);
        auto &tp = c->st->tb_pool;
        std::unique_lock<std::mutex> lk(tp.mu);
        for (;;) {
            int pick = -1;
            for (int i = 0; i < tp.n; i++)  // (prefer one that is big enough already)
                if (!tp.busy[i] && (pick < 0 || (tp.plane[i].cap >= bytes && tp.plane[pick].cap < bytes))) pick = i;
            if (pick >= 0) {
                idx = pick;
                break;
            }
            tp.cv.wait(lk);
        }
        tp.busy[idx] = true;
        st = c->st;
        lk.unlock();
        DevBuf &b = tp.plane[idx];
        if (b.cap < bytes) {
            // (a launch near the budget gets exactly the budget: the next one is no bigger)
            const uint64_t want = std::max<uint64_t>(bytes, std::min<uint64_t>(budget, bytes + bytes / 8));
            if (b.reserve_exact(want)) {
                release();
                return 1;
            }
        }
        ptr = b.p;
        return 0;
    }
    void release() {
        if (!st || idx < 0) return;
        {
            std::lock_guard<std::mutex> lk(st->tb_pool.mu);
            st->tb_pool.busy[idx] = false;
        }
        st->tb_pool.cv.notify_all();
        idx = -1;
        st = nullptr;
    }
    ~tb_plane_lease() { release(); }
};
// The host copy of the reference offsets, complete.  upload_refs fills it; a store that arrived by
// broadcast (sina_hip_store_alloc_like) re-reads it from the device on first use -- exactly once,
// under the store's mutex, whichever forked context gets here first.
inline int ensure_ref_off_host(sina_hip_ctx *c) {
    sina_hip_store *st = c->st;
    if (st->ref_off_host_ready.load(std::memory_order_acquire)) return 0;
    std::lock_guard<std::mutex> lk(st->aux_mu);
    if (st->ref_off_host_ready.load(std::memory_order_relaxed)) return 0;
    st->ref_off_host.assign((size_t)st->n_refs + 1, 0);
    SH_CHECK(hipMemcpy(st->ref_off_host.data(), st->ref_off.p, 8 * ((size_t)st->n_refs + 1), hipMemcpyDeviceToHost));
    st->ref_off_host_ready.store(true, std::memory_order_release);
    return 0;
}
}  // namespace sina_hip

namespace sina_hip {
// DP waves the device holds at once (one wave = one query; mesh_dp_kernel's launch bounds).  A launch
// of exactly that many queries -- or a multiple -- keeps every SIMD at its full wave count until the
// last round; a launch of 4/3 of it runs the last third on a quarter-filled device (measured, 16S:
// 3072 queries 27.7 ms, 4096 41.1 ms, 6144 49.7 ms).
inline uint32_t dp_wave_slots(const sina_hip_ctx *c, int B) {
    const int waves_per_simd = B <= 4 ? 5 : (B <= 8 ? 3 : 2);  // (mesh_dp.hip, dp_default_lds_budget)
    return (uint32_t)(waves_per_simd * 4 * (c->n_cu > 0 ? c->n_cu : 256));
}
// End of a DP launch range that the trace-back budget cut short (q1 < limit): whole rounds of wave
// slots if it holds at least one.
inline uint32_t dp_round_range(uint32_t q0, uint32_t q1, uint32_t limit, uint32_t slots) {
    const uint32_t n = q1 - q0;
    return (q1 < limit && n > slots) ? q0 + n / slots * slots : q1;
}
}  // namespace sina_hip

namespace sina_hip {
// The kernels of this library each fill the device by themselves.  Beside a resident DP kernel (all
// VGPRs of every SIMD taken) the others only get the slots of retiring DP waves and run several
// times longer than alone, while the DP kernel loses those slots for as long: one device-filling
// kernel at a time is faster for both (measured: DP 24 instead of 33 ms per launch inside the
// pipeline).  So every such kernel is queued on the store's one "heavy" stream -- a FIFO the GPU
// works off without the host in between: the kernel waits (event) for what its context queued before
// it; the host thread that queued it waits for its end.  Thin, latency-bound work (copies,
// the backtrack walk) stays on the contexts' streams and runs beside whatever is resident.
// SINA_HIP_SERIALIZE=0 (experiments): every kernel on its own context's stream again, DP launches
// taking turns through a host-side token.
inline bool serialize_kernels() {
    static const bool on = [] {
        const char *v = sina_hip::experiment_env("SINA_HIP_SERIALIZE");
        return !(v && *v == '0');
    }();
    return on;
}
// Waits for an event / for everything queued on a stream of the context WITHOUT burning a core: the
// runtime's hipStreamSynchronize and hipEventSynchronize (hipEventBlockingSync or not) busy-poll, and
// so does its hipLaunchHostFunc helper thread (tools/ubench/wait_cpu.hip: 50 ms of CPU per 50 ms
// waited) -- with seven host threads of a pipeline waiting for the GPU most of the time that was a
// third of the process's CPU time.  Poll-and-sleep instead: the wake-up is at most ~100 us late,
// which only matters if nothing else is queued behind on the heavy stream.
constexpr double kChainTimeoutS = 120.0;
// (timeout_s > 0: gives up with hipErrorNotReady after that long)
inline hipError_t wait_event(hipEvent_t ev, double timeout_s = 0.0) {
    long ns = 50000;  // 50 us, growing to 1 ms: short waits are answered fast, a 17 ms DP kernel costs ~25 wake-ups
    timespec t0;
    if (timeout_s > 0.0) clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (timeout_s > 0.0) {  // (wall time, not the sum of the sleeps asked for)
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s) return hipErrorNotReady;
        }
        timespec ts{0, ns};
        nanosleep(&ts, nullptr);
        if (ns < 1000000) ns += ns / 2;
    }
}
inline hipError_t wait_stream(sina_hip_ctx *c, hipStream_t s) {
    const hipError_t e = hipEventRecord(c->ev[9], s);
    return e != hipSuccess ? e : wait_event(c->ev[9]);
}
// Host <-> device copies of the per-batch arrays go through pinned staging buffers of the context:
// a copy from / to pageable memory makes the runtime pin and unpin the pages (ioctls, TLB shoot-downs
// -- a quarter of a pipeline's CPU time was kernel mode) and wait inside the call.  `slot`: one
// staging buffer per copy of a call (the copies of a call are in flight together).
inline int upload(sina_hip_ctx *c, int slot, void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    if (c->h_stage[slot].reserve(bytes)) return 1;
    memcpy(c->h_stage[slot].p, src, bytes);
    SH_CHECK(hipMemcpyAsync(dst, c->h_stage[slot].p, bytes, hipMemcpyHostToDevice, s));
    return 0;
}
// device -> pinned staging; the caller waits for the stream and then copies out of staged(slot)
inline int download(sina_hip_ctx *c, int slot, const void *src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    if (c->h_stage[slot].reserve(bytes)) return 1;
    SH_CHECK(hipMemcpyAsync(c->h_stage[slot].p, src, bytes, hipMemcpyDeviceToHost, s));
    return 0;
}
// What a device-filling kernel needs to tell the queue that its last workgroup has been dispatched:
// every workgroup calls dry_signal() (common.h) first thing.
// (declared in common.h: struct DryArgs { uint32_t *counter, *flag; uint32_t seq; })
//
// Round 4 -- chained launches.  A DP launch ends with its last waves finishing on a device that empties SIMD
// by SIMD (4.4 ms of a 48 ms launch), and until round 3 the next kernel of the FIFO waited
// for the very last of them.  Now the FIFO alternates between two streams, and a launch waits
//   * for what its own context queued before it (event, as before),
//   * for the END of the launch two places ahead (stream order: same stream),
//   * for the launch right ahead of it only to have RUN DRY -- hipStreamWaitValue32 on a flag word that
//     launch's last-started workgroup writes (tools/ubench/chain.hip: the follower starts 0.3 ms after the
//     flag, i.e. as soon as a slot is free) -- if that launch signals it (DP, DAG build), else for its end.
// So at most two device-filling kernels are resident, and only while the older one has nothing left to
// dispatch: the newer one gets exactly the slots the drain leaves empty.  Nothing runs beside a kernel
// that still has workgroups waiting (what round 1 measured as mutual stretching).  Every wait refers to
// work queued EARLIER, so hardware queues shared between streams cannot dead-lock (see done()).
// SINA_HIP_CHAIN=0: one stream, every launch waits for the end of the one before (round 3).
// Under a profiler that collects hardware counters (rocprofv3 --pmc: ROCPROF_COUNTER_COLLECTION in the environment)
// chaining is off unless SINA_HIP_CHAIN=1 insists: counter collection runs ONE kernel at a time across all queues, in
// the order the tool intercepts them -- the runtime's wait-value poller of a follower can be let in before the DP
// launch it waits for (that launch still being held by its own upload event), and then nothing ever runs again (one
// of three PMC passes hung this way in round 4).  Per-kernel counters do not depend on what a kernel was chained to.
inline bool chain_kernels() {
    static const bool on = [] {
        const char *v = getenv("SINA_HIP_CHAIN");
        if (v && *v) return *v != '0';
        const char *pmc = getenv("ROCPROF_COUNTER_COLLECTION");
        return !(pmc && *pmc && *pmc != '0');
    }();
    return on;
}
// Which launch goes next.  Until round 4 the FIFO was the order in which host threads arrived.  With chained
// launches the order matters, because of what runs in a DP launch's last milliseconds and right behind it:
//   * the launch behind it fills the slots its drain leaves empty -- but starts its own waves staggered by as
//     much, and ends with a drain that wide (a DP wave runs three 16 ms queries back to back: nothing evens a
//     5 ms stagger out).  A DAG build or a k-mer search, made of thousands of short workgroups, absorbs that;
//     a DP launch hands it on to whatever follows it;
//   * the finished launch's backtrack walk starts on its context's own stream the moment it ends and runs beside
//     whatever is resident then: it stretches a DAG build by 1.8 ms and a DP launch by 8 (its waves take the LDS
//     of retiring DP waves).
// Kernel trace of a run that PREFERRED a DP launch behind a DP launch (profiles/r04_burst_trace.txt): two DP
// launches back to back took 102 ms, 2 x 48.3 alone.  So at most kHeavyDepth launches are handed to the GPU
// ahead of time (one running, one queued behind it -- enough to never leave the device idle: every launch
// runs for milliseconds), the other callers wait on the host, and when a place frees up the waiter to take it
// is: behind a DP launch the oldest waiting DAG build, else the oldest k-mer search, and only if neither is
// waiting another DP launch; behind anything else the oldest waiter.
// SINA_HIP_DP_BURST: 0 = arrival order, 1 = DP behind DP preferred (the experiment above), default -1.
enum heavy_kind { kHeavyKmer = 0, kHeavyGraph = 1, kHeavyDp = 2 };
constexpr int kHeavyDepth = 2;
inline int heavy_order_policy() {
    static const int mode = [] {
        const char *v = experiment_env("SINA_HIP_DP_BURST");
        return v && *v ? atoi(v) : -1;
    }();
    return mode;
}
struct heavy_launch {
    sina_hip_ctx *c;
    hipStream_t own, hs;
    std::unique_lock<std::mutex> lk;
    bool failed = false;
    int turn = 0;
    int kind = 0;
    bool signals_dry = false;
    bool admitted = false;
    bool chained = false;  // this launch went through the two-stream FIFO (else: the single heavy stream, or the context's own)
    uint32_t my_seq = 0;
    // `own`: the context stream whose queued work (uploads) the kernel depends on
    heavy_launch(sina_hip_ctx *c_, hipStream_t own_, int kind_ = kHeavyKmer) : c(c_), own(own_), hs(own_), kind(kind_) {
        if (!serialize_kernels() || !c->st->heavy) return;
        sina_hip_store *st = c->st;
        if (st->wedged.load(std::memory_order_relaxed)) {
            failed = true;
            return;
        }
        failed = hipEventRecord(c->ev[10], own) != hipSuccess;
        lk = std::unique_lock<std::mutex>(st->heavy_mu);
        const bool chain = chain_kernels() && st->heavy2 && st->dry_mem && !st->chain_broken.load(std::memory_order_relaxed);
        chained = chain;
        if (chain) {
            const uint64_t me = st->heavy_ticket++;
            st->heavy_waiting.push_back({me, kind});
            st->heavy_cv.wait(lk, [&] {
                if (st->heavy_outstanding >= kHeavyDepth) return false;
                // whose turn is it?
                const sina_hip_store::heavy_waiter *pick = nullptr;
                auto oldest_of = [&](int k) {
                    const sina_hip_store::heavy_waiter *o = nullptr;
                    for (const auto &w : st->heavy_waiting)
                        if ((k < 0 || w.kind == k) && (!o || w.ticket < o->ticket)) o = &w;
                    return o;
                };
                if (st->heavy_last_kind == kHeavyDp && heavy_order_policy() > 0) {
                    pick = oldest_of(kHeavyDp);
                    if (!pick) pick = oldest_of(kHeavyGraph);
                } else if (heavy_order_policy() < 0) {
                    // (round 6: a DAG build, then a k-mer search, goes before a DP launch WHENEVER one waits, not only
                    // behind a DP launch: a batch's scout pass runs between its DAG build and its DP launch, on the
                    // context's own stream -- with the build scheduled right in front of the batch's own DP launch the
                    // FIFO sat idle for the scout's 4.5 ms every step (profiles/r06_heavy_gaps_before.txt); built a DP
                    // launch earlier, the scout runs beside the DP launch of the batch before)
                    pick = oldest_of(kHeavyGraph);
                    if (!pick) pick = oldest_of(kHeavyKmer);
                }
                if (!pick) pick = oldest_of(-1);
                return pick && pick->ticket == me;
            });
            for (size_t i = 0; i < st->heavy_waiting.size(); i++)
                if (st->heavy_waiting[i].ticket == me) {
                    st->heavy_waiting.erase(st->heavy_waiting.begin() + (std::ptrdiff_t)i);
                    break;
                }
            ++st->heavy_outstanding;
            st->heavy_last_kind = kind;
            admitted = true;
        }
        turn = chain ? st->heavy_turn : 0;
        hs = turn ? st->heavy2 : st->heavy;
        failed = failed || hipStreamWaitEvent(hs, c->ev[10], 0) != hipSuccess;
        if (chain && st->heavy_prev_any) {
            if (st->heavy_prev_dry)
                failed = failed || hipStreamWaitValue32(hs, st->dry_flag, st->heavy_prev_seq, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess;
            else
                failed = failed || hipStreamWaitEvent(hs, st->heavy_done[turn ^ 1], 0) != hipSuccess;
        }
        if (chain) my_seq = ++st->heavy_seq;
    }
    hipStream_t stream() const { return hs; }
    // for the LAST kernel of the launch, if it can tell when its last workgroup starts
    DryArgs dry() {
        DryArgs d{nullptr, nullptr, 0};
        sina_hip_store *st = c->st;
        if (!lk.owns_lock() || !chained) return d;
        d.flag = st->dry_flag;
        d.counter = st->dry_mem + my_seq % sina_hip_store::kDryCounters;
        d.seq = my_seq;
        signals_dry = true;
        return d;
    }
    // After the launches: the HOST waits for them.  (Never let a context stream wait for the heavy
    // stream on the GPU side: streams share hardware queues once there are more streams than queues,
    // a queue blocked on the heavy stream would hold up another context's uploads that an EARLIER
    // heavy kernel is waiting for -- a deadlock, seen with five batches in flight.  The heavy stream
    // waits for context streams, context streams wait for nothing but themselves.)
    int done() {
        if (hs == own) return failed ? 1 : 0;
        failed = failed || hipEventRecord(c->ev[11], hs) != hipSuccess;
        if (lk.owns_lock()) {
            sina_hip_store *st = c->st;
            if (chained) {
                failed = failed || hipEventRecord(st->heavy_done[turn], hs) != hipSuccess;
                st->heavy_prev_dry = signals_dry && !failed;
                st->heavy_prev_seq = my_seq;
                st->heavy_prev_any = true;
                st->heavy_turn = turn ^ 1;
            }
            lk.unlock();
            c->st->heavy_cv.notify_all();
        }
        // A tool that serialises kernels across queues in the order it intercepts them (a counter-collecting profiler
        // is the one known: chain_kernels()) can let the runtime's wait-value poller in before the launch it waits
        // for, and then nothing runs again.  A chained launch that has not ended after kChainTimeoutS is taken for
        // that: the call fails with a message that names the cause, and the store stops chaining.
        const hipError_t we = wait_event(c->ev[11], chained ? kChainTimeoutS : 0.0);
        if (we == hipErrorNotReady) {
            c->st->chain_broken.store(true, std::memory_order_relaxed);
            c->st->wedged.store(true, std::memory_order_relaxed);
            leave();
            set_error("heavy_launch: a chained launch did not end within 120 s -- a tool that serialises kernels across "
                      "queues? (set SINA_HIP_CHAIN=0); the store takes no further launches, its buffers stay allocated");
            return 1;
        }
        failed = failed || we != hipSuccess;
        leave();
        if (failed) set_error("heavy_launch: event hand-over failed");
        return failed ? 1 : 0;
    }
    // the launch has ended (or was never made): its place goes to a waiter
    void leave() {
        if (!admitted) return;
        admitted = false;
        sina_hip_store *st = c->st;
        if (!lk.owns_lock()) lk = std::unique_lock<std::mutex>(st->heavy_mu);
        --st->heavy_outstanding;
        lk.unlock();
        st->heavy_cv.notify_all();
    }
    ~heavy_launch() { leave(); }
};
}  // namespace sina_hip

// publishes the context's scratch capacities when an API call ends (see sina_hip_store::cap_hint)
struct sina_hip_hint_guard {
    sina_hip_ctx *c;
    explicit sina_hip_hint_guard(sina_hip_ctx *c) : c(c) {}
    ~sina_hip_hint_guard() { c->publish_hints(); }
};
