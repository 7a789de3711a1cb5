// C-ABI entry points of include/sina_hip.h: context, reference store, alignment.
// (k-mer entry points live in kmer.hip, the device DAG build in graph_build.hip.)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "ctx.h"

namespace sina_hip {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

int plan_dp(sina_hip_ctx *c, uint32_t maxL, DpPlan *pl) {
    if (!pick_geom(maxL, &pl->geom)) SH_FAIL("align: query longer than SINA_HIP_MAX_QUERY_LEN bases");
    const size_t slot = dp_slot_bytes(pl->geom), fixed = dp_fixed_lds_bytes(pl->geom);
    size_t budget = c->lds_budget ? c->lds_budget : dp_default_lds_budget(pl->geom);
    if (budget < fixed + slot) budget = fixed + slot;
    if (budget > 160 * 1024) budget = 160 * 1024;
    int W = (int)((budget - fixed) / slot);
    if (W < 1) SH_FAIL("align: LDS cannot hold one DP row");
    W = std::min(W, dp_max_ring(pl->geom));
    pl->W = W;
    pl->lds = fixed + (size_t)W * slot;
    return 0;
}

// SINA_HIP_DP_PRUNE=0 switches the DP kernel's certified row skip off (every row of every strip is swept);
// SINA_HIP_TEST="rho=<x>" fixes the launches' guess of optimum / bound (tests: 2 = too bold for any query, every
// query takes the second attempt; 0.01 = nearly no bound).  Read per launch.
PrunePlan prune_plan(const sina_hip_align_params *p, float wmax, float wmin, uint32_t maxL, bool profile_batch) {
    PrunePlan pp;
    if (!std::isfinite(wmax) || !std::isfinite(wmin)) return pp;  // (a NaN weight: no bound holds)
    const char *off = getenv("SINA_HIP_DP_PRUNE");
    if (off && off[0] == '0') return pp;
    if (profile_batch || (p->weights != nullptr && p->n_weights > 0) || p->insertion == SINA_INSERTION_FORBID) return pp;
    // gaps must cost, node weights must not be negative (a match gains match_score * weight, nothing else gains)
    if (!(p->gap_penalty >= 0.f) || !(p->gap_ext_penalty >= 0.f) || !(wmin >= 0.f) || !std::isfinite(wmax)) return pp;
    const float kappa = std::max(0.f, std::max(p->match_score, p->mismatch_score));
    if (!std::isfinite(kappa)) return pp;
    pp.kappa64 = 64.0f * 1.0001f * kappa;
    pp.amax = prune_gain_units(wmax, pp.kappa64);
    // (the bounds are exact float32 integers in units of 1/64 only below 2^24)
    if (pp.amax > 250u || (uint64_t)pp.amax * maxL >= (1u << 23)) return pp;
    if (pp.kappa64 <= 0.f) pp.kappa64 = 1e-30f;  // (no step gains anything: every node's gain is the one unit of margin)
    // (a launch whose queries fit ONE strip skips nothing -- column 0 keeps every row in play -- so nobody needs the
    // bound: the DAG build leaves its step 9 out, 9 % of its time for V4 amplicons)
    DpGeom g;
    if (pick_geom(maxL, &g) && g.T <= 64) return pp;
    pp.on = 1;
    return pp;
}

// Host-side preparation of a range of host-built graphs: descriptors, row records
// (sink flag, spill slot for rows with a successor further than W rows away).
struct HostPrep {
    std::vector<QDesc> qd;
    std::vector<uint4> rec;
    std::vector<uint2> rgain;     // the DP kernel's row-skip bound per node + its last successor (common.h), filled when kappa64 > 0
    bool rgain_ok = true;         // ... and valid: every DAG of the range is laid out by columns
    float wmax = 0.f, wmin = 0.f; // node weights of the range
    std::vector<uint32_t> pred;  // id | (LDS slot or spill row) << 16 | spilled << 31 (what mesh_dp_kernel reads)
    std::vector<uint32_t> last;  // scratch: last successor per row
    uint64_t tb_cells = 0, spill_rows = 0, cells = 0;
};

static int prep_range(const sina_hip_graph_batch *g, const uint64_t *qoff, uint32_t q0, uint32_t q1, int Lp, int W,
                       HostPrep *hp, float kappa64) {
    const uint64_t nbase = g->node_off[q0], ebase = g->edge_off[q0];
    const uint64_t nn = g->node_off[q1] - nbase;
    hp->qd.resize(q1 - q0);
    hp->rec.resize(nn);
    hp->rgain.assign(kappa64 > 0.f ? nn : 0, uint2{0u, 0u});
    hp->rgain_ok = true;
    hp->pred.resize(g->edge_off[q1] - ebase + 8);
    hp->tb_cells = hp->spill_rows = hp->cells = 0;
    uint32_t erec_cursor = 0;
    for (uint32_t q = q0; q < q1; q++) {
        QDesc &d = hp->qd[q - q0];
        const uint64_t no = g->node_off[q], eo = g->edge_off[q];
        const uint32_t N = (uint32_t)(g->node_off[q + 1] - no);
        d.node_off = no - nbase;
        d.erec_off = erec_cursor;
        erec_cursor += dp_edge_entries(N);
        d.edge_off = eo - ebase;
        d.q_off = qoff[q] - qoff[q0];
        d.tb_off = hp->tb_cells;
        d.spill_off = hp->spill_rows;
        d.N = N;
        d.L = (uint32_t)(qoff[q + 1] - qoff[q]);
        const uint32_t *po = g->pred_off + no + q;  // N+1 entries, relative to eo
        uint4 *rec = hp->rec.data() + d.node_off;
        for (uint32_t m = 0; m < N; m++) {
            // (what the row record and the kernel's topological sweep can represent: fail, do not truncate)
            if (po[m + 1] < po[m] || po[m + 1] - po[m] > 255u)
                SH_FAIL("align_graphs: a node has more than 255 predecessors (or pred_off is not ascending)");
            for (uint32_t e = po[m]; e < po[m + 1]; e++)
                if (g->pred[eo + e] >= m) SH_FAIL("align_graphs: predecessor ids must be smaller than the node's id");
            uint32_t wbits;
            memcpy(&wbits, &g->node_weight[no + m], 4);
            rec[m].x = po[m];
            rec[m].y = wbits;
            rec[m].z = ((po[m + 1] - po[m]) & 0xffu) | ((uint32_t)(g->node_mask[no + m] & 0xffu) << 8) | kRecSink;
            rec[m].w = kRowNone;
        }
        // last successor of every row (0 = none), sink and fence flags
        std::vector<uint32_t> &last = hp->last;
        last.assign(N, 0);
        for (uint32_t m = 0; m < N; m++) {
            for (uint32_t e = po[m]; e < po[m + 1]; e++) {
                const uint32_t p = g->pred[eo + e];
                rec[p].z &= ~kRecSink;
                last[p] = m;  // rows ascend
                if (m - p > (uint32_t)kFarLds) rec[p].z |= kRecFence;
            }
        }
        d.first_sink = 0;
        d.gmin = 0;
        for (uint32_t m = 0; m < N; m++)
            if (rec[m].z & kRecSink) {
                d.first_sink = m;
                break;
            }
        // LDS slots by liveness, first free slot wins; a row that finds none is spilled.  Rows are
        // allocated in independent segments (dp_slot_segment, common.h), like the device DAG build does.
        uint32_t nsp = 0;
        uint32_t free_at[64];
        const uint32_t seg_len = dp_slot_segment(N);
        for (uint32_t m = 0; m < N; m++) {
            if (m % seg_len == 0)
                for (int x = 0; x < W; x++) free_at[x] = 0;
            if (rec[m].z & kRecSink) continue;  // w stays kRowNone
            if (last[m] == m + 1) continue;      // only the next row reads it: handed over in registers
            int slot = -1;
            const uint32_t seg_end = std::min<uint32_t>(N, (m / seg_len + 1) * seg_len);
            if (!(rec[m].z & kRecFence) && last[m] < seg_end)  // (else: always a spill row)
                for (int x = 0; x < W; x++)
                    if (free_at[x] <= m) {
                        slot = x;
                        break;
                    }
            if (slot >= 0) {
                free_at[slot] = last[m];
                rec[m].w = (uint32_t)slot;
            } else {
                rec[m].w = kRowSpilled | nsp++;
            }
        }
        for (uint32_t m = 0; m < N; m++) {
            uint32_t first_far = 0;
            uint32_t dist = po[m + 1] > po[m] ? 0u : kRecDistFar;
            for (uint32_t e = po[m]; e < po[m + 1]; e++) {
                const uint32_t p = g->pred[eo + e];
                const uint32_t pw = rec[p].w == kRowNone ? 0u : rec[p].w;  // (kRowNone: in registers for this row)
                const bool sp = (pw & kRowSpilled) != 0;
                if (sp && first_far == 0) first_far = e - po[m] + 1;
                dist = std::max(dist, m - p);
                hp->pred[d.edge_off + e] = p | ((pw & 0x7FFFu) << 16) | (sp ? kPredSpilled : 0u);
            }
            rec[m].z |= (first_far << 24) | (std::min(dist, kRecDistFar) << kRecDistShift);
        }
        if (nsp > kMaxSpillRows) SH_FAIL("align_graphs: too many spill rows for one query");
        // The row-skip bound, as the device DAG build computes it (graph_build.hip step 9): R(m) = the sum, over the
        // columns right of node m's, of the column's best node's gain.  It is a bound only for a DAG laid out like
        // mseq's -- columns ascend with the node ids, every edge leads to a column further right --, which a caller's
        // arrays need not be: checked here, and a launch holding a DAG that is not runs without the skip.
        if (kappa64 > 0.f) {
            uint2 *rg = hp->rgain.data() + d.node_off;
            const uint32_t *pos = g->node_pos + no;
            bool ok = true;
            for (uint32_t m = 0; m < N && ok; m++) {
                if (m > 0 && pos[m] < pos[m - 1]) ok = false;
                for (uint32_t e = po[m]; e < po[m + 1] && ok; e++)
                    if (pos[g->pred[eo + e]] >= pos[m]) ok = false;
            }
            if (!ok) hp->rgain_ok = false;
            uint32_t right = 0, cols_right = 0, gmin = 0xFFFFFFFFu;  // columns right of the one being finished
            for (uint32_t m = N; ok && m > 0;) {
                uint32_t first = m - 1, mx = 0;
                while (first > 0 && pos[first - 1] == pos[m - 1]) first--;
                for (uint32_t j = first; j < m; j++) {
                    mx = std::max(mx, prune_gain_units(g->node_weight[no + j], kappa64));
                    rg[j] = uint2{right, last[j] | (cols_right << 16)};
                }
                right += mx;
                cols_right++;
                gmin = std::min(gmin, mx);
                m = first;
            }
            d.gmin = ok ? gmin : 0u;
        }
        d.n_spill = nsp;
        hp->spill_rows += nsp;
        hp->tb_cells += (uint64_t)N * Lp;
        hp->cells += (uint64_t)N * d.L;
    }
    return 0;
}

// Runs DP + backtrack for bq queries whose graphs (qd, rec, pred, node_pos, succ_minpos) and
// query masks are already in the context's device buffers; copies results back.
int run_dp_device(sina_hip_ctx *c, const DpPlan &pl, const QDesc *qd_host, uint32_t bq, uint64_t n_node_entries,
                  uint64_t tb_cells, uint64_t spill_rows, uint64_t cells, uint64_t nqm, const sina_hip_align_params *p, uint32_t width,
                  sina_hip_align_out *out, uint32_t *out_pos, bool want_dbg_value, const PrunePlan &pp,
                  const uint32_t *chain_ref) {
    hipStream_t s = c->stream;
    const int Lp = pl.geom.Lp();
    const bool weighted = p->weights != nullptr && p->n_weights > 0;
    const bool forbid = p->insertion == SINA_INSERTION_FORBID;
    // edge records per strip boundary: every query's region starts on a 64-byte line (common.h, EdgeRec);
    // the callers laid the same offsets into QDesc::erec_off
    uint64_t edge_entries = 0;
    for (uint32_t q = 0; q < bq; q++) edge_entries += dp_edge_entries(qd_host[q].N);
    (void)n_node_entries;
    if (c->spill.reserve(std::max<uint64_t>(spill_rows, 1) * 8 * (uint64_t)Lp) ||
        c->edge.reserve(std::max<uint64_t>(1, (uint64_t)(pl.geom.T / 64 - 1) * edge_entries) * sizeof(EdgeRec)) ||
        c->res.reserve(sizeof(DpResult) * bq) || c->out.reserve(sizeof(sina_hip_align_out) * bq) ||
        c->out_pos.reserve(4 * std::max<uint64_t>(nqm, 1)))
        return 1;
    if (want_dbg_value && c->dbg.reserve(4 * tb_cells)) return 1;
    // longest queries first (workgroups start in index order; see mesh_dp_kernel)
    std::vector<uint32_t> order(bq);
    for (uint32_t q = 0; q < bq; q++) order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        return (uint64_t)qd_host[x].N * qd_host[x].L > (uint64_t)qd_host[y].N * qd_host[y].L;
    });
    if (c->order.reserve(4 * (size_t)bq)) return 1;
    if (upload(c, 0, c->order.p, order.data(), 4 * (size_t)bq, s)) return 1;
    DpArgs a;
    a.dry = DryArgs{nullptr, nullptr, 0};
    a.qd = c->qd.as<QDesc>();
    a.order = c->order.as<uint32_t>();
    a.rec = c->rec.as<uint4>();
    a.pred = c->pred.as<uint32_t>();
    a.node_pos = c->node_pos.as<uint32_t>();
    a.succ_minpos = c->succ_minpos.as<uint32_t>();
    a.qmask = c->qmask.as<uint8_t>();
    a.tb = nullptr;  // (the plane is borrowed below, once the scout is back)
    a.dbg_value = want_dbg_value ? c->dbg.as<float>() : nullptr;
    a.spill = c->spill.as<float>();
    a.edge = c->edge.as<EdgeRec>();
    a.edge_stride = edge_entries;
    a.res = c->res.as<DpResult>();
    a.weights = weighted ? c->weights.as<float>() : nullptr;
    a.n_weights = weighted ? p->n_weights : 0;
    a.ms = -p->match_score;  // scoring_scheme_*(-match, -mismatch, gap, gapext), align.cpp:406-414
    a.mms = -p->mismatch_score;
    a.gp = p->gap_penalty;
    a.gpe = p->gap_ext_penalty;
    a.prof16 = c->profile_batch ? c->prof16.as<float>() : nullptr;
    // certified row skip (mesh_dp.hip): the guess the launch's queries start with -- what the store has learnt
    // from the queries before, or SINA_HIP_TEST=rho=<x>
    c->last_bq = bq;
    c->last_prune_step = pp.on ? pp.amax : 0u;
    a.reach = pp.on ? c->rgain.as<uint2>() : nullptr;
    a.prune = pp.on;
    a.prune_amax = pp.amax;
    a.prune_rho = 0.f;
    bool rho_fixed = false;
    if (pp.on) {
        if (const std::string r = test_knob("rho"); !r.empty()) {
            a.prune_rho = (float)atof(r.c_str());
            rho_fixed = a.prune_rho > 0.f;
        }
        if (!rho_fixed) {  // (the guess a launch without a scout starts from; one WITH a scout takes the guard, below)
            std::lock_guard<std::mutex> slk(c->st->stats_mu);
            a.prune_rho = c->st->prune_rho;
        }
    }
    {
        uint32_t max_n = 0;
        for (uint32_t q = 0; q < bq; q++) max_n = std::max<uint32_t>(max_n, qd_host[q].N);
        a.below_init = (!weighted && !forbid && dp_below_init(max_n, a.gp, a.gpe)) ? 1 : 0;
    }
    // The scout pass (scout.hip): every query's own bound U -- the cost of a real path, its alignment against the chain
    // of its family's first member -- instead of the store's guess alone.  On the context's own stream (150 waves, a
    // lane per query: not a device-filling kernel), before the launch borrows a trace-back plane and asks for its place
    // in the FIFO.  A fixed guess (SINA_HIP_TEST=rho=) or SINA_HIP_TEST=scout=0 leaves it out, and so does a caller
    // that brought its own DAGs (sina_hip_align_graphs: no family to take a chain from).
    a.scout_u = nullptr;
    a.scout_bias = (float)atof(test_knob("scout_add").c_str());
    c->last_scout = false;
    if (const std::string fixed = test_knob("scout_set"); !fixed.empty() && pp.on && !rho_fixed) {
        // (test hook: every query's scout value is this number -- lets a caller-built DAG, which has no family to take
        // a chain from, run under a chosen bound: tests/test_gpu_prune.py)
        std::vector<float> vals(bq, (float)atof(fixed.c_str()));
        if (c->scout_u.reserve(4 * (size_t)bq)) return 1;
        if (upload(c, 7, c->scout_u.p, vals.data(), 4 * (size_t)bq, s)) return 1;
        a.scout_u = c->scout_u.as<float>();
        c->last_scout = true;
    } else if (chain_ref != nullptr && pp.on && !rho_fixed && a.below_init && a.gp >= a.gpe && pl.geom.T > 64 && test_knob("scout") != "0") {
        if (c->scout.reserve(4 * (size_t)bq) || c->scout_u.reserve(4 * (size_t)bq)) return 1;
        if (upload(c, 7, c->scout.p, chain_ref, 4 * (size_t)bq, s)) return 1;
        SH_CHECK(hipEventRecord(c->ev[3], s));
        if (launch_chain_scout(a, bq, c->st->ref_ab.as<uint32_t>(), c->st->ref_off.as<uint64_t>(), c->scout.as<uint32_t>(),
                               c->scout_u.as<float>(), s))
            return 1;
        SH_CHECK(hipEventRecord(c->ev[4], s));
        SH_CHECK(wait_event(c->ev[4]));
        float sms = 0;
        SH_CHECK(hipEventElapsedTime(&sms, c->ev[3], c->ev[4]));
        {
            std::lock_guard<std::mutex> slk(c->st->stats_mu);
            c->st->stats.scout_ms += sms;
            c->st->stats.scout_launches++;
        }
        a.scout_u = c->scout_u.as<float>();
        c->last_scout = true;
    }
    if (a.scout_u != nullptr && !rho_fixed) {
        std::lock_guard<std::mutex> slk(c->st->stats_mu);
        a.prune_rho = c->st->prune_rho_guard;
    }
    // The trace-back plane is the one buffer whose size follows the batch (tens of GB for 16S): borrowed
    // from the device's pool of two (ctx.h) until this launch's results are on the host -- and not before the scout
    // is back: a launch waiting for its scout does not hold a plane another launch could fill meanwhile.
    const uint64_t tb_bytes = tb_cell_bytes(forbid) * tb_cells;
    tb_plane_lease plane;
    if (plane.acquire(c, std::max<uint64_t>(tb_bytes, 16))) return 1;
    c->last_tb = plane.ptr;
    a.tb = plane.ptr;
    // (rows the kernel never visits show the value a skipped row shows its successors)
    if (want_dbg_value && pp.on) SH_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->dbg.p), 0x49742400 /* 1e6f */, tb_cells, s));
    // (debug read-back of the planes: rows the kernel skips leave their trace-back cells unwritten -- "untouched cell"
    // everywhere first, so that unpacking them stays inside the DAG)
    if (c->dbg_planes && !forbid) SH_CHECK(hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(plane.ptr), (unsigned short)kTbNone, tb_cells, s));

    BtArgs b;
    b.qd = a.qd;
    b.rec = a.rec;
    b.pred = a.pred;
    b.node_pos = a.node_pos;
    b.tb = a.tb;
    b.res = a.res;
    b.weights = a.weights;
    b.n_weights = a.n_weights;
    b.out = c->out.as<sina_hip_align_out>();
    b.out_pos = c->out_pos.as<uint32_t>();
    b.nq = bq;
    b.width = width;
    b.Lp = (uint32_t)Lp;
    b.ms = a.ms;
    b.overhang = p->overhang;
    b.lazy_sidx = forbid ? 0 : 1;
    b.qmask = a.qmask;
    b.lowercase = p->lowercase;
    b.self16 = c->profile_batch ? c->self16.as<float>() : nullptr;
    b.asm_cap = 0;
    for (uint32_t q = 0; q < bq; q++) b.asm_cap = std::max<uint32_t>(b.asm_cap, qd_host[q].L);
    uint64_t dp_no = ~0ull;  // this launch's number among the store's DP launches
    // Where the walk runs.  Until round 4: on the context's own stream, launched by the host once it had seen the
    // DP kernel end -- beside whatever device-filling kernel was resident by then.  A kernel trace of round 4
    // showed that "by then" can be late: a walk whose hardware queue shares a dispatch pipe with the FIFO's queue
    // is not dispatched before the resident kernel has handed out its last workgroup, starts 5 ms late, runs
    // beside the NEXT DP launch instead of the DAG build behind its own, takes 17 ms there instead of 3 and
    // stretches that launch by 10 (profiles/r04_bt_delay.txt).  With chained launches the walk and the assembly
    // are queued right behind their DP kernel on the same FIFO stream: they start the moment it ends, run beside
    // the launch that started in its drain (the other FIFO stream), and the launch after that -- often the next DP
    // launch -- is ordered behind them by the stream itself.  SINA_HIP_BT_ON_FIFO=0: the context's stream.
    static const bool bt_fifo_wanted = !(experiment_env("SINA_HIP_BT_ON_FIFO") && experiment_env("SINA_HIP_BT_ON_FIFO")[0] == '0');
    bool bt_done = false;
    {
        // the DP kernel: on the store's heavy stream, behind the uploads queued on c->stream; the
        // backtrack walk and the result copies then follow it on the context's low-priority stream
        std::unique_lock<std::mutex> token(c->st->dp_token, std::defer_lock);
        if (!serialize_kernels()) token.lock();  // (then: one DP kernel at a time by this token)
        SH_CHECK(hipEventRecord(c->ev[8], s));
        s = c->stream_dp;
        SH_CHECK(hipStreamWaitEvent(s, c->ev[8], 0));
        heavy_launch hl(c, s, kHeavyDp);
        SH_CHECK(hipEventRecord(c->ev[0], hl.stream()));
        a.dry = hl.dry();
        if (launch_mesh_dp(pl.geom, weighted, forbid, a, bq, pl.lds, hl.stream())) return 1;
        SH_CHECK(hipEventRecord(c->ev[1], hl.stream()));
        if (hl.lk.owns_lock() && c->st->dp_end[0]) {  // (under the queue's lock: launch order = dp_seq order)
            dp_no = c->st->dp_seq++;
            SH_CHECK(hipEventRecord(c->st->dp_end[dp_no % 8], hl.stream()));
            c->st->dp_end_no[dp_no % 8].store(dp_no, std::memory_order_release);
        }
        if (hl.chained && bt_fifo_wanted) {
            if (launch_backtrack(b, hl.stream())) return 1;
            if (p->assemble && launch_assemble(b, hl.stream())) return 1;
            SH_CHECK(hipEventRecord(c->ev[2], hl.stream()));
            bt_done = true;
        }
        if (hl.done()) return 1;
        if (experiment_env("SINA_HIP_DEBUG_SYNC")) fprintf(stderr, "[sina_hip] DP kernel done: %u queries, geometry %dx%d, weighted %d forbid %d\n", bq, pl.geom.T, pl.geom.B, (int)weighted, (int)forbid);
        if (token.owns_lock()) SH_CHECK(wait_event(c->ev[1]));
    }
    if (!bt_done) {
        if (launch_backtrack(b, s)) return 1;
        if (p->assemble && launch_assemble(b, s)) return 1;
        if (experiment_env("SINA_HIP_DEBUG_SYNC")) {
            SH_CHECK(hipStreamSynchronize(s));
            fprintf(stderr, "[sina_hip] backtrack kernel done\n");
        }
        SH_CHECK(hipEventRecord(c->ev[2], s));
    }
    // (h_out_pos was sized for the whole call by the entry point; this range's columns go to their place in it)
    if (c->h_out.reserve(sizeof(sina_hip_align_out) * bq) || c->h_res.reserve(sizeof(DpResult) * bq)) return 1;
    unsigned char *staged_pos = static_cast<unsigned char *>(c->h_out_pos.p) + 4 * c->out_pos_base;
    SH_CHECK(hipMemcpyAsync(c->h_out.p, c->out.p, sizeof(sina_hip_align_out) * bq, hipMemcpyDeviceToHost, s));
    SH_CHECK(hipMemcpyAsync(c->h_res.p, c->res.p, sizeof(DpResult) * bq, hipMemcpyDeviceToHost, s));
    SH_CHECK(hipMemcpyAsync(staged_pos, c->out_pos.p, 4 * nqm, hipMemcpyDeviceToHost, s));
    SH_CHECK(wait_stream(c, s));
    memcpy(out, c->h_out.p, sizeof(sina_hip_align_out) * bq);
    if (out_pos) memcpy(out_pos, staged_pos, 4 * nqm);
    float ms = 0;
    SH_CHECK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    // ... of which this launch shared the device with the DP launch before it (chained launches, ctx.h)
    float shared = 0;
    // (the ring has eight slots and this thread reads it outside the queue's lock: a slot that has been re-recorded by
    // launch dp_no + 7 meanwhile is not the predecessor's any more -- its number says so -- and counts as no overlap,
    // as does a predecessor that has not ended yet: a short launch can end before the drain of the one before it)
    if (dp_no != ~0ull && dp_no > 0 && c->st->dp_end_no[(dp_no - 1) % 8].load(std::memory_order_acquire) == dp_no - 1) {
        float to_prev_end = 0;
        const hipError_t e = hipEventElapsedTime(&to_prev_end, c->ev[0], c->st->dp_end[(dp_no - 1) % 8]);
        if (e == hipSuccess && c->st->dp_end_no[(dp_no - 1) % 8].load(std::memory_order_acquire) == dp_no - 1)
            shared = std::min(ms, std::max(0.f, to_prev_end));
        else (void)hipGetLastError();
    }
    // what the launch actually swept (certified row skip), and what its queries say about the next launch's guess
    const uint32_t kstrip = 64u * (uint32_t)pl.geom.B;
    uint64_t rows_nominal = 0, rows_swept = 0, cells_swept = 0, n_pruned = 0, n_second = 0, n_full = 0;
    std::vector<float> ratios;
    const DpResult *hres = c->h_res.as<DpResult>();
    for (uint32_t q = 0; q < bq; q++) {
        const uint64_t strips = (qd_host[q].L - 1) / kstrip + 1;
        rows_nominal += strips * qd_host[q].N;
        const DpResult &r = hres[q];
        if (r.attempts == 0) {  // (a kernel that sweeps everything)
            rows_swept += strips * qd_host[q].N;
            cells_swept += (uint64_t)qd_host[q].N * qd_host[q].L;
            continue;
        }
        rows_swept += r.rows_done;
        cells_swept += r.cells_done;
        n_pruned++;
        n_second += r.attempts == 2 ? 1 : 0;
        n_full += r.attempts >= 3 ? 1 : 0;
        if (r.status == 0 && r.gain0 > 0.f && r.raw < 0.f) ratios.push_back(-r.raw / r.gain0);
    }
    // The launch's smallest optimum / bound, less a margin: a query whose first bound fails pays a second sweep, and
    // the launch ends with its slowest wave -- one such query among the last to start costs the whole device a sweep's
    // time, so the guess aims at NO failures among queries like the ones seen (a wider band costs a few per cent).
    // Two guesses per store.  Alone (a launch without a scout pass: caller-built DAGs) a guess that fails ONE query costs
    // the whole launch a sweep's time -- it ends with its slowest wave -- so it aims below the smallest ratio seen
    // (round 5; the 2 % point, tried in round 6: 95 second attempts in 184 320 queries, DP launches 30.1 instead of
    // 26.4 ms).  As the GUARD of the scout's values it only has to catch a scout that lost its query: the 2 % point,
    // six per cent looser still in the kernel -- one poorly aligning query among 9216 does not widen everybody's guard.
    float rho_seen = -1.f, rho_guard_seen = -1.f;
    if (!ratios.empty()) {
        rho_seen = *std::min_element(ratios.begin(), ratios.end()) - 0.015f;
        const size_t at = ratios.size() / 50;
        std::nth_element(ratios.begin(), ratios.begin() + (std::ptrdiff_t)at, ratios.end());
        rho_guard_seen = ratios[at] - 0.015f;
    }
    std::lock_guard<std::mutex> slk(c->st->stats_mu);
    c->st->stats.dp_ms += ms;
    c->st->stats.dp_busy_ms += ms - shared;
    SH_CHECK(hipEventElapsedTime(&ms, c->ev[1], c->ev[2]));
    c->st->stats.backtrack_ms += ms;
    c->st->stats.dp_cells += cells;
    c->st->stats.dp_launches++;
    c->st->stats.dp_rows += rows_nominal;
    c->st->stats.dp_rows_swept += rows_swept;
    c->st->stats.dp_cells_swept += cells_swept;
    c->st->stats.dp_queries_pruned += n_pruned;
    c->st->stats.dp_second_attempts += n_second;
    c->st->stats.dp_full_sweeps += n_full;
    if (rho_seen > 0.f) {  // down at once (a second sweep per query is what a bold guess costs), up by halves
        float &rho = c->st->prune_rho;
        rho = rho_seen < rho ? rho_seen : 0.5f * (rho + rho_seen);
        rho = std::min(0.99f, std::max(0.05f, rho));
        float &rg = c->st->prune_rho_guard;
        rg = rho_guard_seen < rg ? rho_guard_seen : 0.5f * (rg + rho_guard_seen);
        rg = std::min(0.99f, std::max(0.05f, rg));
    }
    c->st->stats.dp_prune_rho = c->st->prune_rho;
    return 0;
}

// The two streams of a context: uploads, k-mer searches' copies ... on `stream`; DP hand-over, result copies (and, with
// SINA_HIP_CHAIN=0 / SINA_HIP_BT_ON_FIFO=0, the backtrack walk) on `stream_dp`.  Both at the DEFAULT priority since round 4.
// Rounds 1-3 created `stream` at the highest and `stream_dp` at the lowest priority (round 1: kernels of different
// batches overlapped freely and a DAG build starved beside a DP kernel).  Since the device-filling kernels go through
// the store's FIFO that no longer decides anything -- except that a lowest-priority walk is not dispatched while a
// default-priority kernel still has workgroups to hand out: with one more stream in the process the walk started
// 5 ms late in half of the launches, ran beside the next DP launch and took 16 ms instead of 7
// (profiles/r04_bt_delay.txt).  SINA_HIP_STREAM_PRIO=1 restores the priorities.
int make_streams(sina_hip_ctx *c) {
    static const bool prio = experiment_env("SINA_HIP_STREAM_PRIO") && experiment_env("SINA_HIP_STREAM_PRIO")[0] == '1';
    if (!prio) {
        SH_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        SH_CHECK(hipStreamCreateWithFlags(&c->stream_dp, hipStreamNonBlocking));
        return 0;
    }
    int least = 0, greatest = 0;
    SH_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    SH_CHECK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest));
    SH_CHECK(hipStreamCreateWithPriority(&c->stream_dp, hipStreamNonBlocking, least));
    return 0;
}

int upload_weights(sina_hip_ctx *c, const sina_hip_align_params *p) {
    if (p->weights != nullptr && p->n_weights > 0) {
        if (c->weights.reserve(sizeof(float) * p->n_weights)) return 1;
        SH_CHECK(hipMemcpyAsync(c->weights.p, p->weights, sizeof(float) * p->n_weights, hipMemcpyHostToDevice,
                                c->stream));
    }
    return 0;
}

static bool weighted_scheme(const sina_hip_align_params *p) { return p->weights != nullptr && p->n_weights > 0; }

static int align_graphs_impl(sina_hip_ctx *c, const sina_hip_graph_batch *g, const uint8_t *qmask,
                             const uint64_t *qoff, const sina_hip_align_params *p, sina_hip_align_out *out,
                             uint32_t *out_pos, float *dbg_value_host, uint32_t *dbg_vm, uint32_t *dbg_vs) {
    if (!c || !g || !qmask || !qoff || !p || !out) SH_FAIL("align_graphs: null argument");
    const uint32_t nq = g->nq;
    if (nq == 0) return 0;
    SH_CHECK(hipSetDevice(c->device));
    if (c->h_out_pos.reserve(4 * std::max<uint64_t>(qoff[nq] - qoff[0], 1))) return 1;
    const bool forbid = p->insertion == SINA_INSERTION_FORBID;
    if (forbid && !g->succ_minpos) SH_FAIL("align_graphs: insertion=forbid needs succ_minpos");
    uint32_t maxL = 0;
    for (uint32_t q = 0; q < nq; q++) {
        const uint64_t L = qoff[q + 1] - qoff[q];
        const uint64_t N = g->node_off[q + 1] - g->node_off[q];
        if (L == 0 || N == 0) SH_FAIL("align_graphs: empty query or graph");
        if (L > SINA_HIP_MAX_QUERY_LEN || N > 65535) SH_FAIL("align_graphs: query longer than SINA_HIP_MAX_QUERY_LEN bases or DAG of more than 65535 nodes");
        maxL = std::max<uint32_t>(maxL, (uint32_t)L);
    }
    DpPlan pl;
    if (plan_dp(c, maxL, &pl)) return 1;
    const int Lp = pl.geom.Lp();
    if (upload_weights(c, p)) return 1;
    float wmax = 0.f, wmin = 0.f;
    if (!g->node_score16 && g->node_weight) {
        const uint64_t n_all = g->node_off[nq] - g->node_off[0];
        for (uint64_t i = 0; i < n_all; i++) {
            const float w = g->node_weight[g->node_off[0] + i];
            if (!(w == w)) wmax = wmin = NAN;  // (a NaN compares false both ways and would slip through: it switches the row skip off)
            if (!(wmax == wmax)) break;
            wmax = (i == 0 || w > wmax) ? w : wmax;
            wmin = (i == 0 || w < wmin) ? w : wmin;
        }
    }
    PrunePlan pp = prune_plan(p, wmax, wmin, maxL, g->node_score16 != nullptr);
    if (c->no_prune) pp.on = 0;  // (sina_hip_debug_mesh, prune = 0)

    const uint64_t tb_budget_cells = tb_plane_budget(c) / tb_cell_bytes(forbid);
    HostPrep hp;
    uint32_t q0 = 0;
    while (q0 < nq) {
        // largest sub-batch whose trace-back plane fits the budget
        uint32_t q1 = q0;
        uint64_t cells = 0;
        while (q1 < nq) {
            const uint64_t N = g->node_off[q1 + 1] - g->node_off[q1];
            if (q1 > q0 && cells + N * Lp > tb_budget_cells) break;
            cells += N * Lp;
            q1++;
        }
        q1 = dp_round_range(q0, q1, nq, dp_wave_slots(c, pl.geom.B));
        if (dbg_vm) q1 = q0 + 1;
        c->dbg_planes = dbg_vm != nullptr;
        if (prep_range(g, qoff, q0, q1, Lp, pl.W, &hp, pp.on ? pp.kappa64 : 0.f)) return 1;
        const uint32_t bq = q1 - q0;
        const uint64_t nbase = g->node_off[q0], nn = g->node_off[q1] - nbase;
        const uint64_t ebase = g->edge_off[q0], ne = g->edge_off[q1] - ebase;
        const uint64_t qbase = qoff[q0], nqm = qoff[q1] - qbase;
        if (c->qd.reserve(sizeof(QDesc) * bq) || c->rec.reserve(sizeof(uint4) * nn) || c->node_pos.reserve(4 * nn) ||
            c->pred.reserve(4 * std::max<uint64_t>(ne, 1)) || c->succ_minpos.reserve(4 * nn) ||
            c->qmask.reserve(nqm) || (pp.on && c->rgain.reserve(8 * nn)))
            return 1;
        hipStream_t s = c->stream;
        SH_CHECK(hipMemcpyAsync(c->qd.p, hp.qd.data(), sizeof(QDesc) * bq, hipMemcpyHostToDevice, s));
        SH_CHECK(hipMemcpyAsync(c->rec.p, hp.rec.data(), sizeof(uint4) * nn, hipMemcpyHostToDevice, s));
        SH_CHECK(hipMemcpyAsync(c->node_pos.p, g->node_pos + nbase, 4 * nn, hipMemcpyHostToDevice, s));
        if (ne) SH_CHECK(hipMemcpyAsync(c->pred.p, hp.pred.data(), 4 * ne, hipMemcpyHostToDevice, s));
        if (g->succ_minpos)
            SH_CHECK(hipMemcpyAsync(c->succ_minpos.p, g->succ_minpos + nbase, 4 * nn, hipMemcpyHostToDevice, s));
        SH_CHECK(hipMemcpyAsync(c->qmask.p, qmask + qbase, nqm, hipMemcpyHostToDevice, s));
        PrunePlan pp_launch = pp;
        if (!hp.rgain_ok) pp_launch.on = 0;
        if (pp.on) SH_CHECK(hipMemcpyAsync(c->rgain.p, hp.rgain.data(), 8 * nn, hipMemcpyHostToDevice, s));
        c->profile_batch = g->node_score16 != nullptr;
        if (c->profile_batch) {  // --fs-no-graph: the profile's match-term tables (sina_hip.h)
            if (!g->self_score16) SH_FAIL("align_graphs: node_score16 without self_score16");
            if (weighted_scheme(p)) SH_FAIL("align_graphs: a profile batch takes no positional weights (scoring_scheme_profile)");
            if (c->prof16.reserve(64 * std::max<uint64_t>(nn, 1)) || c->self16.reserve(64)) return 1;
            SH_CHECK(hipMemcpyAsync(c->prof16.p, g->node_score16 + 16 * nbase, 64 * nn, hipMemcpyHostToDevice, s));
            SH_CHECK(hipMemcpyAsync(c->self16.p, g->self_score16, 64, hipMemcpyHostToDevice, s));
        }
        c->out_pos_base = qbase - qoff[0];
        if (run_dp_device(c, pl, hp.qd.data(), bq, nn, hp.tb_cells, hp.spill_rows, hp.cells, nqm, p, g->width, out + q0,
                          out_pos ? out_pos + qbase : nullptr, dbg_value_host != nullptr, pp_launch))
            return 1;
        if (dbg_vm) {  // single-query debug: unpack the planes
            const QDesc &d = hp.qd[0];
            std::vector<uint32_t> tbh((size_t)d.N * Lp);
            if (forbid) {
                SH_CHECK(hipMemcpy(tbh.data(), c->last_tb, 4 * tbh.size(), hipMemcpyDeviceToHost));
            } else {  // 16-bit cells (common.h)
                std::vector<uint16_t> t16(tbh.size());
                SH_CHECK(hipMemcpy(t16.data(), c->last_tb, 2 * t16.size(), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < tbh.size(); i++) tbh[i] = t16[i];
            }
            std::vector<float> vh;
            if (dbg_value_host) {
                vh.resize(tbh.size());
                SH_CHECK(hipMemcpy(vh.data(), c->dbg.p, 4 * vh.size(), hipMemcpyDeviceToHost));
            }
            // value_midx of a gap-extending deletion is the predecessor's gapm_idx (common.h, Ext / OpLast)
            const uint4 *rec = hp.rec.data() + d.node_off;
            const uint32_t *pr = hp.pred.data() + d.edge_off;
            const uint32_t ext_bit = forbid ? kTbExt : kTb16Ext;
            auto gapm_idx = [&](uint32_t x, uint32_t col) -> uint32_t {
                for (;;) {
                    const uint32_t np = rec[x].z & 0xffu;
                    if (np == 0) return 0;
                    const uint32_t lastp = pr[rec[x].x + np - 1] & 0xffffu;
                    const uint32_t cx = tbh[(size_t)x * Lp + col];
                    if (forbid ? (cx & kTbOpLast) != 0 : !(cx & kTb16XLast)) return lastp;
                    x = lastp;
                }
            };
            for (uint32_t m = 0; m < d.N; m++)
                for (uint32_t x = 0; x < d.L; x++) {
                    const uint32_t cell = tbh[(size_t)m * Lp + x];
                    uint32_t vm, vs;
                    if (forbid) {
                        vm = cell >> 16;
                        vs = cell & kTbSMask;
                    } else {  // type code + predecessor ordinal instead of the indices
                        const uint32_t t = cell & kTbTypeMask;
                        vm = t == kTbIns ? m : (t == kTbNone ? 0u : (pr[rec[m].x + (cell >> kTb16OrdShift)] & 0xffffu));
                        if (t == kTbNone) vs = 0;
                        else if (t == kTbMatch) vs = x - 1;
                        else if (t == kTbDel) vs = x;
                        else {  // insertion: where the run of insertion cells to the left ends
                            uint32_t k = x - 1;
                            while (k > 0 && (tbh[(size_t)m * Lp + k] & kTbTypeMask) == kTbIns) --k;
                            vs = k;
                        }
                    }
                    if (cell & ext_bit) vm = gapm_idx(vm, x);
                    dbg_vm[(size_t)m * d.L + x] = vm;
                    dbg_vs[(size_t)m * d.L + x] = vs;
                    if (dbg_value_host) dbg_value_host[(size_t)m * d.L + x] = vh[(size_t)m * Lp + x];
                }
        }
        q0 = q1;
    }
    c->dbg_planes = false;  // (only this call's launches were the debug entry's: a later launch must not clear a plane)
    return 0;
}

}  // namespace sina_hip

using namespace sina_hip;

extern "C" {

int sina_hip_abi_version(void) { return SINA_HIP_ABI_VERSION; }
const char *sina_hip_last_error(void) { return g_last_error.c_str(); }

// a partly built context is taken apart again when init / fork fails half-way
static void discard_ctx(sina_hip_ctx *c) {
    if (!c) return;
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream_dp) (void)hipStreamDestroy(c->stream_dp);
    c->free_all();
    delete c;
}
static int finish_ctx(sina_hip_ctx *c) {  // streams + events of a new context
    for (auto &e : c->ev) e = nullptr;
    if (make_streams(c)) return 1;
    for (auto &e : c->ev) SH_CHECK(hipEventCreate(&e));
    return 0;
}

// Runs when the library is loaded, i.e. normally before the HIP runtime has started: the pipeline's
// streams need more hardware queues than the runtime's default of four (see sina_amd/__init__.py).
// ROC_SIGNAL_POOL_SIZE: the runtime recycles completion signals out of a pool (default 64); four batches in
// flight, each with a dozen copies, kernels and events on several streams, run it dry, and from then on the
// runtime's helper thread creates and waits for interrupt signals one ioctl at a time -- 0.8 of a core in
// kernel mode at 125 k sequences/s (bench.py under SINA_HOST_PROFILE=1: "timed region: thread ..." lines;
// tools/ubench/bench_env_matrix.sh).  With 256 the thread is idle for 16S (3.2 -> 2.5 busy cores, same rate);
// the V4 shape (365 k sequences/s) needs 1024 for that (7.9 -> 6.8).
// An explicit setting in the environment wins, and a host that wants its process environment left alone
// altogether sets SINA_HIP_NO_RUNTIME_DEFAULTS (to anything but "0") before it loads the library: nothing is
// touched then -- the pipeline still works, with the costs described above.
__attribute__((constructor)) static void sina_hip_runtime_defaults() {
    const char *off = getenv("SINA_HIP_NO_RUNTIME_DEFAULTS");
    if (off && *off && !(off[0] == '0' && off[1] == 0)) return;
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    setenv("ROC_SIGNAL_POOL_SIZE", "1024", 0);
}

int sina_hip_init(int device, sina_hip_ctx **ctx) {
    if (!ctx) SH_FAIL("init: null ctx pointer");
    int ndev = 0;
    SH_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) SH_FAIL("init: no such HIP device");
    SH_CHECK(hipSetDevice(device));
    sina_hip_ctx *c = new sina_hip_ctx();
    c->device = device;
    c->st = new sina_hip_store();
    c->owns_store = true;
    memset(&c->st->stats, 0, sizeof(c->st->stats));
    // (the FIFO of device-filling kernels: two streams taking turns + the "queue has run dry" words, ctx.h)
    auto make_heavy = [](sina_hip_store *st) {
        if (hipStreamCreateWithFlags(&st->heavy, hipStreamNonBlocking) != hipSuccess) return 1;
        if (hipStreamCreateWithFlags(&st->heavy2, hipStreamNonBlocking) != hipSuccess) return 1;
        for (auto &e : st->heavy_done)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 1;
        for (auto &e : st->dp_end)
            if (hipEventCreate(&e) != hipSuccess) return 1;
        // (chained launches need hipStreamWaitValue32: a device without it keeps dry_mem == nullptr and every launch
        // waits for the end of the one before, as until round 3)
        int can_wait_value = 0;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&can_wait_value, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess) {
            (void)hipGetLastError();
            can_wait_value = 0;
        }
        if (!can_wait_value) return 0;
        // (the flag word in signal memory -- what hipStreamWaitValue32 is documented for --, the counters in a plain block;
        // a runtime that will not give signal memory leaves the store unchained)
        if (hipExtMallocWithFlags(reinterpret_cast<void **>(&st->dry_flag), 8, hipMallocSignalMemory) != hipSuccess) {
            (void)hipGetLastError();
            st->dry_flag = nullptr;
            return 0;
        }
        const size_t bytes = 4 * (size_t)sina_hip_store::kDryCounters;
        if (hipMalloc(reinterpret_cast<void **>(&st->dry_mem), bytes) != hipSuccess) return 1;
        if (hipMemset(st->dry_flag, 0, 8) != hipSuccess) return 1;
        return hipMemset(st->dry_mem, 0, bytes) != hipSuccess ? 1 : 0;
    };
    if (finish_ctx(c) || make_heavy(c->st)) {
        const std::string why = sina_hip_last_error();
        discard_ctx(c);
        set_error(why.empty() ? "init: could not create the context's streams" : why);
        return 1;
    }
    c->bind_hints();
    c->lds_budget = (size_t)std::max(0, atoi(test_knob("lds_kb").c_str())) * 1024;  // (test hook: LDS per DP workgroup)
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cu < 1) {
        (void)hipGetLastError();
        c->n_cu = 256;
    }
    *ctx = c;
    return 0;
}

int sina_hip_fork(sina_hip_ctx *parent, sina_hip_ctx **ctx) {
    if (!parent || !ctx) SH_FAIL("fork: null argument");
    SH_CHECK(hipSetDevice(parent->device));
    sina_hip_ctx *c = new sina_hip_ctx();
    c->device = parent->device;
    c->st = parent->st;  // same reference store, index and counters; never freed by the fork
    c->owns_store = false;
    c->lds_budget = parent->lds_budget;
    c->n_cu = parent->n_cu;
    c->bind_hints();
    if (finish_ctx(c)) {
        const std::string why = sina_hip_last_error();
        discard_ctx(c);
        set_error(why);
        return 1;
    }
    *ctx = c;
    return 0;
}

int sina_hip_prewarm(sina_hip_ctx *c, int kind) {
    if (!c || kind < 0 || kind > 2) SH_FAIL("prewarm: null ctx or unknown kind");
    std::lock_guard<std::mutex> lk(c->mu);
    SH_CHECK(hipSetDevice(c->device));
    return c->prewarm(kind);
}

void sina_hip_destroy(sina_hip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->stream_dp) (void)hipStreamSynchronize(c->stream_dp);
    c->free_all();
    for (auto &e : c->ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    if (c->stream_dp) (void)hipStreamDestroy(c->stream_dp);
    delete c;
}

int sina_hip_sync(sina_hip_ctx *c) {
    if (!c) SH_FAIL("sync: null ctx");
    SH_CHECK(hipSetDevice(c->device));
    SH_CHECK(hipStreamSynchronize(c->stream));
    SH_CHECK(hipStreamSynchronize(c->stream_dp));
    if (c->st->heavy) SH_CHECK(hipStreamSynchronize(c->st->heavy));
    if (c->st->heavy2) SH_CHECK(hipStreamSynchronize(c->st->heavy2));
    return 0;
}

const uint32_t *sina_hip_staged_out_pos(sina_hip_ctx *c) {
    return c ? static_cast<const uint32_t *>(c->h_out_pos.p) : nullptr;
}

void sina_hip_align_params_default(sina_hip_align_params *p) {
    memset(p, 0, sizeof(*p));
    p->match_score = 2;
    p->mismatch_score = -1;
    p->gap_penalty = 5;
    p->gap_ext_penalty = 2;
    p->fs_weight = 1;
    p->overhang = SINA_OVERHANG_ATTACH;
    p->lowercase = SINA_LOWERCASE_NONE;
    p->insertion = SINA_INSERTION_SHIFT;
}

int sina_hip_upload_refs(sina_hip_ctx *c, const uint32_t *ab, const uint64_t *off, uint32_t n_refs,
                         uint32_t width) {
    if (!c || !ab || !off) SH_FAIL("upload_refs: null argument");
    if (!c->owns_store) SH_FAIL("upload_refs: a forked context cannot change the reference store");
    std::lock_guard<std::mutex> lk(c->mu);
    SH_CHECK(hipSetDevice(c->device));
    const uint64_t total = off[n_refs];
    if (c->st->ref_ab.reserve(4 * std::max<uint64_t>(total, 1)) || c->st->ref_off.reserve(8 * ((uint64_t)n_refs + 1)))
        return 1;
    SH_CHECK(hipMemcpyAsync(c->st->ref_ab.p, ab, 4 * total, hipMemcpyHostToDevice, c->stream));
    SH_CHECK(hipMemcpyAsync(c->st->ref_off.p, off, 8 * ((uint64_t)n_refs + 1), hipMemcpyHostToDevice, c->stream));
    SH_CHECK(hipStreamSynchronize(c->stream));
    {
        std::lock_guard<std::mutex> alk(c->st->aux_mu);
        c->st->ref_off_host.assign(off, off + n_refs + 1);
        c->st->ref_off_host_ready.store(true, std::memory_order_release);
    }
    c->st->n_refs = n_refs;
    c->st->width = width;
    c->st->total_bases = total;
    c->st->have_refs = true;
    return 0;
}

int sina_hip_align_graphs(sina_hip_ctx *c, const sina_hip_graph_batch *g, const uint8_t *qmask,
                          const uint64_t *qoff, const sina_hip_align_params *p, sina_hip_align_out *out,
                          uint32_t *out_pos) {
    if (!c) SH_FAIL("align_graphs: null ctx");
    std::lock_guard<std::mutex> lk(c->mu);
    sina_hip_hint_guard hints(c);
    return align_graphs_impl(c, g, qmask, qoff, p, out, out_pos, nullptr, nullptr, nullptr);
}

int sina_hip_debug_mesh(sina_hip_ctx *c, const sina_hip_graph_batch *g, const uint8_t *qmask, uint32_t qlen,
                        const sina_hip_align_params *p, uint32_t *tb_vm, uint32_t *tb_vs, float *value, int prune) {
    if (!c || !g || g->nq != 1 || !tb_vm || !tb_vs) SH_FAIL("debug_mesh: needs exactly one query");
    std::lock_guard<std::mutex> lk(c->mu);
    struct no_prune_scope {  // (this call's launches only; the context is locked)
        sina_hip_ctx *c;
        ~no_prune_scope() { c->no_prune = false; }
    } scope{c};
    c->no_prune = prune == 0;
    const uint64_t qoff[2] = {0, qlen};
    sina_hip_align_out o;
    std::vector<uint32_t> pos(qlen);
    return align_graphs_impl(c, g, qmask, qoff, p, &o, pos.data(), value, tb_vm, tb_vs);
}

int sina_hip_debug_dp_info(sina_hip_ctx *c, uint32_t q, sina_hip_dp_info *out) {
    if (!c || !out) SH_FAIL("debug_dp_info: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (q >= c->last_bq || !c->res.p) SH_FAIL("debug_dp_info: no such query in the last launch");
    SH_CHECK(hipSetDevice(c->device));
    DpResult r;
    SH_CHECK(hipMemcpy(&r, c->res.as<DpResult>() + q, sizeof r, hipMemcpyDeviceToHost));
    out->end_m = r.end_m;
    out->end_s = r.end_s;
    out->raw = r.raw;
    out->status = r.status;
    out->rows_swept = r.rows_done;
    out->cells_swept = r.cells_done;
    out->attempts = r.attempts;
    out->gain0 = r.gain0;
    out->ubound = r.ubound;
    out->prune_step = c->last_prune_step;
    QDesc d;
    SH_CHECK(hipMemcpy(&d, c->qd.as<QDesc>() + q, sizeof d, hipMemcpyDeviceToHost));
    out->prune_gmin = d.gmin;
    out->scout = NAN;
    if (c->last_scout && c->scout_u.p) SH_CHECK(hipMemcpy(&out->scout, c->scout_u.as<float>() + q, sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int sina_hip_debug_rgain(sina_hip_ctx *c, uint32_t n, uint32_t *out, uint32_t *cols_right) {
    if (!c || !out) SH_FAIL("debug_rgain: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->rgain.p || c->rgain.cap < 8 * (size_t)n) SH_FAIL("debug_rgain: no bound of that many nodes on the device");
    SH_CHECK(hipSetDevice(c->device));
    std::vector<uint2> tmp(n);
    SH_CHECK(hipMemcpy(tmp.data(), c->rgain.p, 8 * (size_t)n, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) {
        out[i] = tmp[i].x;
        if (cols_right) cols_right[i] = tmp[i].y >> 16;
    }
    return 0;
}

int sina_hip_get_stats(sina_hip_ctx *c, sina_hip_stats *s) {
    if (!c || !s) SH_FAIL("get_stats: null argument");
    std::lock_guard<std::mutex> slk(c->st->stats_mu);
    *s = c->st->stats;
    s->n_dense_lists = c->st->dense_ready.load() ? c->st->n_dense : 0;
    return 0;
}

}  // extern "C"
