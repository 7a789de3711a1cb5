// The context object behind the opaque sina_hip_ctx handle.
#pragma once

#include <cmath>

#include "common.h"

struct sina_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    hipEvent_t ev[8];

    // reference store + k-mer index (HBM-resident for the life of the context)
    sina_hip::DevBuf ref_ab, ref_off, idx_off, idx_ids;
    std::vector<uint64_t> ref_off_host;  // host copy of the offsets (sizing of DAG-build scratch)
    uint32_t n_refs = 0, width = 0, k = 0, nofast = 0;
    uint64_t n_postings = 0, total_bases = 0;
    bool have_refs = false, have_index = false;

    // per-batch scratch, grown on demand and reused
    sina_hip::DevBuf qd, rec, node_pos, pred, succ_minpos, qmask, tb, spill, res, weights, out, out_pos, dbg;
    sina_hip::DevBuf k_qoff, k_scores, k_out_ids, k_out_scores, k_out_n, k_tmp0, k_tmp1, k_tmp2;
    sina_hip::DevBuf g_fam_ids, g_fam_off, g_tmp0, g_tmp1, g_tmp2, g_tmp3, g_sizes, g_wtab;
    float wtab_fs_weight = NAN;  // fs_weight the device weight table was computed for

    size_t lds_budget = 40 * 1024;
    uint64_t tb_budget_bytes = (uint64_t)24 << 30;
    sina_hip_stats stats;

    void free_all() {
        sina_hip::DevBuf *all[] = {&ref_ab, &ref_off, &idx_off, &idx_ids, &qd, &rec, &node_pos, &pred, &succ_minpos,
                                   &qmask, &tb, &spill, &res, &weights, &out, &out_pos, &dbg, &k_qoff,
                                   &k_scores, &k_out_ids, &k_out_scores, &k_out_n, &k_tmp0, &k_tmp1, &k_tmp2,
                                   &g_fam_ids, &g_fam_off, &g_tmp0, &g_tmp1, &g_tmp2, &g_tmp3, &g_sizes, &g_wtab};
        for (auto *b : all) b->release();
    }
};
