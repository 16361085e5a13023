// C-ABI entry points for pairwise matching (include/esfm.h, rows a-1..a-3 of SURVEY.md section 8).
// Host logic only: argument checks, the pair table, scratch management, kernel sequencing.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

#include "match_kernels.hpp"

using esfm::PairDesc;

namespace {


struct PairPlan {
    std::vector<PairDesc> tab;
    std::vector<int32_t> blk_pair;   // the pair of every 512-query block of the one-product front pass (its workgroups look their work up here)
    std::vector<int32_t> by_train;   // pair indices sorted by train set (l2_finish_kernel walks the pairs in this order: the workgroups that
                                     // fetch rows of one train set run next to each other, on one XCD, and find them in its L2)
    int n_blocks = 0;
    int n_blocks2 = 0;      // workgroups of the one-product front pass
    int max_nt = 0;
    int64_t total_queries = 0;
    int64_t total_rows = 0;
};

int make_plan(const int32_t *set_row_offset, int n_sets, const int32_t *pairs, int n_pairs, int query_block,
              int64_t *out_offset, PairPlan *plan)
{
    ESFM_REQUIRE(set_row_offset != nullptr && n_sets >= 1, "set_row_offset/n_sets");
    ESFM_REQUIRE(n_pairs >= 0 && (n_pairs == 0 || pairs != nullptr), "pairs/n_pairs");
    ESFM_REQUIRE(set_row_offset[0] == 0, "set_row_offset[0] must be 0");
    for (int s = 0; s < n_sets; ++s) ESFM_REQUIRE(set_row_offset[s + 1] >= set_row_offset[s], "set_row_offset must be non-decreasing");
    plan->total_rows = set_row_offset[n_sets];
    plan->tab.resize((size_t)n_pairs);
    int64_t off = 0;
    int64_t blk = 0, blk2 = 0;
    const int qb2 = esfm::l2_x1_query_block();
    for (int p = 0; p < n_pairs; ++p) {
        const int qs = pairs[2 * p], ts = pairs[2 * p + 1];
        ESFM_REQUIRE(qs >= 0 && qs < n_sets && ts >= 0 && ts < n_sets, "pair refers to a set out of range");
        PairDesc &d = plan->tab[(size_t)p];
        d.q_row0 = set_row_offset[qs]; d.nq = set_row_offset[qs + 1] - set_row_offset[qs];
        d.t_row0 = set_row_offset[ts]; d.nt = set_row_offset[ts + 1] - set_row_offset[ts];
        ESFM_REQUIRE(d.nt < (1 << 21), "train sets are limited to 2^21-1 rows");   // index field of the packed top-2 keys
        ESFM_REQUIRE(d.nq < (1 << 23), "query sets are limited to 2^23-1 rows");   // 32-bit byte offsets into a set (row fetches through buffer descriptors)
        d.out_off = off; d.blk_off = (int32_t)blk; d.blk_off2 = (int32_t)blk2;
        plan->max_nt = std::max(plan->max_nt, (int)d.nt);
        if (out_offset) out_offset[p] = off;
        off += d.nq;
        blk += (d.nq + query_block - 1) / query_block;
        blk2 += (d.nq + qb2 - 1) / qb2;
        ESFM_REQUIRE(blk < (int64_t)1 << 31, "too many workgroups for one launch; split the pair list");
        plan->blk_pair.resize((size_t)blk2, p);
    }
    if (out_offset) out_offset[n_pairs] = off;
    plan->by_train.resize((size_t)n_pairs);
    for (int p = 0; p < n_pairs; ++p) plan->by_train[(size_t)p] = p;
    std::stable_sort(plan->by_train.begin(), plan->by_train.end(),
                     [&](int32_t a, int32_t b) { return plan->tab[(size_t)a].t_row0 < plan->tab[(size_t)b].t_row0; });
    plan->n_blocks = (int)blk;
    plan->n_blocks2 = (int)blk2;
    plan->total_queries = off;
    return ESFM_OK;
}

// Upload the pair table through a pinned staging buffer.  The context remembers the last table:
// an identical pair list (the common case in a loop over the same frames) is not re-sent.
int upload_pairs(esfm_ctx *ctx, const PairPlan &plan, const PairDesc **dev_tab)
{
    // one blob: the pair table, then the pairs' indices sorted by train set (pair_order_of below), then the front pass's block table
    const size_t tab_bytes = plan.tab.size() * sizeof(PairDesc), ord_bytes = plan.by_train.size() * sizeof(int32_t);
    const size_t bytes = tab_bytes + ord_bytes + plan.blk_pair.size() * sizeof(int32_t);
    if (bytes == 0) { *dev_tab = nullptr; return ESFM_OK; }
    std::vector<char> blob(bytes);
    memcpy(blob.data(), plan.tab.data(), tab_bytes);
    memcpy(blob.data() + tab_bytes, plan.by_train.data(), ord_bytes);
    if (!plan.blk_pair.empty()) memcpy(blob.data() + tab_bytes + ord_bytes, plan.blk_pair.data(), bytes - tab_bytes - ord_bytes);
    if (ctx->pair_tab.cap >= bytes && ctx->pinned_cap >= bytes && ctx->last_pair_bytes == bytes &&
        memcmp(ctx->pinned, blob.data(), bytes) == 0) {
        *dev_tab = ctx->pair_tab.as<PairDesc>();
        return ESFM_OK;
    }
    // the pinned buffer may still be the source of an in-flight copy: drain before rewriting it
    ctx->last_pair_bytes = 0;   // the cache is valid only once the new table's copy has been enqueued
    ESFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (int rc = ctx->pin(bytes)) return rc;
    if (int rc = ctx->pair_tab.reserve(bytes)) return rc;
    memcpy(ctx->pinned, blob.data(), bytes);
    ESFM_HIP_TRY(esfm::copy_h2d(ctx->pair_tab.ptr, ctx->pinned, bytes, ctx->stream));
    ctx->last_pair_bytes = bytes;
    *dev_tab = ctx->pair_tab.as<PairDesc>();
    return ESFM_OK;
}
inline const int32_t *pair_order_of(const PairDesc *dev_tab, int n_pairs) { return reinterpret_cast<const int32_t *>(dev_tab + n_pairs); }
inline const int32_t *blk_pair_of(const PairDesc *dev_tab, int n_pairs) { return pair_order_of(dev_tab, n_pairs) + n_pairs; }

// Where the ratio test's survivors go (the match entry points); NULL: the raw 2-NN table is the result.
struct MatchOut { int32_t *query_idx, *train_idx; float *distance; int32_t *n_out; };

// (re)allocates a buffer of counters that must read zero: a fresh allocation is cleared once, after that the kernels leave it clean
int reserve_zeroed(esfm::DevBuf &b, size_t bytes, hipStream_t st, bool *grew = nullptr)
{
    if (bytes <= b.cap) return ESFM_OK;
    if (int rc = b.reserve(bytes)) return rc;
    ESFM_HIP_TRY(hipMemsetAsync(b.ptr, 0, b.cap, st));
    if (grew) *grew = true;
    return ESFM_OK;
}

bool is_prepared(const esfm_ctx *ctx, esfm_metric metric, const void *desc_dev, int64_t total_rows, int width)
{
    return ctx->prep_desc != nullptr && ctx->prep_desc == desc_dev && ctx->prep_metric == (int)metric && ctx->prep_rows == total_rows &&
           ctx->prep_width == width;
}

size_t desc_bytes(esfm_metric metric, int64_t total_rows, int width)
{
    return (size_t)total_rows * (metric == ESFM_L2_F32 ? sizeof(float) * (size_t)width : (size_t)width);
}

// esfm_ctx_set_prepared_check(ctx, 1): a call that is about to rely on prepared operands first re-derives the buffer's fingerprint
// and compares it with the one taken at prepare time (one read of the buffer + one host round trip per call: a debugging aid, off
// by default).  A buffer that was rewritten in place -- or freed and replaced by another allocation at the same address -- fails
// the call with ESFM_ERR_STALE_PREPARED and ends the prepared state, instead of matching against the old rows' images.
int verify_prepared(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev, int64_t total_rows, int width)
{
    if (!ctx->prep_check) return ESFM_OK;
    if (!ctx->prep_has_sum) { ctx->prep_desc = nullptr; return ESFM_OK; }      // prepared before the check was switched on: re-derive
    unsigned long long *sums = ctx->prep_sum.as<unsigned long long>();
    if (int rc = esfm::launch_buffer_checksum(ctx->stream, desc_dev, desc_bytes(metric, total_rows, width), sums + 1)) return rc;
    unsigned long long h[2] = {0ull, 0ull};
    ESFM_HIP_TRY(esfm::copy_d2h(h, sums, sizeof(h), ctx->stream));
    ESFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (h[0] == h[1]) return ESFM_OK;
    ctx->prep_desc = nullptr;
    esfm::set_error("the descriptor buffer %p was modified (or replaced by another allocation at the same address) after esfm_match_prepare_dev: "
                    "call esfm_match_prepare_dev again, or esfm_match_release_prepared before rewriting / freeing a prepared buffer", desc_dev);
    return ESFM_ERR_STALE_PREPARED;
}

// 2-NN table for every query of every pair, written to knn_idx/knn_dist (device, 2 per query).
// `ratio`: the caller will only keep the queries with d0 < ratio d1 (the match entry points), so the one-product pass may drop the
// ones that provably fail (train index -2, see l2_knn_bf16x1_kernel); INFINITY: every query's exact 2-NN (the knn2 entry points).
// `mo` != NULL: the path may run the ratio test + compaction itself (64-float L2: inside l2_finish_kernel) and says so in *ratio_done.
int knn2_core(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev, int width, const PairPlan &plan,
              const PairDesc *dev_tab, int32_t *knn_idx, float *knn_dist, double ratio, const MatchOut *mo, bool *ratio_done)
{
    if (ratio_done) *ratio_done = false;
    const int n_pairs = (int)plan.tab.size();
    if (n_pairs == 0 || plan.total_queries == 0) return ESFM_OK;
    hipStream_t st = ctx->stream;
    if (is_prepared(ctx, metric, desc_dev, plan.total_rows, width))
        if (int rc = verify_prepared(ctx, metric, desc_dev, plan.total_rows, width)) return rc;
    const bool prepared = is_prepared(ctx, metric, desc_dev, plan.total_rows, width);
    if (metric == ESFM_L2_F32) {
        const float *desc = reinterpret_cast<const float *>(desc_dev);
        // counters: [0,16) and [16,32) the two phases of the one-product path, [32,48) the other L2 passes, [48,64) scratch
        if (int rc = reserve_zeroed(ctx->counters, 64 * sizeof(int32_t), st)) return rc;
        if ((ctx->l2_audit == 3 || ctx->l2_audit == 4) && !(esfm::l2_bf16_pass(width) && esfm::l2_one_product_pass() && esfm::l2_x1_supported(plan.max_nt))) {
            esfm::set_error("audit modes 3 and 4 need the one-product pass (64-float descriptors, train sets <= 65536 rows, ESFM_L2_PASS unset)");
            return ESFM_ERR_UNSUPPORTED;
        }
        ctx->last_n_queries = plan.total_queries;
        int32_t *cnt_other = ctx->counters.as<int32_t>() + 32;       // the passes that zero their counters themselves
        ctx->counters_cur = cnt_other;
        if (esfm::l2_mfma_supported(width) && ctx->l2_audit != 2) {
            const int64_t cap64 = std::min<int64_t>(plan.total_queries, (int64_t)1 << 30);
            if (int rc = ctx->flagged.reserve(sizeof(int32_t) * 2 * (size_t)cap64)) return rc;
            const bool front = esfm::l2_bf16_pass(width) && esfm::l2_one_product_pass() && esfm::l2_x1_supported(plan.max_nt);
            if (front) {
                // 64-float rows.  Launch 1 (only when the descriptor buffer has not been prepared): bf16 images, norms, residual norms.
                // Launch 2: one bf16 product per f32 product, fused top-K fold, ratio screen (l2_knn_bf16x1_kernel).  Launch 3: exact
                // re-rank of the screen's survivors + certificate, the uncertified ones through the threshold filter, overflowed chunks by
                // brute force, ratio test + compaction (l2_finish_kernel).  Audit modes: 1 no brute force, 3 / 4 launch 3 stops after
                // the re-rank.
                if (!prepared) {
                    ctx->prep_desc = nullptr;          // the images below replace whatever was prepared
                    if (int rc = ctx->norms.reserve(sizeof(float) * (size_t)std::max<int64_t>(plan.total_rows, 1))) return rc;
                    if (int rc = ctx->l2_hi.reserve(esfm::l2_hi_bytes(plan.total_rows))) return rc;
                }
                if (int rc = ctx->pair_list2.reserve(sizeof(int32_t) * (size_t)plan.total_queries)) return rc;
                if (int rc = ctx->knn_d2.reserve(sizeof(float) * (size_t)plan.total_queries)) return rc;
                if (int rc = ctx->surv_list.reserve(esfm::l2_survivor_entry_bytes() * (size_t)plan.total_queries)) return rc;
                for (int k = 0; k < 2; ++k) {        // the two phases of the per-pair counters (survivors, uncertified)
                    bool g1 = false, g2 = false;
                    if (int rc = reserve_zeroed(k ? ctx->pair_cnt2b : ctx->pair_cnt2, sizeof(int32_t) * (size_t)n_pairs, st, &g1)) return rc;
                    if (int rc = reserve_zeroed(k ? ctx->surv_cntb : ctx->surv_cnt, sizeof(int32_t) * (size_t)n_pairs, st, &g2)) return rc;
                    if (g1 || g2) {                  // a fresh allocation is clean as a whole; its sibling is cleared with it
                        ESFM_HIP_TRY(hipMemsetAsync((k ? ctx->pair_cnt2b : ctx->pair_cnt2).ptr, 0, (k ? ctx->pair_cnt2b : ctx->pair_cnt2).cap, st));
                        ESFM_HIP_TRY(hipMemsetAsync((k ? ctx->surv_cntb : ctx->surv_cnt).ptr, 0, (k ? ctx->surv_cntb : ctx->surv_cnt).cap, st));
                        ctx->l2_phase_pairs[k] = 0;
                    }
                }
                if (int rc = reserve_zeroed(ctx->fin_done, sizeof(int32_t) * (size_t)n_pairs, st)) return rc;
                const int ph = ctx->l2_phase;
                int32_t *cur_unc = (ph ? ctx->pair_cnt2b : ctx->pair_cnt2).as<int32_t>(), *oth_unc = (ph ? ctx->pair_cnt2 : ctx->pair_cnt2b).as<int32_t>();
                int32_t *cur_surv = (ph ? ctx->surv_cntb : ctx->surv_cnt).as<int32_t>(), *oth_surv = (ph ? ctx->surv_cnt : ctx->surv_cntb).as<int32_t>();
                int32_t *cur_counters = ctx->counters.as<int32_t>() + 16 * ph, *oth_counters = ctx->counters.as<int32_t>() + 16 * (1 - ph);
                ctx->counters_cur = cur_counters;
                if (!prepared) {
                    if (int rc = esfm::launch_l2_split_bf16(st, desc, plan.total_rows, nullptr, ctx->norms.as<float>(), ctx->counters.as<int32_t>() + 48,
                                                            nullptr, 0, ctx->l2_hi.ptr, nullptr))
                        return rc;
                }
                {
                    esfm::KernelTimer tm(ctx, ESFM_K_L2_KNN);
                    if (int rc = esfm::launch_l2_knn_bf16x1(st, ctx->num_cu, desc, ctx->l2_hi.ptr, plan.total_rows, ctx->norms.as<float>(), dev_tab,
                                                            blk_pair_of(dev_tab, n_pairs), plan.n_blocks2, knn_idx, knn_dist, cur_counters, (int)cap64, cur_surv, ctx->surv_list.ptr, ratio,
                                                            /* markers: only where something reads the table itself */ mo == nullptr || ctx->l2_audit == 3 || ctx->l2_audit == 4,
                                                            ctx->l2_audit == 4 ? ctx->flagged.as<int32_t>() : nullptr, oth_unc, oth_surv,
                                                            ctx->l2_phase_pairs[1 - ph], oth_counters))
                        return rc;
                }
                ctx->l2_phase_pairs[1 - ph] = 0;
                ctx->l2_phase_pairs[ph] = n_pairs;
                ctx->l2_phase = 1 - ph;
                esfm::KernelTimer tm(ctx, ESFM_K_L2_SECOND);
                if (int rc = esfm::launch_l2_finish(st, desc, ctx->l2_hi.ptr, plan.total_rows, ctx->norms.as<float>(), dev_tab, pair_order_of(dev_tab, n_pairs), n_pairs, cur_surv,
                                                    ctx->surv_list.ptr, cur_unc, ctx->pair_list2.as<int32_t>(), ctx->knn_d2.as<float>(), knn_idx, knn_dist,
                                                    cur_counters, ctx->flagged.as<int32_t>(), (int)cap64, ctx->fin_done.as<int32_t>(), ctx->l2_audit,
                                                    mo != nullptr && ctx->l2_audit != 3 && ctx->l2_audit != 4, ratio,
                                                    mo ? mo->query_idx : nullptr, mo ? mo->train_idx : nullptr, mo ? mo->distance : nullptr, mo ? mo->n_out : nullptr))
                    return rc;
                if (ctx->l2_audit == 3 || ctx->l2_audit == 4) return ESFM_OK;   // audit: the first pass's own answers and failures / rejections
                if (mo && ratio_done) *ratio_done = true;
                return ESFM_OK;
            }
            ctx->prep_desc = nullptr;          // the passes below write their own norms / images
            if (int rc = ctx->norms.reserve(sizeof(float) * (size_t)std::max<int64_t>(plan.total_rows, 1))) return rc;
            if (esfm::l2_bf16_pass(width)) {
                // 64-float rows without the one-product pass (ESFM_L2_PASS=bf16x3, train sets of more than 65536 rows): the three-product
                // kernel of round 2 + the exact re-scan of its uncertified queries, pair by pair.  Audit mode 1 stops before the re-scan.
                if (int rc = ctx->hm_exp.reserve(esfm::l2_split_bytes(width, plan.total_rows))) return rc;
                if (int rc = ctx->pair_cnt.reserve(sizeof(int32_t) * (size_t)n_pairs)) return rc;
                if (int rc = ctx->pair_list.reserve(sizeof(int32_t) * (size_t)plan.total_queries)) return rc;
                if (int rc = esfm::launch_l2_split_bf16(st, desc, plan.total_rows, ctx->hm_exp.ptr, ctx->norms.as<float>(), cnt_other,
                                                        ctx->pair_cnt.as<int32_t>(), n_pairs, nullptr, nullptr))
                    return rc;
                {
                    esfm::KernelTimer tm(ctx, ESFM_K_L2_KNN);
                    if (int rc = esfm::launch_l2_knn_bf16(st, desc, ctx->hm_exp.ptr, plan.total_rows, ctx->norms.as<float>(), dev_tab, n_pairs,
                                                          plan.n_blocks, knn_idx, knn_dist, ctx->flagged.as<int32_t>(),
                                                          cnt_other, (int)cap64, ctx->pair_cnt.as<int32_t>(), ctx->pair_list.as<int32_t>()))
                        return rc;
                }
                if (ctx->l2_audit == 1) return ESFM_OK;   // audit: leave the pass's own answer in place
                // certificate failures, binned per pair by the pass: exact re-scan, the pair's queries sharing every train row
                esfm::KernelTimer tm(ctx, ESFM_K_L2_RESCAN);
                return esfm::launch_l2_rescan64_pairs(st, desc, dev_tab, n_pairs, ctx->pair_cnt.as<int32_t>(), ctx->pair_list.as<int32_t>(),
                                                      knn_idx, knn_dist);
            }
            ESFM_HIP_TRY(hipMemsetAsync(cnt_other, 0, 64, st));
            if (int rc = esfm::launch_l2_norms(st, desc, width, plan.total_rows, ctx->norms.as<float>())) return rc;
            {
                esfm::KernelTimer tm(ctx, ESFM_K_L2_KNN);
                if (int rc = esfm::launch_l2_knn_mfma(st, width, desc, ctx->norms.as<float>(), dev_tab, n_pairs, plan.n_blocks, knn_idx,
                                                      knn_dist, ctx->flagged.as<int32_t>(), cnt_other, (int)cap64))
                    return rc;
            }
            if (ctx->l2_audit == 1) return ESFM_OK;   // audit: leave the pass's own answer in place
            // certificate failures: exact scan, grid-stride over the device-side count (no host sync)
            const int grid = (int)std::min<int64_t>(plan.total_queries, 8 * (int64_t)ctx->num_cu);
            esfm::KernelTimer tm(ctx, ESFM_K_L2_RESCAN);
            if (int rc = esfm::launch_l2_exact_scan(st, width, desc, dev_tab, n_pairs, ctx->flagged.as<int32_t>(),
                                                    cnt_other, plan.total_queries, grid, knn_idx, knn_dist))
                return rc;
        } else {
            // no MFMA build for this width (or audit mode 2): exact scan of every query (correct, not fast)
            ESFM_HIP_TRY(hipMemsetAsync(cnt_other, 0, 64, st));
            const int grid = (int)std::min<int64_t>(plan.total_queries, 64 * (int64_t)ctx->num_cu);
            if (int rc = esfm::launch_l2_exact_scan(st, width, desc, dev_tab, n_pairs, nullptr, cnt_other,
                                                    plan.total_queries, grid, knn_idx, knn_dist))
                return rc;
        }
        return ESFM_OK;
    }
    if (metric == ESFM_HAMMING) {
        if (!esfm::hamming_supported(width)) {
            esfm::set_error("hamming descriptors must be 16, 32 or 64 bytes (got %d)", width);
            return ESFM_ERR_UNSUPPORTED;
        }
        // 256-bit descriptors: the FP4-MFMA form while the train sets fit its position code, else the byte-per-bit i8 form; what
        // esfm_match_prepare_dev left in hm_exp counts only if it is the form this call runs
        const bool fp4 = esfm::hamming_fp4_supported(width, plan.max_nt);
        const bool have = prepared && ctx->prep_hm_fp4 == fp4;
        if (!have) {
            ctx->prep_desc = nullptr;
            if (int rc = ctx->hm_exp.reserve(esfm::hamming_expanded_bytes(width, plan.total_rows))) return rc;
        }
        esfm::KernelTimer tm(ctx, ESFM_K_HAMMING_KNN);
        if (fp4) {
            // the match entry points: ratio test + compaction inside the same launch (the last block of a pair does it)
            const bool fused = mo != nullptr && ratio_done != nullptr;
            if (fused) { if (int rc = reserve_zeroed(ctx->fin_done, sizeof(int32_t) * (size_t)n_pairs, st)) return rc; }
            if (int rc = esfm::launch_hamming_fp4(st, desc_dev, plan.total_rows, ctx->hm_exp.ptr, dev_tab, blk_pair_of(dev_tab, n_pairs), plan.n_blocks2, knn_idx,
                                                  knn_dist, ratio, /*expanded=*/have, fused ? ctx->fin_done.as<int32_t>() : nullptr, n_pairs,
                                                  fused ? mo->query_idx : nullptr, fused ? mo->train_idx : nullptr, fused ? mo->distance : nullptr,
                                                  fused ? mo->n_out : nullptr))
                return rc;
            if (fused) *ratio_done = true;
            return ESFM_OK;
        }
        return esfm::launch_hamming_knn(st, width, desc_dev, plan.total_rows, ctx->hm_exp.ptr, dev_tab, n_pairs, plan.n_blocks, knn_idx,
                                        knn_dist, /*expanded=*/have);
    }
    esfm::set_error("unknown metric %d", (int)metric);
    return ESFM_ERR_INVALID_ARG;
}

int check_common(esfm_ctx *ctx, esfm_metric metric, int width)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    if (metric != ESFM_L2_F32 && metric != ESFM_HAMMING) { esfm::set_error("unknown metric %d", (int)metric); return ESFM_ERR_INVALID_ARG; }
    if (width <= 0) { esfm::set_error("descriptor width must be positive"); return ESFM_ERR_INVALID_ARG; }
    return esfm::set_device(ctx);
}

// Host-pointer single pair: stage [train rows | query rows] into one device buffer, run the
// batched path with the pair (1, 0), copy back.
int single_pair(esfm_ctx *ctx, esfm_metric metric, const void *q, int nq, const void *t, int nt, int width, bool filtered,
                double ratio, int32_t *o_a, int32_t *o_b, float *o_c, int32_t *n_out)
{
    if (int rc = check_common(ctx, metric, width)) return rc;
    ESFM_REQUIRE(nq >= 0 && nt >= 0, "negative row count");
    ESFM_REQUIRE(nq == 0 || q != nullptr, "q is NULL");
    ESFM_REQUIRE(nt == 0 || t != nullptr, "t is NULL");
    if (n_out) *n_out = 0;
    if (nq == 0) return ESFM_OK;
    const size_t row_bytes = metric == ESFM_L2_F32 ? sizeof(float) * (size_t)width : (size_t)width;
    const size_t tb = row_bytes * (size_t)nt, qb = row_bytes * (size_t)nq;
    if (int rc = ctx->stage_a.reserve(tb + qb + 16)) return rc;
    hipStream_t st = ctx->stream;
    char *d = ctx->stage_a.as<char>();
    if (tb) ESFM_HIP_TRY(esfm::copy_h2d(d, t, tb, st));
    ESFM_HIP_TRY(esfm::copy_h2d(d + tb, q, qb, st));
    const int32_t offs[3] = {0, nt, nt + nq};
    const int32_t pr[2] = {1, 0};
    int64_t out_off[2];
    PairPlan plan;
    if (int rc = make_plan(offs, 2, pr, 1, metric == ESFM_L2_F32 ? esfm::l2_query_block(width) : esfm::hamming_query_block(width), out_off, &plan)) return rc;
    const PairDesc *dev_tab = nullptr;
    if (int rc = upload_pairs(ctx, plan, &dev_tab)) return rc;
    if (int rc = ctx->knn_idx.reserve(sizeof(int32_t) * 2 * (size_t)nq)) return rc;
    if (int rc = ctx->knn_dist.reserve(sizeof(float) * 2 * (size_t)nq)) return rc;
    if (filtered) {
        if (int rc = ctx->stage_b.reserve(sizeof(int32_t) * (size_t)nq)) return rc;
        if (int rc = ctx->stage_c.reserve(sizeof(int32_t) * (size_t)nq)) return rc;
        if (int rc = ctx->stage_d.reserve(sizeof(float) * (size_t)nq)) return rc;
        if (int rc = ctx->stage_e.reserve(sizeof(int32_t))) return rc;
    }
    const MatchOut mo = {ctx->stage_b.as<int32_t>(), ctx->stage_c.as<int32_t>(), ctx->stage_d.as<float>(), ctx->stage_e.as<int32_t>()};
    bool ratio_done = false;
    if (int rc = knn2_core(ctx, metric, d, width, plan, dev_tab, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<float>(), filtered ? ratio : (double)INFINITY,
                           filtered ? &mo : nullptr, &ratio_done))
        return rc;
    if (!filtered) {
        ESFM_HIP_TRY(esfm::copy_d2h(o_a, ctx->knn_idx.ptr, sizeof(int32_t) * 2 * (size_t)nq, st));
        ESFM_HIP_TRY(esfm::copy_d2h(o_c, ctx->knn_dist.ptr, sizeof(float) * 2 * (size_t)nq, st));
        ESFM_HIP_TRY(hipStreamSynchronize(st));
        return ESFM_OK;
    }
    if (!ratio_done)
        if (int rc = esfm::launch_ratio_compact(st, dev_tab, 1, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<float>(), ratio,
                                                mo.query_idx, mo.train_idx, mo.distance, mo.n_out))
            return rc;
    int32_t n = 0;
    ESFM_HIP_TRY(esfm::copy_d2h(&n, ctx->stage_e.ptr, sizeof(int32_t), st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    if (n > 0) {
        ESFM_HIP_TRY(esfm::copy_d2h(o_a, ctx->stage_b.ptr, sizeof(int32_t) * (size_t)n, st));
        ESFM_HIP_TRY(esfm::copy_d2h(o_b, ctx->stage_c.ptr, sizeof(int32_t) * (size_t)n, st));
        ESFM_HIP_TRY(esfm::copy_d2h(o_c, ctx->stage_d.ptr, sizeof(float) * (size_t)n, st));
        ESFM_HIP_TRY(hipStreamSynchronize(st));
    }
    if (n_out) *n_out = n;
    return ESFM_OK;
}

}  // namespace

extern "C" {

int esfm_knn2_l2_f32(esfm_ctx *ctx, const float *q, int nq, const float *t, int nt, int dim, int32_t *idx, float *dist)
{
    if (nq > 0 && (!idx || !dist)) { esfm::set_error("idx/dist is NULL"); return ESFM_ERR_INVALID_ARG; }
    return single_pair(ctx, ESFM_L2_F32, q, nq, t, nt, dim, false, 0.0, idx, nullptr, dist, nullptr);
}

int esfm_knn2_hamming(esfm_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes, int32_t *idx, float *dist)
{
    if (nq > 0 && (!idx || !dist)) { esfm::set_error("idx/dist is NULL"); return ESFM_ERR_INVALID_ARG; }
    return single_pair(ctx, ESFM_HAMMING, q, nq, t, nt, nbytes, false, 0.0, idx, nullptr, dist, nullptr);
}

int esfm_match_l2_f32(esfm_ctx *ctx, const float *q, int nq, const float *t, int nt, int dim, double ratio,
                      int32_t *query_idx, int32_t *train_idx, float *distance, int32_t *n_out)
{
    if (!n_out || (nq > 0 && (!query_idx || !train_idx || !distance))) { esfm::set_error("output pointer is NULL"); return ESFM_ERR_INVALID_ARG; }
    return single_pair(ctx, ESFM_L2_F32, q, nq, t, nt, dim, true, ratio, query_idx, train_idx, distance, n_out);
}

int esfm_match_hamming(esfm_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes, double ratio,
                       int32_t *query_idx, int32_t *train_idx, float *distance, int32_t *n_out)
{
    if (!n_out || (nq > 0 && (!query_idx || !train_idx || !distance))) { esfm::set_error("output pointer is NULL"); return ESFM_ERR_INVALID_ARG; }
    return single_pair(ctx, ESFM_HAMMING, q, nq, t, nt, nbytes, true, ratio, query_idx, train_idx, distance, n_out);
}

static int knn2_pairs_dev(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev, const int32_t *set_row_offset, int n_sets,
                         int width, const int32_t *pairs, int n_pairs, double ratio, int32_t *knn_idx_dev, float *knn_dist_dev,
                         int64_t *out_offset)
{
    if (int rc = check_common(ctx, metric, width)) return rc;
    ESFM_REQUIRE(out_offset != nullptr, "out_offset is NULL");
    PairPlan plan;
    if (int rc = make_plan(set_row_offset, n_sets, pairs, n_pairs, metric == ESFM_L2_F32 ? esfm::l2_query_block(width) : esfm::hamming_query_block(width),
                           out_offset, &plan))
        return rc;
    if (plan.total_queries == 0) return ESFM_OK;
    ESFM_REQUIRE(desc_dev && knn_idx_dev && knn_dist_dev, "device pointer is NULL");
    const PairDesc *dev_tab = nullptr;
    if (int rc = upload_pairs(ctx, plan, &dev_tab)) return rc;
    return knn2_core(ctx, metric, desc_dev, width, plan, dev_tab, knn_idx_dev, knn_dist_dev, ratio, nullptr, nullptr);
}

int esfm_knn2_pairs_dev(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev, const int32_t *set_row_offset, int n_sets,
                        int width, const int32_t *pairs, int n_pairs, int32_t *knn_idx_dev, float *knn_dist_dev,
                        int64_t *out_offset)
{
    return knn2_pairs_dev(ctx, metric, desc_dev, set_row_offset, n_sets, width, pairs, n_pairs, (double)INFINITY, knn_idx_dev, knn_dist_dev, out_offset);
}

// Audit of the Hamming matcher's ratio screen (tests): the raw table of a call that screens with `ratio` -- the queries the pass
// dropped as "cannot pass d0 < ratio d1" carry train index -2 in both slots, every other query its exact 2-NN.
int esfm_knn2_pairs_screened_dev(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev, const int32_t *set_row_offset, int n_sets,
                                 int width, const int32_t *pairs, int n_pairs, double ratio, int32_t *knn_idx_dev, float *knn_dist_dev,
                                 int64_t *out_offset)
{
    if (metric != ESFM_HAMMING) { esfm::set_error("esfm_knn2_pairs_screened_dev: Hamming only (the L2 screen is audited through esfm_ctx_set_l2_audit mode 4)"); return ESFM_ERR_UNSUPPORTED; }
    if (!(ratio == ratio)) { esfm::set_error("ratio is NaN"); return ESFM_ERR_INVALID_ARG; }
    return knn2_pairs_dev(ctx, metric, desc_dev, set_row_offset, n_sets, width, pairs, n_pairs, ratio, knn_idx_dev, knn_dist_dev, out_offset);
}

int esfm_match_pairs_dev(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev, const int32_t *set_row_offset, int n_sets,
                         int width, const int32_t *pairs, int n_pairs, double ratio, int32_t *query_idx_dev,
                         int32_t *train_idx_dev, float *distance_dev, int32_t *n_out_dev, int64_t *out_offset)
{
    if (int rc = check_common(ctx, metric, width)) return rc;
    ESFM_REQUIRE(out_offset != nullptr, "out_offset is NULL");
    PairPlan plan;
    if (int rc = make_plan(set_row_offset, n_sets, pairs, n_pairs, metric == ESFM_L2_F32 ? esfm::l2_query_block(width) : esfm::hamming_query_block(width),
                           out_offset, &plan))
        return rc;
    if (n_pairs == 0) return ESFM_OK;
    ESFM_REQUIRE(n_out_dev != nullptr, "n_out_dev is NULL");
    ESFM_REQUIRE(plan.total_queries == 0 || (desc_dev && query_idx_dev && train_idx_dev && distance_dev), "device pointer is NULL");
    const PairDesc *dev_tab = nullptr;
    if (int rc = upload_pairs(ctx, plan, &dev_tab)) return rc;
    if (int rc = ctx->knn_idx.reserve(sizeof(int32_t) * 2 * (size_t)std::max<int64_t>(plan.total_queries, 1))) return rc;
    if (int rc = ctx->knn_dist.reserve(sizeof(float) * 2 * (size_t)std::max<int64_t>(plan.total_queries, 1))) return rc;
    const MatchOut mo = {query_idx_dev, train_idx_dev, distance_dev, n_out_dev};
    bool ratio_done = false;
    if (plan.total_queries == 0) ESFM_HIP_TRY(hipMemsetAsync(n_out_dev, 0, sizeof(int32_t) * (size_t)n_pairs, ctx->stream));
    if (int rc = knn2_core(ctx, metric, desc_dev, width, plan, dev_tab, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<float>(), ratio, &mo, &ratio_done)) return rc;
    if (ratio_done || plan.total_queries == 0) return ESFM_OK;
    return esfm::launch_ratio_compact(ctx->stream, dev_tab, n_pairs, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<float>(), ratio,
                                      query_idx_dev, train_idx_dev, distance_dev, n_out_dev);
}

// Host-pointer form of the batched pair loop (SURVEY 8b's esfm_match_pairs): upload once, prepare once, one launch sequence for
// the whole pair list, one read-back.  The uploaded rows stay in the context (ctx->bank) and stay prepared, so a second call on
// the same host buffer contents would still re-upload (the library cannot know the rows are unchanged) -- callers that match
// the same sets repeatedly keep them on the device and use esfm_match_pairs_dev.
int esfm_match_pairs(esfm_ctx *ctx, esfm_metric metric, const void *desc_host, const int32_t *set_row_offset, int n_sets, int width,
                     const int32_t *pairs, int n_pairs, double ratio, int32_t *query_idx, int32_t *train_idx, float *distance,
                     int32_t *n_out, int64_t *out_offset)
{
    if (int rc = check_common(ctx, metric, width)) return rc;
    ESFM_REQUIRE(out_offset != nullptr, "out_offset is NULL");
    PairPlan plan;
    if (int rc = make_plan(set_row_offset, n_sets, pairs, n_pairs, metric == ESFM_L2_F32 ? esfm::l2_query_block(width) : esfm::hamming_query_block(width),
                           out_offset, &plan))
        return rc;
    if (n_pairs == 0) return ESFM_OK;
    ESFM_REQUIRE(n_out != nullptr, "n_out is NULL");
    for (int p = 0; p < n_pairs; ++p) n_out[p] = 0;
    if (plan.total_queries == 0) return ESFM_OK;
    ESFM_REQUIRE(desc_host && query_idx && train_idx && distance, "host pointer is NULL");
    hipStream_t st = ctx->stream;
    const size_t row_bytes = metric == ESFM_L2_F32 ? sizeof(float) * (size_t)width : (size_t)width;
    const size_t bytes = row_bytes * (size_t)plan.total_rows;
    ctx->prep_desc = nullptr;                      // the bank below is rewritten: whatever was prepared from it is stale
    if (int rc = ctx->bank.reserve(bytes + 16)) return rc;
    ESFM_HIP_TRY(esfm::copy_h2d(ctx->bank.ptr, desc_host, bytes, st));
    if (int rc = esfm_match_prepare_dev(ctx, metric, ctx->bank.ptr, plan.total_rows, width)) return rc;
    const size_t nq = (size_t)plan.total_queries;
    if (int rc = ctx->stage_b.reserve(sizeof(int32_t) * nq)) return rc;
    if (int rc = ctx->stage_c.reserve(sizeof(int32_t) * nq)) return rc;
    if (int rc = ctx->stage_d.reserve(sizeof(float) * nq)) return rc;
    if (int rc = ctx->stage_e.reserve(sizeof(int32_t) * (size_t)n_pairs)) return rc;
    std::vector<int64_t> off2((size_t)n_pairs + 1);
    if (int rc = esfm_match_pairs_dev(ctx, metric, ctx->bank.ptr, set_row_offset, n_sets, width, pairs, n_pairs, ratio, ctx->stage_b.as<int32_t>(),
                                      ctx->stage_c.as<int32_t>(), ctx->stage_d.as<float>(), ctx->stage_e.as<int32_t>(), off2.data()))
        return rc;
    ESFM_HIP_TRY(esfm::copy_d2h(n_out, ctx->stage_e.ptr, sizeof(int32_t) * (size_t)n_pairs, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    // The lists sit at out_offset[p] in arrays of sum(nq) slots; a ratio test keeps a few per cent of the queries.  When the matches
    // are less than a quarter of the slots they are packed on the device, read back as three dense arrays and placed from a host
    // copy (config 4's shape: 267 M slots, 4.4 M matches -- 52 MB over PCIe instead of 3.2 GB); otherwise the arrays go back whole.
    size_t total = 0;
    for (int p = 0; p < n_pairs; ++p) total += (size_t)n_out[p];
    if (total == 0) return ESFM_OK;
    if (total * 4 >= nq) {
        ESFM_HIP_TRY(esfm::copy_d2h(query_idx, ctx->stage_b.ptr, sizeof(int32_t) * nq, st));
        ESFM_HIP_TRY(esfm::copy_d2h(train_idx, ctx->stage_c.ptr, sizeof(int32_t) * nq, st));
        ESFM_HIP_TRY(esfm::copy_d2h(distance, ctx->stage_d.ptr, sizeof(float) * nq, st));
        ESFM_HIP_TRY(hipStreamSynchronize(st));
        return ESFM_OK;
    }
    std::vector<long long> tab(2 * (size_t)n_pairs);
    {
        long long run = 0;
        for (int p = 0; p < n_pairs; ++p) { tab[2 * (size_t)p] = off2[(size_t)p]; tab[2 * (size_t)p + 1] = run; run += n_out[p]; }
    }
    const size_t tab_bytes = (sizeof(long long) * tab.size() + 255) & ~(size_t)255;
    if (int rc = ctx->stage_a.reserve(tab_bytes + 12 * total + 64)) return rc;
    char *base = ctx->stage_a.as<char>();
    int32_t *dq = reinterpret_cast<int32_t *>(base + tab_bytes), *dtn = dq + total;
    float *dd = reinterpret_cast<float *>(dtn + total);
    ESFM_HIP_TRY(esfm::copy_h2d(base, tab.data(), sizeof(long long) * tab.size(), st));
    if (int rc = esfm::launch_pack_match_lists(st, reinterpret_cast<const long long *>(base), ctx->stage_e.as<int32_t>(), n_pairs, ctx->stage_b.as<int32_t>(),
                                               ctx->stage_c.as<int32_t>(), ctx->stage_d.as<float>(), dq, dtn, dd))
        return rc;
    std::vector<int32_t> hq(2 * total);
    std::vector<float> hd(total);
    ESFM_HIP_TRY(esfm::copy_d2h(hq.data(), dq, sizeof(int32_t) * 2 * total, st));
    ESFM_HIP_TRY(esfm::copy_d2h(hd.data(), dd, sizeof(float) * total, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    for (int p = 0; p < n_pairs; ++p) {
        const size_t n = (size_t)n_out[p], so = (size_t)tab[2 * (size_t)p], dof = (size_t)tab[2 * (size_t)p + 1];
        if (!n) continue;
        memcpy(query_idx + so, hq.data() + dof, sizeof(int32_t) * n);
        memcpy(train_idx + so, hq.data() + total + dof, sizeof(int32_t) * n);
        memcpy(distance + so, hd.data() + dof, sizeof(float) * n);
    }
    return ESFM_OK;
}

int esfm_match_prepare_dev(esfm_ctx *ctx, esfm_metric metric, const void *desc_dev, int64_t total_rows, int width)
{
    if (int rc = check_common(ctx, metric, width)) return rc;
    ESFM_REQUIRE(total_rows >= 0 && total_rows < ((int64_t)1 << 31), "total_rows");
    ESFM_REQUIRE(total_rows == 0 || desc_dev != nullptr, "desc_dev is NULL");
    ctx->prep_desc = nullptr;
    if (total_rows == 0) return ESFM_OK;
    hipStream_t st = ctx->stream;
    if (metric == ESFM_L2_F32 && esfm::l2_bf16_pass(width) && esfm::l2_one_product_pass()) {
        if (int rc = reserve_zeroed(ctx->counters, 64 * sizeof(int32_t), st)) return rc;
        if (int rc = ctx->norms.reserve(sizeof(float) * (size_t)total_rows)) return rc;
        if (int rc = ctx->l2_hi.reserve(esfm::l2_hi_bytes(total_rows))) return rc;
        if (int rc = esfm::launch_l2_split_bf16(st, reinterpret_cast<const float *>(desc_dev), total_rows, nullptr, ctx->norms.as<float>(),
                                                ctx->counters.as<int32_t>() + 48, nullptr, 0, ctx->l2_hi.ptr, nullptr))
            return rc;
    } else if (metric == ESFM_HAMMING && esfm::hamming_supported(width) && esfm::hamming_expanded_bytes(width, total_rows) > 0) {
        if (int rc = ctx->hm_exp.reserve(esfm::hamming_expanded_bytes(width, total_rows))) return rc;
        // (the FP4 form's images unless it is switched off: a later call whose train sets do not fit its position code re-derives)
        ctx->prep_hm_fp4 = esfm::hamming_fp4_supported(width, 0);
        if (ctx->prep_hm_fp4) { if (int rc = esfm::launch_hamming_expand_fp4(st, desc_dev, total_rows, ctx->hm_exp.ptr)) return rc; }
        else if (int rc = esfm::launch_hamming_expand(st, width, desc_dev, total_rows, ctx->hm_exp.ptr)) return rc;
    } else {
        return ESFM_OK;       // nothing to derive for this metric / width: the match calls work on the rows themselves
    }
    ctx->prep_has_sum = false;
    if (ctx->prep_check) {
        if (int rc = ctx->prep_sum.reserve(2 * sizeof(unsigned long long))) return rc;
        if (int rc = esfm::launch_buffer_checksum(st, desc_dev, desc_bytes(metric, total_rows, width), ctx->prep_sum.as<unsigned long long>())) return rc;
        ctx->prep_has_sum = true;
    }
    ctx->prep_desc = desc_dev; ctx->prep_metric = (int)metric; ctx->prep_rows = total_rows; ctx->prep_width = width;
    return ESFM_OK;
}

int esfm_ctx_set_prepared_check(esfm_ctx *ctx, int enable)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ctx->prep_check = enable ? 1 : 0;
    return ESFM_OK;
}

int esfm_match_release_prepared(esfm_ctx *ctx)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ctx->prep_desc = nullptr;
    return ESFM_OK;
}

int esfm_match_release_prepared_buffer(esfm_ctx *ctx, const void *desc_dev)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    if (ctx->prep_desc == desc_dev) ctx->prep_desc = nullptr;
    return ESFM_OK;
}

int esfm_match_prepared_buffer(esfm_ctx *ctx, const void **desc_dev_out)
{
    if (!ctx || !desc_dev_out) { esfm::set_error("esfm_match_prepared_buffer: bad arguments"); return ESFM_ERR_INVALID_ARG; }
    *desc_dev_out = ctx->prep_desc;
    return ESFM_OK;
}

int esfm_match_last_stats(esfm_ctx *ctx, int64_t *n_queries, int64_t *n_rescanned)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    if (int rc = esfm::set_device(ctx)) return rc;
    int32_t c = 0;
    if (ctx->counters_cur) {
        ESFM_HIP_TRY(esfm::copy_d2h(&c, ctx->counters_cur, sizeof(int32_t), ctx->stream));
        ESFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (n_queries) *n_queries = ctx->last_n_queries;
    if (n_rescanned) *n_rescanned = c;
    return ESFM_OK;
}

int esfm_match_last_second_pass(esfm_ctx *ctx, int64_t *n_second_pass)
{
    if (!ctx || !n_second_pass) { esfm::set_error("esfm_match_last_second_pass: bad arguments"); return ESFM_ERR_INVALID_ARG; }
    if (int rc = esfm::set_device(ctx)) return rc;
    int32_t c[2] = {0, 0};
    if (ctx->counters_cur) {
        ESFM_HIP_TRY(esfm::copy_d2h(c, ctx->counters_cur, sizeof(c), ctx->stream));
        ESFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *n_second_pass = c[1];
    return ESFM_OK;
}

int esfm_match_debug_counters(esfm_ctx *ctx, int32_t *out16)
{
    if (!ctx || !out16) { esfm::set_error("esfm_match_debug_counters: bad arguments"); return ESFM_ERR_INVALID_ARG; }
    if (int rc = esfm::set_device(ctx)) return rc;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    if (ctx->counters_cur) {
        ESFM_HIP_TRY(esfm::copy_d2h(out16, ctx->counters_cur, 16 * sizeof(int32_t), ctx->stream));
        ESFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ESFM_OK;
}

int esfm_ctx_set_l2_audit(esfm_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 4) { esfm::set_error("esfm_ctx_set_l2_audit: bad arguments"); return ESFM_ERR_INVALID_ARG; }
    ctx->l2_audit = mode;
    return ESFM_OK;
}

int esfm_match_last_flagged(esfm_ctx *ctx, int32_t *out, int64_t cap, int64_t *n)
{
    if (!ctx || !n || cap < 0 || (cap > 0 && !out)) { esfm::set_error("esfm_match_last_flagged: bad arguments"); return ESFM_ERR_INVALID_ARG; }
    if (int rc = esfm::set_device(ctx)) return rc;
    int32_t c = 0;
    if (ctx->counters_cur) {
        ESFM_HIP_TRY(esfm::copy_d2h(&c, ctx->counters_cur, sizeof(int32_t), ctx->stream));
        ESFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *n = c;
    const int64_t k = std::min<int64_t>(std::min<int64_t>(c, cap), (int64_t)(ctx->flagged.cap / (2 * sizeof(int32_t))));
    if (k > 0) {
        ESFM_HIP_TRY(esfm::copy_d2h(out, ctx->flagged.ptr, sizeof(int32_t) * 2 * (size_t)k, ctx->stream));
        ESFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ESFM_OK;
}

// Pair list of the reference's loop (sfm.cpp:140-143), sharded.  Greedy longest-processing-time
// over pairs sorted by descending cost keeps shards within one pair's cost of each other; inside a
// shard the reference order (i ascending, j ascending) is kept so results concatenate trivially.
int esfm_shard_pair_list(int n_frames, const int32_t *rows_per_frame, int rank, int world, int32_t *pairs_out)
{
    if (n_frames < 0 || world < 1 || rank < 0 || rank >= world || !pairs_out) {
        esfm::set_error("esfm_shard_pair_list: bad arguments");
        return ESFM_ERR_INVALID_ARG;
    }
    struct Item { int i, j; double cost; };
    std::vector<Item> items;
    items.reserve((size_t)n_frames * (size_t)std::max(n_frames - 1, 0) / 2);
    for (int i = 0; i < n_frames; ++i)
        for (int j = 0; j < i; ++j)
            items.push_back({i, j, rows_per_frame ? (double)rows_per_frame[i] * (double)rows_per_frame[j] : 1.0});
    std::vector<int> order(items.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = (int)k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return items[(size_t)a].cost > items[(size_t)b].cost; });
    std::vector<double> load((size_t)world, 0.0);
    std::vector<char> mine(items.size(), 0);
    for (int k : order) {
        int best = 0;
        for (int w = 1; w < world; ++w) if (load[(size_t)w] < load[(size_t)best]) best = w;
        load[(size_t)best] += items[(size_t)k].cost;
        if (best == rank) mine[(size_t)k] = 1;
    }
    int n = 0;
    for (size_t k = 0; k < items.size(); ++k)
        if (mine[k]) { pairs_out[2 * n] = items[k].i; pairs_out[2 * n + 1] = items[k].j; ++n; }
    return n;
}

}  // extern "C"
