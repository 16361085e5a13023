// C-ABI entry points for two-view geometric verification (include/esfm.h): the replacement
// for cv::findEssentialMat(..., CV_RANSAC, prob, threshold, mask) and cv::recoverPose as
// MotionEstimator::estimate2D2D_E5P_RANSAC calls them (reference cpp_code/src/estimate_motion.cpp:49-67).
//
// Host side: the cv::RNG sample stream (a 64-bit multiply-with-carry recurrence, inherently sequential), OpenCV's
// best-model / adaptive-iteration-count bookkeeping replayed over the inlier counts the GPU produced for a chunk of
// iterations of every pair at once, and the 3 x 3 SVD that turns an essential matrix into its four candidate poses.
// Everything that touches the correspondences -- the 5-point kernel, Sampson scoring, masks, triangulation and the
// cheirality test -- runs in ransac_kernels.hip.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "five_point_core.hpp"
#include "ransac_host.hpp"
#include "ransac_kernels.hpp"

using esfm::RansacPair;
using esfm::ransac::CvRng;
using esfm::ransac::draw_subset;
using esfm::ransac::kMaxIters;
using esfm::ransac::kModelPoints;
using esfm::ransac::update_num_iters;

namespace {

constexpr int kChunk = 64;            // iterations of every active pair evaluated per round

int fill_pairs(int n_pairs, const int32_t *off, const float *K4, double threshold, std::vector<RansacPair> &tab)
{
    tab.resize((size_t)n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        RansacPair &r = tab[(size_t)p];
        memset(&r, 0, sizeof(r));
        r.first = off[p]; r.count = off[p + 1] - off[p]; r.active = 1;
        r.fx = (double)K4[4 * p]; r.cx = (double)K4[4 * p + 1]; r.fy = (double)K4[4 * p + 2]; r.cy = (double)K4[4 * p + 3];
        if (!(std::isfinite(r.fx) && std::isfinite(r.fy) && std::isfinite(r.cx) && std::isfinite(r.cy)) || r.fx == 0.0 || r.fy == 0.0) {
            esfm::set_error("bad camera intrinsics for pair %d", p);
            return ESFM_ERR_NUMERIC;
        }
        const double t = threshold / ((r.fx + r.fy) / 2.0);   // five-point.cpp: threshold /= (fx + fy) / 2
        r.thresh_sq = (float)(t * t);
        r.dist_thresh = 50.0;
    }
    return ESFM_OK;
}

// eigen-decomposition of a symmetric 3 x 3 by cyclic Jacobi (columns of V)
void jacobi3(double A[9], double V[9])
{
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep) {     // (the CPU restatement's jacobi_eig to the letter -- cap, stopping rule, sums: R and t come out bit-identical)
        const double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8], off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off <= 1e-40 * diag || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[3 * p + q];
                if (apq == 0.0) continue;
                const double th = (A[4 * q] - A[4 * p]) / (2.0 * apq);
                const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(1.0 + th * th));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
                for (int r = 0; r < 3; ++r) { const double x = A[3 * r + p], y = A[3 * r + q]; A[3 * r + p] = c * x - s * y; A[3 * r + q] = s * x + c * y; }
                for (int r = 0; r < 3; ++r) { const double x = A[3 * p + r], y = A[3 * q + r]; A[3 * p + r] = c * x - s * y; A[3 * q + r] = s * x + c * y; }
                for (int r = 0; r < 3; ++r) { const double x = V[3 * r + p], y = V[3 * r + q]; V[3 * r + p] = c * x - s * y; V[3 * r + q] = s * x + c * y; }
            }
    }
}

double det3(const double *M) { return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]); }

void mul3(const double *A, const double *B, double *C)
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

// cv::decomposeEssentialMat: E = U diag(s) V'; det U, det V' forced positive; R1 = U W V', R2 = U W' V', t = U(:, 2)
void decompose_essential(const double *E, double *R1, double *R2, double *t)
{
    double G[9], V[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) G[3 * a + b] = E[a] * E[b] + E[3 + a] * E[3 + b] + E[6 + a] * E[6 + b];
    jacobi3(G, V);
    int o[3] = {0, 1, 2};                           // singular values descending; equal ones keep their order (an insertion sort, as there)
    for (int i = 1; i < 3; ++i) { const int v = o[i]; int j = i - 1; while (j >= 0 && G[4 * o[j]] < G[4 * v]) { o[j + 1] = o[j]; --j; } o[j + 1] = v; }
    double v[3][3], u[3][3];
    for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) v[k][a] = V[3 * a + o[k]];
    for (int k = 0; k < 2; ++k) {
        double nn = 0.0;
        for (int a = 0; a < 3; ++a) { u[k][a] = E[3 * a] * v[k][0] + E[3 * a + 1] * v[k][1] + E[3 * a + 2] * v[k][2]; nn += u[k][a] * u[k][a]; }
        nn = std::sqrt(nn);
        for (int a = 0; a < 3; ++a) u[k][a] /= nn;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1]; u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2]; u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    double U[9], Vt[9];
    for (int a = 0; a < 3; ++a) for (int k = 0; k < 3; ++k) { U[3 * a + k] = u[k][a]; Vt[3 * k + a] = v[k][a]; }
    if (det3(U) < 0) for (double &x : U) x = -x;
    if (det3(Vt) < 0) for (double &x : Vt) x = -x;
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9];
    mul3(U, W, T); mul3(T, Vt, R1);
    mul3(U, Wt, T); mul3(T, Vt, R2);
    t[0] = U[2]; t[1] = U[5]; t[2] = U[8];
}

int check_pair_args(esfm_ctx *ctx, int n_pairs, const int32_t *off, const float *pts1, const float *pts2, const float *K4)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(n_pairs >= 0, "negative pair count");
    if (n_pairs == 0) return ESFM_OK;
    ESFM_REQUIRE(off && K4, "NULL argument");
    ESFM_REQUIRE(off[0] == 0, "point_offset[0] must be 0");
    for (int p = 0; p < n_pairs; ++p) ESFM_REQUIRE(off[p + 1] >= off[p], "point_offset must be non-decreasing");
    ESFM_REQUIRE(off[n_pairs] == 0 || (pts1 && pts2), "NULL point arrays");
    return esfm::set_device(ctx);
}

}  // namespace

extern "C" {

int esfm_find_essential_pairs(esfm_ctx *ctx, int n_pairs, const int32_t *point_offset, const float *pts1, const float *pts2,
                              const float *K4_per_pair, double prob, double threshold, double *E_out, uint8_t *mask, int32_t *status,
                              int32_t *iterations)
{
    if (int rc = check_pair_args(ctx, n_pairs, point_offset, pts1, pts2, K4_per_pair)) return rc;
    if (n_pairs == 0) return ESFM_OK;
    ESFM_REQUIRE(E_out && status, "NULL output");
    ESFM_REQUIRE(prob > 0.0 && prob < 1.0, "confidence must be in (0, 1)");   // CV_Assert( confidence > 0 && confidence < 1 )
    const int n_total = point_offset[n_pairs];
    ESFM_REQUIRE(n_total == 0 || mask, "mask is NULL");
    std::vector<RansacPair> tab;
    if (int rc = fill_pairs(n_pairs, point_offset, K4_per_pair, threshold, tab)) return rc;
    hipStream_t st = ctx->stream;

    struct State { CvRng rng; int niters = kMaxIters, max_good = 0, iter = 0; bool done = false, exact5 = false; };
    std::vector<State> S((size_t)n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        status[p] = 0;
        if (iterations) iterations[p] = 0;
        for (int k = 0; k < 9; ++k) E_out[9 * (size_t)p + k] = 0.0;
        const int n = tab[(size_t)p].count;
        if (n < kModelPoints) { S[(size_t)p].done = true; tab[(size_t)p].active = 0; }   // registrator returns false: no model, no mask
        if (n == kModelPoints) S[(size_t)p].exact5 = true;
    }
    if (n_total > 0) memset(mask, 0, (size_t)n_total);

    const size_t n_slots = (size_t)n_pairs * kChunk;
    if (int rc = ctx->stage_a.reserve(sizeof(float) * 2 * (size_t)std::max(n_total, 1))) return rc;
    if (int rc = ctx->stage_b.reserve(sizeof(float) * 2 * (size_t)std::max(n_total, 1))) return rc;
    if (int rc = ctx->stage_c.reserve(sizeof(RansacPair) * (size_t)n_pairs)) return rc;
    if (int rc = ctx->stage_d.reserve(sizeof(int32_t) * (16 * n_slots + 3 * (size_t)n_pairs))) return rc;   // samples | n_models | counts | take
    if (int rc = ctx->stage_e.reserve(sizeof(double) * 90 * n_slots + sizeof(double) * 9 * (size_t)n_pairs + sizeof(double) * 4 * (size_t)std::max(n_total, 1) + 32 +
                                      (size_t)std::max(n_total, 1))) return rc;
    float *d_p1 = ctx->stage_a.as<float>(), *d_p2 = ctx->stage_b.as<float>();
    RansacPair *d_tab = ctx->stage_c.as<RansacPair>();
    int32_t *d_samples = ctx->stage_d.as<int32_t>();
    int32_t *d_nmodels = d_samples + 5 * n_slots;
    int32_t *d_counts = d_nmodels + n_slots;
    int32_t *d_take = d_counts + 10 * n_slots;
    double *d_models = ctx->stage_e.as<double>();
    double *d_best = d_models + 90 * n_slots;
    double *d_npts = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(d_best + 9 * (size_t)n_pairs) + 31) & ~(uintptr_t)31);   // (double4 records)
    uint8_t *d_mask = reinterpret_cast<uint8_t *>(d_npts + 4 * (size_t)std::max(n_total, 1));
    if (n_total > 0) {
        ESFM_HIP_TRY(esfm::copy_h2d(d_p1, pts1, sizeof(float) * 2 * (size_t)n_total, st));
        ESFM_HIP_TRY(esfm::copy_h2d(d_p2, pts2, sizeof(float) * 2 * (size_t)n_total, st));
    }
    ESFM_HIP_TRY(hipMemsetAsync(d_best, 0, sizeof(double) * 9 * (size_t)n_pairs, st));
    // the correspondences in normalised coordinates, once for all rounds (the table's geometry does not change between rounds)
    ESFM_HIP_TRY(esfm::copy_h2d(d_tab, tab.data(), sizeof(RansacPair) * (size_t)n_pairs, st));
    if (int rc = esfm::launch_essential_normalise(st, d_tab, n_pairs, d_p1, d_p2, d_npts)) return rc;

    // the per-round tables in pinned memory (pageable copies of 0.4 MB up and 0.85 MB down per round were staged by the runtime)
    if (int rc = ctx->pin_rounds(sizeof(int32_t) * (16 * n_slots + 3 * (size_t)n_pairs))) return rc;
    int32_t *samples = static_cast<int32_t *>(ctx->pinned_rounds), *nmodels = samples + 5 * n_slots, *counts = nmodels + n_slots,
            *take = counts + 10 * n_slots;
    for (;;) {
        bool any = false;
        for (int p = 0; p < n_pairs; ++p) {
            State &s = S[(size_t)p];
            tab[(size_t)p].active = s.done ? 0 : 1;
            int32_t *dst = &samples[5 * (size_t)p * kChunk];
            if (s.done) { for (int k = 0; k < kChunk; ++k) dst[5 * k] = -1; continue; }
            any = true;
            const int n = tab[(size_t)p].count;
            for (int k = 0; k < kChunk; ++k) {
                if (s.exact5) { if (k == 0) for (int j = 0; j < 5; ++j) dst[j] = j; else dst[5 * k] = -1; }
                else draw_subset(s.rng, n, dst + 5 * k);
            }
        }
        if (!any) break;
        ESFM_HIP_TRY(esfm::copy_h2d(d_tab, tab.data(), sizeof(RansacPair) * (size_t)n_pairs, st));
        ESFM_HIP_TRY(esfm::copy_h2d(d_samples, samples, sizeof(int32_t) * 5 * n_slots, st));
        ESFM_HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * 10 * n_slots, st));
        if (int rc = esfm::launch_essential_chunk(st, d_tab, n_pairs, d_p1, d_p2, d_npts, d_samples, kChunk, d_models, d_nmodels, d_counts, ctx)) return rc;
        ESFM_HIP_TRY(esfm::copy_d2h(nmodels, d_nmodels, sizeof(int32_t) * 11 * n_slots, st));   // n_models | counts
        ESFM_HIP_TRY(hipStreamSynchronize(st));
        // replay RANSACPointSetRegistrator::run over this chunk
        int n_take = 0;
        for (int p = 0; p < n_pairs; ++p) {
            State &s = S[(size_t)p];
            if (s.done) continue;
            const int n = tab[(size_t)p].count;
            long best_slot = -1; int best_m = -1;
            if (s.exact5) {
                const size_t g = (size_t)p * kChunk;
                if (nmodels[g] > 0) { best_slot = (long)g; best_m = 0; s.max_good = n; }
                s.done = true; s.iter = 0;
            } else {
                for (int k = 0; k < kChunk && s.iter < s.niters; ++k, ++s.iter) {
                    const size_t g = (size_t)p * kChunk + (size_t)k;
                    for (int m = 0; m < nmodels[g]; ++m) {
                        const int good = counts[10 * g + (size_t)m];
                        if (good > std::max(s.max_good, kModelPoints - 1)) {
                            best_slot = (long)g; best_m = m; s.max_good = good;
                            s.niters = update_num_iters(prob, (double)(n - good) / n, kModelPoints, s.niters);
                        }
                    }
                }
                if (s.iter >= s.niters) s.done = true;
            }
            if (best_slot >= 0) { take[3 * n_take] = p; take[3 * n_take + 1] = (int32_t)best_slot; take[3 * n_take + 2] = best_m; ++n_take; }
        }
        // the pairs' new best models in ONE launch (a 72-byte device-to-device copy per pair was 300 API calls per round: most of the
        // batch's host time once the solver stopped being it)
        if (n_take > 0) {
            ESFM_HIP_TRY(esfm::copy_h2d(d_take, take, sizeof(int32_t) * 3 * (size_t)n_take, st));
            if (int rc = esfm::launch_essential_take_best(st, d_take, n_take, d_models, d_best)) return rc;
        }
    }
    for (int p = 0; p < n_pairs; ++p) {
        tab[(size_t)p].active = 1;
        status[p] = S[(size_t)p].max_good > 0 ? 1 : 0;
        if (iterations) iterations[p] = S[(size_t)p].iter;
    }
    ESFM_HIP_TRY(esfm::copy_h2d(d_tab, tab.data(), sizeof(RansacPair) * (size_t)n_pairs, st));
    if (n_total > 0) {
        if (int rc = esfm::launch_essential_mask(st, d_tab, n_pairs, d_p1, d_p2, d_best, d_mask)) return rc;
        ESFM_HIP_TRY(esfm::copy_d2h(mask, d_mask, (size_t)n_total, st));
    }
    ESFM_HIP_TRY(esfm::copy_d2h(E_out, d_best, sizeof(double) * 9 * (size_t)n_pairs, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    for (int p = 0; p < n_pairs; ++p) {
        if (status[p]) {
            if (S[(size_t)p].exact5) memset(mask + point_offset[p], 1, (size_t)tab[(size_t)p].count);   // bestMask.setTo(1)
        } else if (tab[(size_t)p].count > 0) {
            memset(mask + point_offset[p], 0, (size_t)tab[(size_t)p].count);
        }
    }
    return ESFM_OK;
}

int esfm_find_essential_mat(esfm_ctx *ctx, const float *pts1, const float *pts2, int n, const float *K4, double prob, double threshold,
                            double *E, uint8_t *mask, int32_t *iterations)
{
    if (n < 0) { esfm::set_error("negative point count"); return ESFM_ERR_INVALID_ARG; }
    const int32_t off[2] = {0, n};
    int32_t status = 0;
    if (int rc = esfm_find_essential_pairs(ctx, 1, off, pts1, pts2, K4, prob, threshold, E, mask, &status, iterations)) return rc;
    if (!status) { esfm::set_error("no essential matrix found (fewer than 5 points or no model with 5 inliers)"); return ESFM_ERR_NUMERIC; }
    return ESFM_OK;
}

int esfm_recover_pose_pairs(esfm_ctx *ctx, int n_pairs, const int32_t *point_offset, const float *pts1, const float *pts2,
                            const float *K4_per_pair, const double *E, uint8_t *mask, double *R, double *t, int32_t *good)
{
    if (int rc = check_pair_args(ctx, n_pairs, point_offset, pts1, pts2, K4_per_pair)) return rc;
    if (n_pairs == 0) return ESFM_OK;
    ESFM_REQUIRE(E && R && t, "NULL argument");
    const int n_total = point_offset[n_pairs];
    std::vector<RansacPair> tab;
    if (int rc = fill_pairs(n_pairs, point_offset, K4_per_pair, 1.0, tab)) return rc;
    // the four candidates in OpenCV's order: (R1, t), (R2, t), (R1, -t), (R2, -t)
    std::vector<double> poses(48 * (size_t)n_pairs), R1(9 * (size_t)n_pairs), R2(9 * (size_t)n_pairs), tt(3 * (size_t)n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        for (int k = 0; k < 9; ++k) if (!std::isfinite(E[9 * (size_t)p + k])) { esfm::set_error("non-finite essential matrix"); return ESFM_ERR_NUMERIC; }
        decompose_essential(E + 9 * (size_t)p, &R1[9 * (size_t)p], &R2[9 * (size_t)p], &tt[3 * (size_t)p]);
        for (int c = 0; c < 4; ++c) {
            const double *Rc = (c & 1) ? &R2[9 * (size_t)p] : &R1[9 * (size_t)p];
            const double sg = c < 2 ? 1.0 : -1.0;
            double *P = &poses[12 * (4 * (size_t)p + (size_t)c)];
            for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) P[4 * r + k] = Rc[3 * r + k]; P[4 * r + 3] = sg * tt[3 * (size_t)p + (size_t)r]; }
        }
    }
    hipStream_t st = ctx->stream;
    const size_t nt = (size_t)std::max(n_total, 1);
    if (int rc = ctx->stage_a.reserve(sizeof(float) * 2 * nt)) return rc;
    if (int rc = ctx->stage_b.reserve(sizeof(float) * 2 * nt)) return rc;
    if (int rc = ctx->stage_c.reserve(sizeof(RansacPair) * (size_t)n_pairs)) return rc;
    if (int rc = ctx->stage_d.reserve(sizeof(double) * 48 * (size_t)n_pairs + sizeof(int32_t) * 4 * (size_t)n_pairs)) return rc;
    if (int rc = ctx->stage_e.reserve(5 * nt)) return rc;
    float *d_p1 = ctx->stage_a.as<float>(), *d_p2 = ctx->stage_b.as<float>();
    RansacPair *d_tab = ctx->stage_c.as<RansacPair>();
    double *d_poses = ctx->stage_d.as<double>();
    int32_t *d_good = reinterpret_cast<int32_t *>(d_poses + 48 * (size_t)n_pairs);
    uint8_t *d_cand = ctx->stage_e.as<uint8_t>();
    uint8_t *d_in = d_cand + 4 * nt;
    if (n_total > 0) {
        ESFM_HIP_TRY(esfm::copy_h2d(d_p1, pts1, sizeof(float) * 2 * (size_t)n_total, st));
        ESFM_HIP_TRY(esfm::copy_h2d(d_p2, pts2, sizeof(float) * 2 * (size_t)n_total, st));
        if (mask) ESFM_HIP_TRY(esfm::copy_h2d(d_in, mask, (size_t)n_total, st));
    }
    ESFM_HIP_TRY(esfm::copy_h2d(d_tab, tab.data(), sizeof(RansacPair) * (size_t)n_pairs, st));
    ESFM_HIP_TRY(esfm::copy_h2d(d_poses, poses.data(), sizeof(double) * poses.size(), st));
    if (int rc = esfm::launch_pose_cheirality(st, d_tab, n_pairs, d_p1, d_p2, d_poses, mask ? d_in : nullptr, (int)nt, d_cand, d_good)) return rc;
    std::vector<int32_t> g4(4 * (size_t)n_pairs);
    std::vector<uint8_t> cand(mask ? 4 * nt : 0);
    ESFM_HIP_TRY(esfm::copy_d2h(g4.data(), d_good, sizeof(int32_t) * g4.size(), st));
    if (mask && n_total > 0) ESFM_HIP_TRY(esfm::copy_d2h(cand.data(), d_cand, 4 * nt, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    for (int p = 0; p < n_pairs; ++p) {
        const int32_t *g = &g4[4 * (size_t)p];
        int best;   // recoverPose's if / else-if chain: the first candidate in order that is >= all others
        if (g[0] >= g[1] && g[0] >= g[2] && g[0] >= g[3]) best = 0;
        else if (g[1] >= g[0] && g[1] >= g[2] && g[1] >= g[3]) best = 1;
        else if (g[2] >= g[0] && g[2] >= g[1] && g[2] >= g[3]) best = 2;
        else best = 3;
        const double *Rc = (best & 1) ? &R2[9 * (size_t)p] : &R1[9 * (size_t)p];
        for (int k = 0; k < 9; ++k) R[9 * (size_t)p + k] = Rc[k];
        for (int k = 0; k < 3; ++k) t[3 * (size_t)p + k] = (best < 2 ? 1.0 : -1.0) * tt[3 * (size_t)p + (size_t)k];
        if (good) good[p] = g[best];
        if (mask) memcpy(mask + point_offset[p], cand.data() + (size_t)best * nt + point_offset[p], (size_t)(point_offset[p + 1] - point_offset[p]));
    }
    return ESFM_OK;
}

int esfm_recover_pose(esfm_ctx *ctx, const double *E, const float *pts1, const float *pts2, int n, const float *K4, double *R, double *t,
                      uint8_t *mask, int32_t *good)
{
    if (n < 0) { esfm::set_error("negative point count"); return ESFM_ERR_INVALID_ARG; }
    const int32_t off[2] = {0, n};
    return esfm_recover_pose_pairs(ctx, 1, off, pts1, pts2, K4, E, mask, R, t, good);
}

// The 5-point kernel alone on the GPU: setup + roots kernels on n_samples samples of five normalised correspondences.
int esfm_five_point_models(esfm_ctx *ctx, const double *q1, const double *q2, int n_samples, double *E_out, int32_t *n_models, double *stages)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(n_samples >= 0, "negative sample count");
    if (n_samples == 0) return ESFM_OK;
    ESFM_REQUIRE(q1 && q2 && E_out && n_models, "NULL argument");
    if (int rc = esfm::set_device(ctx)) return rc;
    hipStream_t st = ctx->stream;
    const size_t n = (size_t)n_samples;
    std::vector<double> q(20 * n), mod(90 * n), dbg(stages ? 32 * n : 0);
    for (size_t g = 0; g < n; ++g) for (int k = 0; k < 10; ++k) { q[20 * g + k] = q1[10 * g + k]; q[20 * g + 10 + k] = q2[10 * g + k]; }
    if (int rc = ctx->stage_e.reserve(sizeof(double) * (20 + 90 + 32) * n)) return rc;
    if (int rc = ctx->stage_d.reserve(sizeof(int32_t) * n)) return rc;
    double *d_q = ctx->stage_e.as<double>(), *d_models = d_q + 20 * n, *d_dbg = d_models + 90 * n;
    int32_t *d_nm = ctx->stage_d.as<int32_t>();
    ESFM_HIP_TRY(esfm::copy_h2d(d_q, q.data(), sizeof(double) * 20 * n, st));
    ESFM_HIP_TRY(hipMemsetAsync(d_models, 0, sizeof(double) * 90 * n, st));
    if (int rc = esfm::launch_five_point_setup_samples(st, d_q, n_samples, d_models, d_nm)) return rc;
    if (stages) {
        ESFM_HIP_TRY(esfm::copy_d2h(mod.data(), d_models, sizeof(double) * 90 * n, st));
        ESFM_HIP_TRY(hipMemsetAsync(d_dbg, 0, sizeof(double) * 32 * n, st));
    }
    if (int rc = esfm::launch_five_point_roots(st, n_samples, d_models, d_nm, stages ? d_dbg : nullptr)) return rc;
    ESFM_HIP_TRY(esfm::copy_d2h(E_out, d_models, sizeof(double) * 90 * n, st));
    ESFM_HIP_TRY(esfm::copy_d2h(n_models, d_nm, sizeof(int32_t) * n, st));
    if (stages) ESFM_HIP_TRY(esfm::copy_d2h(dbg.data(), d_dbg, sizeof(double) * 32 * n, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    if (stages)
        for (size_t g = 0; g < n; ++g) {
            double *s = stages + 117 * g;
            const double *w = &mod[90 * g], *d = &dbg[32 * g];
            for (int k = 0; k < 117; ++k) s[k] = 0.0;
            for (int k = 0; k < 36; ++k) s[k] = w[esfm::fivept::kSetupN + k];
            s[116] = -1.0;
            if (d[30] < 0.0) continue;                    // degenerate sample / no degree-10 term: the CPU sides leave the rest zero too
            for (int k = 0; k < 50; ++k) s[36 + k] = w[k];
            for (int k = 0; k < 30; ++k) s[86 + k] = d[k];
            s[116] = d[30];
        }
    return ESFM_OK;
}

// ... and the host build of the same header, estimate after estimate (no GPU): what the CPU suite compares with the oracle bit for bit.
int esfm_five_point_models_host(const double *q1, const double *q2, int n_samples, double *E_out, int32_t *n_models, double *stages)
{
    if (n_samples < 0 || (n_samples && !(q1 && q2 && E_out && n_models))) { esfm::set_error("esfm_five_point_models_host: bad arguments"); return ESFM_ERR_INVALID_ARG; }
    for (size_t g = 0; g < (size_t)n_samples; ++g) {
        int sweeps = -1;
        for (int k = 0; k < 90; ++k) E_out[90 * g + k] = 0.0;
        n_models[g] = esfm::fivept::five_point_models_host(q1 + 10 * g, q2 + 10 * g, E_out + 90 * g, stages ? stages + 117 * g : nullptr, &sweeps);
        if (stages) stages[117 * g + 116] = (double)sweeps;
    }
    return ESFM_OK;
}

// The index stream alone (host arithmetic, no GPU): the first n_samples 5-subsets getSubset draws for `count` points.
int esfm_ransac_sample_stream(int count, int n_samples, int32_t *idx)
{
    if (count < kModelPoints || n_samples < 0 || (n_samples && !idx)) { esfm::set_error("esfm_ransac_sample_stream: bad arguments"); return ESFM_ERR_INVALID_ARG; }
    CvRng rng;
    for (int s = 0; s < n_samples; ++s) draw_subset(rng, count, idx + 5 * (size_t)s);
    return ESFM_OK;
}

}  // extern "C"
