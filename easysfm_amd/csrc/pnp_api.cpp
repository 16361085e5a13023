// C-ABI entry point for 3-D/2-D registration (include/esfm.h): the replacement for
// cv::solvePnPRansac(pts3d, pts2d, K, 0, rvec, tvec, false, iterationsCount, reprojectionError, confidence, inliers,
// cv::SOLVEPNP_EPNP) at reference cpp_code/src/estimate_motion.cpp:161-162.
//
// Host side: cv::RNG sample stream and the registrator's bookkeeping (ransac_host.hpp) over chunks of hypotheses that the
// GPU solves (EPnP on 5 points per thread) and scores; then the EPnP re-fit on all inliers, whose reductions over the
// correspondences run on the GPU and whose fixed-size algebra (3 x 3, 12 x 12 eigen-decompositions, betas) runs here from
// the same epnp_core.hpp the kernels use; finally cv::Rodrigues of the rotation.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pnp_kernels.hpp"
#include "ransac_host.hpp"

using esfm::PnpProblem;
namespace rs = esfm::ransac;
namespace ep = esfm::epnp;

namespace {

// inlier sets up to this size are re-fitted on the host (serial sums, the oracle's order: the pose is then bit-identical to the CPU
// restatement's).  8192 covers every frame the pipeline can meet (at most `feature_parameter` <= 8000 keypoints per image, scripts'
// values: sfm.cpp:54); the re-fit is O(n) host work -- 1 505 inliers: as long as the four device reductions' round trips it replaces.
// (1024 until round 6: the bench's own problem, 1 505 inliers, took the looser device path.)
constexpr int kPnpHostRefit = 8192;
constexpr int kPnpChunk = 1024;   // hypotheses per round (one problem at a time: the chunk is what fills the GPU)

// cv::Rodrigues, matrix -> vector [upstream calib3d.cpp]
void rodrigues_to_vec(const double *R, double *rvec)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { rvec[0] = rvec[1] = rvec[2] = 0.0; return; }
        const double t0 = (R[0] + 1) * 0.5, t1 = (R[4] + 1) * 0.5, t2 = (R[8] + 1) * 0.5;
        rx = std::sqrt(std::max(t0, 0.0)); ry = std::sqrt(std::max(t1, 0.0)) * (R[1] < 0 ? -1.0 : 1.0); rz = std::sqrt(std::max(t2, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
        if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        const double k = theta / std::sqrt(rx * rx + ry * ry + rz * rz);
        rvec[0] = rx * k; rvec[1] = ry * k; rvec[2] = rz * k;
        return;
    }
    const double vth = 1.0 / (2.0 * s) * theta;
    rvec[0] = rx * vth; rvec[1] = ry * vth; rvec[2] = rz * vth;
}

}  // namespace

extern "C" {

int esfm_solve_pnp_ransac(esfm_ctx *ctx, const float *pts3d, const float *pts2d, int n, const float *K4, int iterations_count,
                          double reprojection_error, double confidence, double *rvec, double *tvec, double *R_out, uint8_t *inlier_mask,
                          int32_t *n_inliers, int32_t *iterations)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(n >= 0 && pts3d && pts2d && K4 && rvec && tvec, "NULL argument / negative size");
    ESFM_REQUIRE(confidence > 0.0 && confidence < 1.0, "confidence must be in (0, 1)");
    if (n_inliers) *n_inliers = 0;
    if (iterations) *iterations = 0;
    if (n < rs::kModelPoints) {   // OpenCV falls back to P3P at n == 4 and asserts below; neither is built
        esfm::set_error("solvePnPRansac (EPnP) needs at least 5 correspondences (got %d)", n);
        return ESFM_ERR_UNSUPPORTED;
    }
    PnpProblem pb;
    memset(&pb, 0, sizeof(pb));
    pb.n = n; pb.fu = (double)K4[0]; pb.uc = (double)K4[1]; pb.fv = (double)K4[2]; pb.vc = (double)K4[3];
    pb.thresh_sq = (float)(reprojection_error * reprojection_error);
    if (!(std::isfinite(pb.fu) && std::isfinite(pb.fv) && std::isfinite(pb.uc) && std::isfinite(pb.vc))) { esfm::set_error("non-finite intrinsics"); return ESFM_ERR_NUMERIC; }
    if (int rc = esfm::set_device(ctx)) return rc;
    hipStream_t st = ctx->stream;
    const size_t nn = (size_t)n;
    if (int rc = ctx->stage_a.reserve(sizeof(float) * 3 * nn)) return rc;
    if (int rc = ctx->stage_b.reserve(sizeof(float) * 2 * nn)) return rc;
    if (int rc = ctx->stage_c.reserve(sizeof(int32_t) * 7 * (size_t)kPnpChunk)) return rc;                 // samples | valid | counts
    if (int rc = ctx->stage_d.reserve(sizeof(double) * 12 * (size_t)kPnpChunk + sizeof(double) * 256)) return rc;   // poses | small vectors
    if (int rc = ctx->stage_e.reserve(nn + 64)) return rc;
    float *d_p3 = ctx->stage_a.as<float>(), *d_p2 = ctx->stage_b.as<float>();
    int32_t *d_samples = ctx->stage_c.as<int32_t>();
    int32_t *d_valid = d_samples + 5 * (size_t)kPnpChunk, *d_counts = d_valid + kPnpChunk;
    double *d_poses = ctx->stage_d.as<double>();
    double *d_small = d_poses + 12 * (size_t)kPnpChunk;   // [0,12): best pose | [16, 16+64): geo in | [96, 96+96): sums out
    uint8_t *d_mask = ctx->stage_e.as<uint8_t>();
    ESFM_HIP_TRY(esfm::copy_h2d(d_p3, pts3d, sizeof(float) * 3 * nn, st));
    ESFM_HIP_TRY(esfm::copy_h2d(d_p2, pts2d, sizeof(float) * 2 * nn, st));

    // ---- RANSAC over EPnP hypotheses
    rs::CvRng rng;
    int niters = std::max(iterations_count, 1), max_good = 0, iter = 0;
    bool have_best = false;
    std::vector<int32_t> samples(5 * (size_t)kPnpChunk), counts((size_t)kPnpChunk);
    if (n == rs::kModelPoints) {
        // count == modelPoints: the kernel runs once on all points and every point is an inlier
        for (int j = 0; j < 5; ++j) samples[(size_t)j] = j;
        ESFM_HIP_TRY(esfm::copy_h2d(d_samples, samples.data(), sizeof(int32_t) * 5, st));
        if (int rc = esfm::launch_pnp_chunk(st, pb, d_p3, d_p2, d_samples, 1, d_poses, d_valid, d_counts, esfm::kPnpFullSweeps, false, ctx)) return rc;
        int32_t ok = 0;
        ESFM_HIP_TRY(esfm::copy_d2h(&ok, d_valid, sizeof(int32_t), st));
        ESFM_HIP_TRY(hipStreamSynchronize(st));
        if (ok) { ESFM_HIP_TRY(hipMemcpyAsync(d_small, d_poses, sizeof(double) * 12, hipMemcpyDeviceToDevice, st)); have_best = true; max_good = n; }
        ESFM_HIP_TRY(hipMemsetAsync(d_mask, 1, nn, st));
    } else {
        while (iter < niters) {
            // The first rounds are small: with the usual inlier ratios the adaptive count falls below 64 after the first good hypothesis,
            // and a launch lasts as long as its SLOWEST hypothesis (a degenerate sample runs the 12 x 12 Jacobi into its 30-sweep cap:
            // ~2 ms) -- one of 64 samples is rarely that, one of 1024 nearly always.  Which hypotheses count is decided by `iter <
            // niters` below, in the sample stream's order: the chunking changes no result.
            const int chunk = iter == 0 ? 64 : (iter < 320 ? 256 : kPnpChunk);
            const int n_hyp = std::min(chunk, niters - iter);
            for (int k = 0; k < n_hyp; ++k) rs::draw_subset(rng, n, &samples[5 * (size_t)k]);
            ESFM_HIP_TRY(esfm::copy_h2d(d_samples, samples.data(), sizeof(int32_t) * 5 * (size_t)n_hyp, st));
            // first pass with a short sweep budget: the one hypothesis in a hundred whose diagonalisation stalls (all 30 sweeps: ~2 ms against
            // 0.5) comes back unfinished (count -1) instead of holding the launch, and is solved in full below only if the replay reaches it
            // before the adaptive count ends the search -- the same arithmetic on the same sample then, so nothing changes but the time
            static const int first_sweeps = [] { const char *e = getenv("ESFM_PNP_FIRST_SWEEPS"); return e ? std::max(1, atoi(e)) : esfm::kPnpFirstSweeps; }();   // (ESFM_PNP_FIRST_SWEEPS=30: no deferral -- A/B and tests)
            if (int rc = esfm::launch_pnp_chunk(st, pb, d_p3, d_p2, d_samples, n_hyp, d_poses, d_valid, d_counts, first_sweeps, false, ctx)) return rc;
            ESFM_HIP_TRY(esfm::copy_d2h(counts.data(), d_counts, sizeof(int32_t) * (size_t)n_hyp, st));
            ESFM_HIP_TRY(hipStreamSynchronize(st));
            int best_k = -1;
            for (int k = 0; k < n_hyp && iter < niters; ++k, ++iter) {
                if (counts[(size_t)k] < 0) {
                    // every unfinished hypothesis the replay can still reach -- niters only ever shrinks -- in ONE launch (side by side, as
                    // they ran before they were deferred; one after the other they would cost 2 ms each)
                    const int reach = std::min(n_hyp - k, niters - iter);
                    if (int rc = esfm::launch_pnp_chunk(st, pb, d_p3, d_p2, d_samples + 5 * (size_t)k, reach, d_poses + 12 * (size_t)k, d_valid + k, d_counts + k,
                                                        esfm::kPnpFullSweeps, true, ctx)) return rc;
                    ESFM_HIP_TRY(esfm::copy_d2h(&counts[(size_t)k], d_counts + k, sizeof(int32_t) * (size_t)reach, st));
                    ESFM_HIP_TRY(hipStreamSynchronize(st));
                }
                const int good = counts[(size_t)k];   // 0 for a sample whose pose is not finite (runKernel returned no model)
                if (good > std::max(max_good, rs::kModelPoints - 1)) {
                    best_k = k; max_good = good;
                    niters = rs::update_num_iters(confidence, (double)(n - good) / n, rs::kModelPoints, niters);
                }
            }
            if (best_k >= 0) {
                ESFM_HIP_TRY(hipMemcpyAsync(d_small, d_poses + 12 * (size_t)best_k, sizeof(double) * 12, hipMemcpyDeviceToDevice, st));
                have_best = true;
            }
        }
        if (have_best) { if (int rc = esfm::launch_pnp_mask(st, pb, d_p3, d_p2, d_small, d_mask)) return rc; }
    }
    if (iterations) *iterations = iter;
    if (!have_best || max_good <= 0) {
        esfm::set_error("solvePnPRansac: no pose with at least 5 inliers");
        return ESFM_ERR_NUMERIC;   // OpenCV returns false and releases `inliers`
    }
    std::vector<uint8_t> mask(nn);
    ESFM_HIP_TRY(esfm::copy_d2h(mask.data(), d_mask, nn, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));

    // ---- EPnP on all inliers (solvePnP(opoints_inliers, ipoints_inliers, ..., SOLVEPNP_EPNP))
    // A small inlier set is re-fitted here on the host, every sum serial and in index order (epnp::solve_n = the CPU restatement's
    // epnp_pose to the letter): small sets are the ill-conditioned ones -- a dozen correspondences, half of them poor -- and there the
    // last-bit differences of the device reductions below were amplified into poses a percent apart (tests/stress_pnp.py); it also saves
    // the four reduction round trips.  Large sets are well conditioned and stay on the device (agreement to 1e-7).
    {
        int m_host = 0;
        for (size_t i = 0; i < nn; ++i) m_host += mask[i] ? 1 : 0;
        if (m_host >= rs::kModelPoints && m_host <= kPnpHostRefit) {
            std::vector<double> w(3 * (size_t)m_host), px(2 * (size_t)m_host), al(4 * (size_t)m_host), pc(3 * (size_t)m_host);
            for (size_t i = 0, k = 0; i < nn; ++i)
                if (mask[i]) { for (int c = 0; c < 3; ++c) w[3 * k + c] = (double)pts3d[3 * i + c]; px[2 * k] = (double)pts2d[2 * i]; px[2 * k + 1] = (double)pts2d[2 * i + 1]; ++k; }
            const ep::Cam cam = {pb.fu, pb.fv, pb.uc, pb.vc};
            double Rb[9], tb[3];
            ep::solve_n(cam, w.data(), px.data(), m_host, al.data(), pc.data(), Rb, tb);
            for (int k = 0; k < 9; ++k) if (!std::isfinite(Rb[k])) { esfm::set_error("EPnP re-fit produced a non-finite pose"); return ESFM_ERR_NUMERIC; }
            for (int k = 0; k < 3; ++k) if (!std::isfinite(tb[k])) { esfm::set_error("EPnP re-fit produced a non-finite pose"); return ESFM_ERR_NUMERIC; }
            rodrigues_to_vec(Rb, rvec);
            for (int k = 0; k < 3; ++k) tvec[k] = tb[k];
            if (R_out) for (int k = 0; k < 9; ++k) R_out[k] = Rb[k];
            if (inlier_mask) memcpy(inlier_mask, mask.data(), nn);
            if (n_inliers) *n_inliers = m_host;
            return ESFM_OK;
        }
    }
    double *d_geo = d_small + 16, *d_sums = d_small + 96;
    double h[96];
    if (int rc = esfm::launch_pnp_moment_sums(st, pb, d_p3, d_mask, d_sums)) return rc;
    ESFM_HIP_TRY(esfm::copy_d2h(h, d_sums, sizeof(double) * 13, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    const int m = (int)std::lrint(h[12]);
    double sum_pw[3] = {h[0], h[1], h[2]}, cws[4][3], geo[51];
    ep::control_points(sum_pw, h + 3, m, cws, geo + 3);
    for (int k = 0; k < 3; ++k) geo[k] = cws[0][k];
    ESFM_HIP_TRY(esfm::copy_h2d(d_geo, geo, sizeof(double) * 12, st));
    if (int rc = esfm::launch_pnp_mtm_sums(st, pb, d_p3, d_p2, d_mask, d_geo, d_sums)) return rc;
    ESFM_HIP_TRY(esfm::copy_d2h(h, d_sums, sizeof(double) * 78, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    double MtM[144];
    {
        int e = 0;
        for (int a = 0; a < 12; ++a) for (int b = a; b < 12; ++b) { MtM[12 * a + b] = h[e]; MtM[12 * b + a] = h[e]; ++e; }
    }
    double v[4][12], betas[3][4];
    ep::betas_from_mtm(MtM, cws, v, betas);
    // solve_for_sign looks at the first correspondence's camera-frame depth
    int first = 0;
    while (first < n && !mask[(size_t)first]) ++first;
    double a0[4];
    {
        const double w[3] = {(double)pts3d[3 * (size_t)first], (double)pts3d[3 * (size_t)first + 1], (double)pts3d[3 * (size_t)first + 2]};
        ep::alphas_of(cws[0], geo + 3, w, a0);
    }
    for (int c = 0; c < 3; ++c) {
        double ccs[4][3];
        ep::ccs_of(betas[c], v, ccs);
        for (int q = 0; q < 4; ++q) for (int j = 0; j < 3; ++j) geo[12 + 12 * c + 3 * q + j] = ccs[q][j];
        const double z0 = a0[0] * ccs[0][2] + a0[1] * ccs[1][2] + a0[2] * ccs[2][2] + a0[3] * ccs[3][2];
        geo[48 + c] = z0 < 0.0 ? -1.0 : 1.0;
    }
    ESFM_HIP_TRY(esfm::copy_h2d(d_geo, geo, sizeof(double) * 51, st));
    if (int rc = esfm::launch_pnp_rt_sums(st, pb, d_p3, d_mask, d_geo, d_sums)) return rc;
    ESFM_HIP_TRY(esfm::copy_d2h(h, d_sums, sizeof(double) * 36, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    double poses3[36];
    for (int c = 0; c < 3; ++c) ep::rt_from_sums(m, h + 12 * c, sum_pw, h + 12 * c + 3, poses3 + 12 * c, poses3 + 12 * c + 9);
    ESFM_HIP_TRY(esfm::copy_h2d(d_geo, poses3, sizeof(double) * 36, st));
    if (int rc = esfm::launch_pnp_reproj_sums(st, pb, d_p3, d_p2, d_mask, d_geo, d_sums)) return rc;
    ESFM_HIP_TRY(esfm::copy_d2h(h, d_sums, sizeof(double) * 3, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    int N = 0;   // compute_pose: candidate 1, then 2 / 3 only if strictly better
    if (h[1] / m < h[0] / m) N = 1;
    if (h[2] / m < h[N] / m) N = 2;
    const double *Rb = poses3 + 12 * N, *tb = Rb + 9;
    for (int k = 0; k < 12; ++k) if (!std::isfinite(Rb[k])) { esfm::set_error("EPnP re-fit produced a non-finite pose"); return ESFM_ERR_NUMERIC; }
    rodrigues_to_vec(Rb, rvec);
    for (int k = 0; k < 3; ++k) tvec[k] = tb[k];
    if (R_out) for (int k = 0; k < 9; ++k) R_out[k] = Rb[k];
    if (inlier_mask) memcpy(inlier_mask, mask.data(), nn);
    if (n_inliers) *n_inliers = m;
    return ESFM_OK;
}

}  // extern "C"
