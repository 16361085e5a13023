// Launch interface between ransac_api.cpp (cv::RNG replay, RANSAC bookkeeping, pose selection on the host) and ransac_kernels.hip.
#pragma once

#include "common.hpp"

namespace esfm {

struct RansacPair {       // one image pair's correspondences inside the concatenated point arrays
    int32_t first, count;
    int32_t active, pad;  // still iterating in this round
    double fx, cx, fy, cy;
    float thresh_sq;      // (float)((threshold / ((fx + fy) / 2))^2), findInliers' cut on the float error
    float pad2;
    double dist_thresh;   // recoverPose's distanceThresh (50)
};

// npts[4 i ..] = correspondence i in normalised coordinates (x1, y1, x2, y2), doubles; `active` is not looked at
int launch_essential_normalise(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, double *npts);
int launch_essential_chunk(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, const double *npts, const int32_t *samples,
                           int chunk, double *models, int32_t *n_models, int32_t *counts, esfm_ctx *timing_ctx);
// the solver alone on samples given as normalised coordinates (q[20 g ..] = q1[10], q2[10]): setup leaves the 86 intermediate values of
// five_point_core.hpp in models[90 g ..] and n_models[g] = 1 / -1; roots turns them into models and counts (dbg: 32 doubles per sample or NULL)
int launch_five_point_setup_samples(hipStream_t st, const double *q, int n, double *models, int32_t *n_models);
int launch_five_point_roots(hipStream_t st, int n, double *models, int32_t *n_models, double *dbg);
// best[9 take[3 e]] = models[90 take[3 e + 1] + 9 take[3 e + 2]] (9 doubles) for e < n_take
int launch_essential_take_best(hipStream_t st, const int32_t *take, int n_take, const double *models, double *best);
int launch_essential_mask(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, const double *best, uint8_t *mask);
int launch_pose_cheirality(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, const double *poses,
                           const uint8_t *in_mask, int n_total, uint8_t *cand_mask, int32_t *good);

}  // namespace esfm
