// Launch interface between pnp_api.cpp and pnp_kernels.hip.
#pragma once

#include <algorithm>

#include "common.hpp"
#include "epnp_core.hpp"

namespace esfm {

struct PnpProblem {
    int32_t n, pad;
    double fu, fv, uc, vc;
    float thresh_sq, pad2;   // (float)(reprojectionError^2)
};

// Solve + score n_hyp hypotheses.  sweep_cap < kPnpFullSweeps: the 12 x 12 Jacobi diagonalisation gives up after that many sweeps
// (valid = kPnpUnfinished, count = -1) -- a safety net since the routine's threshold became 10 DBL_EPSILON (every matrix of a 4 096-sample
// count converges in 5 - 7 sweeps; with the earlier 1e-36 / 1e-40 one in a hundred / forty stalled and ran into the cap), and a launch lasts as long as its slowest hypothesis; the caller has such hypotheses solved again
// with kPnpFullSweeps (only_unfinished: the launch touches nothing but the entries marked kPnpUnfinished / count -1 of the range it is
// given) only if the RANSAC replay gets as far as needing one.
constexpr int kPnpFullSweeps = epnp::kJacobiSweeps, kPnpFirstSweeps = 12, kPnpUnfinished = 2;
int launch_pnp_chunk(hipStream_t st, const PnpProblem &pb, const float *p3, const float *p2, const int32_t *samples, int n_hyp, double *poses,
                     int32_t *valid, int32_t *counts, int sweep_cap, bool only_unfinished, esfm_ctx *timing_ctx);
int launch_pnp_mask(hipStream_t st, const PnpProblem &pb, const float *p3, const float *p2, const double *pose, uint8_t *mask);
int launch_pnp_moment_sums(hipStream_t st, const PnpProblem &pb, const float *p3, const uint8_t *mask, double *out /*13*/);
int launch_pnp_mtm_sums(hipStream_t st, const PnpProblem &pb, const float *p3, const float *p2, const uint8_t *mask, const double *geo /*12*/,
                        double *out /*78*/);
int launch_pnp_rt_sums(hipStream_t st, const PnpProblem &pb, const float *p3, const uint8_t *mask, const double *geo /*51*/, double *out /*36*/);
int launch_pnp_reproj_sums(hipStream_t st, const PnpProblem &pb, const float *p3, const float *p2, const uint8_t *mask, const double *poses3 /*36*/,
                           double *out /*3*/);

}  // namespace esfm
