// C-ABI entry points for the sparse-cloud outlier filter (include/esfm.h): the replacement for
// CProceesing::SORFilter = pcl::StatisticalOutlierRemoval (reference cpp_code/include/cloudprocessing.hpp:24-36).
// The k-nearest-neighbour pass runs in cloud_kernels.hip; the global mean / standard deviation / threshold are two
// sequential double sums over N floats and stay on the host in PCL's own order, so the threshold is bit-identical.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace esfm {
int launch_sor_knn_mean(hipStream_t st, const float *pts_dev, int n, int stride, int mean_k, float *mean_dist_dev, esfm_ctx *timing_ctx);
}

extern "C" {

int esfm_sor_mean_distances_dev(esfm_ctx *ctx, const float *points_dev, int n, int stride_floats, int mean_k, float *mean_dist_dev)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(n >= 0 && stride_floats >= 3, "bad size / stride");
    ESFM_REQUIRE(mean_k >= 1 && mean_k <= 63, "mean_k must be in [1, 63] (the wave keeps mean_k + 1 <= 64 neighbours)");
    ESFM_REQUIRE(n == 0 || (points_dev && mean_dist_dev), "NULL device pointer");
    if (int rc = esfm::set_device(ctx)) return rc;
    return esfm::launch_sor_knn_mean(ctx->stream, points_dev, n, stride_floats, mean_k, mean_dist_dev, ctx);
}

int esfm_sor_filter(esfm_ctx *ctx, const float *points, int n, int stride_floats, int mean_k, double std_mul, float *mean_dist,
                    uint8_t *keep, int32_t *n_keep, double *threshold)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(n >= 0 && stride_floats >= 3, "bad size / stride");
    ESFM_REQUIRE(mean_k >= 1 && mean_k <= 63, "mean_k must be in [1, 63] (the wave keeps mean_k + 1 <= 64 neighbours)");
    ESFM_REQUIRE(n == 0 || (points && keep), "NULL pointer");
    if (n_keep) *n_keep = 0;
    if (n == 0) { if (threshold) *threshold = NAN; return ESFM_OK; }
    if (int rc = esfm::set_device(ctx)) return rc;
    const size_t in_bytes = sizeof(float) * (size_t)n * (size_t)stride_floats;
    if (int rc = ctx->stage_a.reserve(in_bytes)) return rc;
    if (int rc = ctx->stage_b.reserve(sizeof(float) * (size_t)n)) return rc;
    ESFM_HIP_TRY(esfm::copy_h2d(ctx->stage_a.ptr, points, in_bytes, ctx->stream));
    if (int rc = esfm::launch_sor_knn_mean(ctx->stream, ctx->stage_a.as<float>(), n, stride_floats, mean_k, ctx->stage_b.as<float>(), ctx)) return rc;
    std::vector<float> local;
    float *md = mean_dist;
    if (!md) { local.resize((size_t)n); md = local.data(); }
    ESFM_HIP_TRY(esfm::copy_d2h(md, ctx->stage_b.ptr, sizeof(float) * (size_t)n, ctx->stream));
    ESFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    // statistical_outlier_removal.hpp: mean and standard deviation of the distance vector, then the cut
    int valid = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = points + (size_t)i * stride_floats;
        valid += std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
    }
    double sum = 0.0, sq_sum = 0.0;
    for (int i = 0; i < n; ++i) { sum += (double)md[i]; sq_sum += (double)(md[i] * md[i]); }
    const double mean = sum / (double)valid;
    const double variance = (sq_sum - sum * sum / (double)valid) / ((double)valid - 1.0);
    const double thr = mean + std_mul * std::sqrt(variance);
    if (threshold) *threshold = thr;
    int kept = 0;
    for (int i = 0; i < n; ++i) { keep[i] = !((double)md[i] > thr); kept += keep[i]; }
    if (n_keep) *n_keep = kept;
    return ESFM_OK;
}

}  // extern "C"
