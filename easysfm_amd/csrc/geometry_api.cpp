// C-ABI entry points for two-view triangulation (include/esfm.h, SURVEY.md section 8 row f-1, triangulation part): the
// replacement for cv::triangulatePoints at cpp_code/src/estimate_motion.cpp:263 and :333.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace esfm {
int launch_triangulate(hipStream_t st, const void *jobs_dev, int n_jobs, const float *pts1_dev, const float *pts2_dev, int n_total,
                       float *out_dev, esfm_ctx *timing_ctx);
size_t triangulate_job_bytes();
}

namespace {
struct HostJob { float P1[12], P2[12]; int32_t first, count; };
}

extern "C" {

int esfm_triangulate_pairs(esfm_ctx *ctx, int n_pairs, const float *proj1, const float *proj2, const int32_t *point_offset,
                           const float *pts1, const float *pts2, float *points4d)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(n_pairs >= 0, "negative pair count");
    if (n_pairs == 0) return ESFM_OK;
    ESFM_REQUIRE(proj1 && proj2 && point_offset, "NULL argument");
    ESFM_REQUIRE(point_offset[0] == 0, "point_offset[0] must be 0");
    for (int p = 0; p < n_pairs; ++p) ESFM_REQUIRE(point_offset[p + 1] >= point_offset[p], "point_offset must be non-decreasing");
    const int n_total = point_offset[n_pairs];
    if (n_total == 0) return ESFM_OK;
    ESFM_REQUIRE(pts1 && pts2 && points4d, "NULL point arrays");
    static_assert(sizeof(HostJob) == 104, "job layout");
    if (sizeof(HostJob) != esfm::triangulate_job_bytes()) { esfm::set_error("job layout mismatch"); return ESFM_ERR_HIP; }
    if (int rc = esfm::set_device(ctx)) return rc;
    std::vector<HostJob> jobs;
    jobs.reserve((size_t)n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        if (point_offset[p + 1] == point_offset[p]) continue;   // the device search needs strictly increasing `first`
        HostJob j;
        for (int k = 0; k < 12; ++k) {
            j.P1[k] = proj1[12 * (size_t)p + k]; j.P2[k] = proj2[12 * (size_t)p + k];
            if (!std::isfinite(j.P1[k]) || !std::isfinite(j.P2[k])) { esfm::set_error("non-finite projection matrix"); return ESFM_ERR_NUMERIC; }
        }
        j.first = point_offset[p]; j.count = point_offset[p + 1] - point_offset[p];
        jobs.push_back(j);
    }
    hipStream_t st = ctx->stream;
    const size_t pb = sizeof(float) * 2 * (size_t)n_total;
    if (int rc = ctx->stage_a.reserve(pb)) return rc;
    if (int rc = ctx->stage_b.reserve(pb)) return rc;
    if (int rc = ctx->stage_c.reserve(sizeof(float) * 4 * (size_t)n_total)) return rc;
    if (int rc = ctx->stage_d.reserve(sizeof(HostJob) * jobs.size())) return rc;
    ESFM_HIP_TRY(esfm::copy_h2d(ctx->stage_a.ptr, pts1, pb, st));
    ESFM_HIP_TRY(esfm::copy_h2d(ctx->stage_b.ptr, pts2, pb, st));
    ESFM_HIP_TRY(esfm::copy_h2d(ctx->stage_d.ptr, jobs.data(), sizeof(HostJob) * jobs.size(), st));
    if (int rc = esfm::launch_triangulate(st, ctx->stage_d.ptr, (int)jobs.size(), ctx->stage_a.as<float>(), ctx->stage_b.as<float>(), n_total,
                                          ctx->stage_c.as<float>(), ctx))
        return rc;
    ESFM_HIP_TRY(esfm::copy_d2h(points4d, ctx->stage_c.ptr, sizeof(float) * 4 * (size_t)n_total, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    return ESFM_OK;
}

int esfm_triangulate_points(esfm_ctx *ctx, const float *proj1, const float *proj2, const float *pts1, const float *pts2, int n,
                            float *points4d)
{
    if (n < 0) { esfm::set_error("negative point count"); return ESFM_ERR_INVALID_ARG; }
    const int32_t off[2] = {0, n};
    return esfm_triangulate_pairs(ctx, 1, proj1, proj2, off, pts1, pts2, points4d);
}

}  // extern "C"
