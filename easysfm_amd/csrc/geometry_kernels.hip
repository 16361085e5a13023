// Two-view triangulation for gfx950 (MI355X): the device side of the replacement for cv::triangulatePoints as the
// reference calls it (cpp_code/src/estimate_motion.cpp:263 getDepthFast, :333 doTriangulation).  SURVEY.md section 8 row f-1
// (triangulation part).
//
//   triangulate_dlt_kernel   one thread per correspondence: the 4 x 4 DLT matrix A in f64 (rows x P[2] - P[0], y P[2] - P[1]
//                            per view), then its right singular vector of the smallest singular value (cvSVD's row 3 of V^T)
//                            by one-sided Jacobi rotations on the columns of A, written as 4 floats.
// Embarrassingly parallel, ~600 f64 FLOP and 32 B in / 16 B out per point: latency-bound at the sizes the pipeline
// produces (10^2 .. 10^4 correspondences per pair); the batch entry point takes many pairs in one launch.
#include "common.hpp"

#include <float.h>
#include <math.h>

namespace esfm {

struct TriPair {   // one (P1, P2, point range) job of the batched launch
    float P1[12], P2[12];
    int32_t first, count;
};

__device__ __forceinline__ void triangulate_one(const float *__restrict__ P1, const float *__restrict__ P2, float x1, float y1,
                                                float x2, float y2, float out[4])
{
    double A[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        A[0][k] = (double)x1 * (double)P1[8 + k] - (double)P1[k];
        A[1][k] = (double)y1 * (double)P1[8 + k] - (double)P1[4 + k];
        A[2][k] = (double)x2 * (double)P2[8 + k] - (double)P2[k];
        A[3][k] = (double)y2 * (double)P2[8 + k] - (double)P2[4 + k];
    }
    // One-sided (Hestenes) Jacobi on the columns of A, rotations accumulated in V -- what OpenCV's JacobiSVD does, and the CPU
    // restatement's loop (oracle/geometry_ref.c) to the letter: same sums, same skip rule, same sweep cap, so the vector (its sign
    // included) comes out bit-identical.  (Until round 6: cyclic Jacobi on A'A here -- the squared condition number, another
    // stopping rule; the two sides agreed to 2e-6 on the float vector, up to sign.)
    double V[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) V[p][q] = p == q ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool changed = false;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a += A[r][p] * A[r][p]; b += A[r][q] * A[r][q]; c += A[r][p] * A[r][q]; }
                if (fabs(c) <= DBL_EPSILON * sqrt(a * b) || c == 0.0) continue;
                changed = true;
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double u = A[r][p], v = A[r][q];
                    A[r][p] = cs * u - sn * v; A[r][q] = sn * u + cs * v;
                    const double vu = V[r][p], vv = V[r][q];
                    V[r][p] = cs * vu - sn * vv; V[r][q] = sn * vu + cs * vv;
                }
            }
        if (!changed) break;
    }
    int best = 0;
    double bn = DBL_MAX;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double sq = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) sq += A[r][k] * A[r][k];
        if (sq < bn) { bn = sq; best = k; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = V[r][0];
        v = best == 1 ? V[r][1] : v; v = best == 2 ? V[r][2] : v; v = best == 3 ? V[r][3] : v;
        out[r] = (float)v;
    }
}

__global__ __launch_bounds__(256) void triangulate_dlt_kernel(const TriPair *__restrict__ jobs, int n_jobs, const float2 *__restrict__ pts1,
                                                              const float2 *__restrict__ pts2, int n_total, float4 *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    int lo = 0, hi = n_jobs - 1;   // last job with first <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first <= i) lo = mid; else hi = mid - 1;
    }
    const TriPair &jb = jobs[lo];
    const float2 a = pts1[i], b = pts2[i];
    float r[4];
    triangulate_one(jb.P1, jb.P2, a.x, a.y, b.x, b.y, r);
    out[i] = make_float4(r[0], r[1], r[2], r[3]);
}

int launch_triangulate(hipStream_t st, const void *jobs_dev, int n_jobs, const float *pts1_dev, const float *pts2_dev, int n_total,
                       float *out_dev, esfm_ctx *timing_ctx)
{
    if (n_total <= 0 || n_jobs <= 0) return ESFM_OK;
    {
        KernelTimer tm(timing_ctx, ESFM_K_TRIANGULATE);
        hipLaunchKernelGGL(triangulate_dlt_kernel, dim3((n_total + 255) / 256), dim3(256), 0, st, reinterpret_cast<const TriPair *>(jobs_dev),
                           n_jobs, reinterpret_cast<const float2 *>(pts1_dev), reinterpret_cast<const float2 *>(pts2_dev), n_total,
                           reinterpret_cast<float4 *>(out_dev));
    }
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

size_t triangulate_job_bytes() { return sizeof(TriPair); }

}  // namespace esfm
