// Two-view triangulation for gfx950 (MI355X): the device side of the replacement for cv::triangulatePoints as the
// reference calls it (cpp_code/src/estimate_motion.cpp:263 getDepthFast, :333 doTriangulation).  SURVEY.md section 8 row f-1
// (triangulation part).
//
//   triangulate_dlt_kernel   one thread per correspondence: the 4 x 4 DLT matrix A in f64 (rows x P[2] - P[0], y P[2] - P[1]
//                            per view), then the eigenvector of the smallest eigenvalue of A'A by cyclic Jacobi rotations
//                            on the symmetric 4 x 4 (= the right singular vector of the smallest singular value, which
//                            cvSVD returns as row 3 of V^T), written as 4 floats.
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
    // M = A'A (symmetric), V = I
    double M[4][4], V[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) s += A[r][p] * A[r][q];
            M[p][q] = s;
            V[p][q] = p == q ? 1.0 : 0.0;
        }
    // cyclic Jacobi eigenvalue iteration; 4 x 4 converges quadratically, 10 sweeps are far more than needed
    for (int sweep = 0; sweep < 10; ++sweep) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            diag += M[p][p] * M[p][p];
#pragma unroll
            for (int q = p + 1; q < 4; ++q) off += M[p][q] * M[p][q];
        }
        if (off <= 1e-32 * diag) break;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double apq = M[p][q];
                if (apq == 0.0) continue;
                const double theta = (M[q][q] - M[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // M <- M J  (columns p, q)
                    const double mp = M[r][p], mq = M[r][q];
                    M[r][p] = c * mp - s * mq; M[r][q] = s * mp + c * mq;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // M <- J' M  (rows p, q)
                    const double mp = M[p][r], mq = M[q][r];
                    M[p][r] = c * mp - s * mq; M[q][r] = s * mp + c * mq;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double vp = V[r][p], vq = V[r][q];
                    V[r][p] = c * vp - s * vq; V[r][q] = s * vp + c * vq;
                }
            }
    }
    int best = 0;
    double bv = M[0][0];
#pragma unroll
    for (int k = 1; k < 4; ++k) if (M[k][k] < bv) { bv = M[k][k]; best = k; }
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
