// Sparse-cloud statistical outlier removal for gfx950 (MI355X): the device side of the replacement for
// CProceesing::SORFilter (reference cpp_code/include/cloudprocessing.hpp:24-36, called at cpp_code/test/sfm.cpp:333), i.e.
// pcl::StatisticalOutlierRemoval with MeanK = 50, StddevMulThresh = 2.0.  SURVEY.md section 8 row f-3.
//
//   sor_knn_mean_kernel   per point: exact (mean_k + 1)-nearest neighbours by the float squared distance
//                         ((dx*dx + dy*dy) + dz*dz), then mean of the sqrt of entries 1..mean_k (entry 0 = the point itself)
//
// One wave owns one query point and streams every candidate past it, 64 per step, out of an LDS tile that the 16 waves of
// the workgroup share.  The wave keeps the 64 smallest distances seen so far as ONE sorted value per lane; a candidate
// only matters if it beats the 64th (a wave-uniform threshold), which after the first few hundred candidates almost none
// does, so the steady state is 3 LDS reads + 8 VALU + a compare and a scalar branch per 64 candidates.  Survivors are
// appended to a per-wave LDS buffer and merged 64 at a time with a bitonic network over the lanes.
// VALU-bound (N^2 / 64 wave-steps); HBM traffic is the cloud itself once per workgroup (L2-resident).
#include "common.hpp"

#include <float.h>
#include <math.h>

namespace esfm {

constexpr int kSorWaves = 16;
constexpr int kSorThreads = kSorWaves * 64;
constexpr int kSorTile = kSorThreads;   // candidates per LDS tile: one per thread

__device__ __forceinline__ float bitonic_sort_desc(float v, int lane)
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float o = __shfl_xor(v, j);
            const bool down = (lane & k) == 0;   // descending block
            const bool lower = (lane & j) == 0;
            v = (lower == down) ? fmaxf(v, o) : fminf(v, o);
        }
    }
    return v;
}

// v is bitonic over the lanes -> ascending
__device__ __forceinline__ float bitonic_merge_asc(float v, int lane)
{
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const float o = __shfl_xor(v, j);
        v = (lane & j) == 0 ? fminf(v, o) : fmaxf(v, o);
    }
    return v;
}

__global__ __launch_bounds__(kSorThreads) void sor_knn_mean_kernel(const float *__restrict__ pts, int n, int stride, int mean_k,
                                                                   float *__restrict__ mean_dist)
{
    __shared__ float tx[kSorTile], ty[kSorTile], tz[kSorTile];
    __shared__ float buf[kSorWaves][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x * kSorWaves + wave;
    const bool has_q = q < n;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (has_q) { qx = pts[(size_t)q * stride]; qy = pts[(size_t)q * stride + 1]; qz = pts[(size_t)q * stride + 2]; }
    const bool q_ok = has_q && isfinite(qx) && isfinite(qy) && isfinite(qz);
    float cur = INFINITY;   // lane l: the (l+1)-th smallest distance so far
    float T = INFINITY;     // = cur of lane 63
    int nbuf = 0;
    float *mybuf = buf[wave];

    auto merge64 = [&](float b) {
        b = bitonic_sort_desc(b, lane);
        cur = bitonic_merge_asc(fminf(cur, b), lane);
        T = __shfl(cur, 63);
    };

    for (int t0 = 0; t0 < n; t0 += kSorTile) {
        __syncthreads();
        {
            const int j = t0 + tid;
            float x = NAN, y = 0.f, z = 0.f;
            if (j < n) {
                x = pts[(size_t)j * stride]; y = pts[(size_t)j * stride + 1]; z = pts[(size_t)j * stride + 2];
                if (!(isfinite(x) && isfinite(y) && isfinite(z))) x = NAN;   // the search structure holds finite points only
            }
            tx[tid] = x; ty[tid] = y; tz[tid] = z;
        }
        __syncthreads();
        if (!q_ok) continue;
        const int cnt = min(kSorTile, n - t0);
        for (int c = 0; c < cnt; c += 64) {
            const int j = c + lane;
            const float dx = qx - tx[j], dy = qy - ty[j], dz = qz - tz[j];
            float d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
            const bool pass = d < T;   // NaN (padding / non-finite candidate) never passes
            const unsigned long long mask = __ballot(pass);
            if (mask == 0ull) continue;
            const int pos = nbuf + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (pass) mybuf[pos] = d;
            nbuf += __popcll(mask);
            if (nbuf >= 64) {
                const float b = mybuf[lane];
                const float hi = mybuf[64 + lane];
                merge64(b);
                nbuf -= 64;
                if (lane < nbuf) mybuf[lane] = hi;
            }
        }
    }
    if (!has_q) return;
    if (!q_ok) { if (lane == 0) mean_dist[q] = 0.0f; return; }
    if (nbuf > 0) merge64(lane < nbuf ? mybuf[lane] : INFINITY);
    // dist_sum += sqrt(nn_dists[k]), k = 1..mean_k, ascending, double accumulator, float sqrt [upstream]
    double s = 0.0;
    for (int k = 1; k <= mean_k; ++k) {
        const float v = __shfl(cur, k);
        if (v < INFINITY) s += (double)sqrtf(v);
    }
    if (lane == 0) mean_dist[q] = (float)(s / (double)mean_k);
}

int launch_sor_knn_mean(hipStream_t st, const float *pts_dev, int n, int stride, int mean_k, float *mean_dist_dev, esfm_ctx *timing_ctx)
{
    if (n <= 0) return ESFM_OK;
    const int grid = (n + kSorWaves - 1) / kSorWaves;
    {
        KernelTimer tm(timing_ctx, ESFM_K_SOR_KNN);
        hipLaunchKernelGGL(sor_knn_mean_kernel, dim3(grid), dim3(kSorThreads), 0, st, pts_dev, n, stride, mean_k, mean_dist_dev);
    }
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm
