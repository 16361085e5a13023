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
typedef float float2v __attribute__((ext_vector_type(2)));

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

// kSorQ query points per wave: a candidate's coordinates are read from LDS once per kSorQ distance evaluations, and the loop
// overhead is shared (one query per wave: 3 LDS reads + loop control per 8 arithmetic instructions; 0.585 ms for 30.6 k points).
constexpr int kSorQ = 4;

__global__ __launch_bounds__(kSorThreads) void sor_knn_mean_kernel(const float *__restrict__ pts, int n, int stride, int mean_k,
                                                                   float *__restrict__ mean_dist)
{
    __shared__ float tx[kSorTile], ty[kSorTile], tz[kSorTile];
    __shared__ float buf[kSorWaves][kSorQ][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = (blockIdx.x * kSorWaves + wave) * kSorQ;
    float qx[kSorQ], qy[kSorQ], qz[kSorQ], cur[kSorQ], T[kSorQ];
    int nbuf[kSorQ];
    bool q_ok[kSorQ];
#pragma unroll
    for (int u = 0; u < kSorQ; ++u) {
        const int q = q0 + u;
        qx[u] = qy[u] = qz[u] = 0.f;
        if (q < n) { qx[u] = pts[(size_t)q * stride]; qy[u] = pts[(size_t)q * stride + 1]; qz[u] = pts[(size_t)q * stride + 2]; }
        q_ok[u] = q < n && isfinite(qx[u]) && isfinite(qy[u]) && isfinite(qz[u]);
        cur[u] = INFINITY;   // lane l: the (l+1)-th smallest distance so far
        T[u] = q_ok[u] ? INFINITY : -INFINITY;     // = cur of lane 63; -inf: nothing ever passes for a query that is not searched
        nbuf[u] = 0;
    }

    auto merge64 = [&](int u, float b) {
        b = bitonic_sort_desc(b, lane);
        cur[u] = bitonic_merge_asc(fminf(cur[u], b), lane);
        T[u] = __shfl(cur[u], 63);
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
        // Two candidate blocks of 64 per step, their distances as the two halves of packed f32 instructions (v_pk_add_f32 /
        // v_pk_mul_f32: each half an IEEE single operation, bit-identical to the scalar form): 8 arithmetic instructions per 128
        // candidates and query.  Padding candidates are NaN: never pass.
        for (int c = 0; c < kSorTile; c += 128) {
            if (c >= n - t0) break;
            const float2v cx = {tx[c + lane], tx[c + 64 + lane]}, cy = {ty[c + lane], ty[c + 64 + lane]}, cz = {tz[c + lane], tz[c + 64 + lane]};
#pragma unroll
            for (int u = 0; u < kSorQ; ++u) {
                const float2v dx = float2v{qx[u], qx[u]} - cx, dy = float2v{qy[u], qy[u]} - cy, dz = float2v{qz[u], qz[u]} - cz;
                float2v d2 = dx * dx;
                d2 = d2 + dy * dy;
                d2 = d2 + dz * dz;
                const float dd[2] = {d2.x, d2.y};
                float *mybuf = buf[wave][u];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const bool pass = dd[b] < T[u];
                    const unsigned long long mask = __ballot(pass);
                    if (mask == 0ull) continue;
                    const int pos = nbuf[u] + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (pass) mybuf[pos] = dd[b];
                    nbuf[u] += __popcll(mask);
                    if (nbuf[u] >= 64) {
                        const float bb = mybuf[lane];
                        const float hi = mybuf[64 + lane];
                        merge64(u, bb);
                        nbuf[u] -= 64;
                        if (lane < nbuf[u]) mybuf[lane] = hi;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < kSorQ; ++u) {
        const int q = q0 + u;
        if (q >= n) continue;
        if (!q_ok[u]) { if (lane == 0) mean_dist[q] = 0.0f; continue; }
        if (nbuf[u] > 0) merge64(u, lane < nbuf[u] ? buf[wave][u][lane] : INFINITY);
        // dist_sum += sqrt(nn_dists[k]), k = 1..mean_k, ascending, double accumulator, float sqrt [upstream]
        double s = 0.0;
        for (int k = 1; k <= mean_k; ++k) {
            const float v = __shfl(cur[u], k);
            if (v < INFINITY) s += (double)sqrtf(v);
        }
        if (lane == 0) mean_dist[q] = (float)(s / (double)mean_k);
    }
}

int launch_sor_knn_mean(hipStream_t st, const float *pts_dev, int n, int stride, int mean_k, float *mean_dist_dev, esfm_ctx *timing_ctx)
{
    if (n <= 0) return ESFM_OK;
    const int grid = (n + kSorWaves * kSorQ - 1) / (kSorWaves * kSorQ);
    {
        KernelTimer tm(timing_ctx, ESFM_K_SOR_KNN);
        hipLaunchKernelGGL(sor_knn_mean_kernel, dim3(grid), dim3(kSorThreads), 0, st, pts_dev, n, stride, mean_k, mean_dist_dev);
    }
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm
