// ORB on gfx950 (MI355X): the device side of the replacement for cv::ORB::create(max_num)->detect + ->compute
// (reference FeatureMatching::detectFeaturesORB, cpp_code/src/feature_matching.cpp:14-41, called at cpp_code/test/sfm.cpp:116).
// The steps and their arithmetic are stated in the header of oracle/orb_ref.c; every integer expression here is that one and the
// few float ones keep its operation order (-ffp-contract=off), so keypoints and descriptors agree with the CPU restatement
// bit for bit.  The image work is byte traffic over ~1.5 x the image (8 pyramid levels): one thread per pixel or per keypoint,
// coalesced rows; the selection between the stages (retainBest) is a sort of a few thousand candidates and stays on the host,
// as OpenCV's does.
//
//   orb_resize_kernel    pyramid level from the level above it (bilinear, weights in 1/256)
//   orb_blur_kernel      7 x 7 Gaussian of a level (integer weights, both passes in one kernel: the sums are exact)
//   orb_fast_kernel      FAST-9/16 corner score of every pixel of every level
//   orb_nms_kernel       3 x 3 non-maximum suppression + border filter, candidates appended through one atomic counter
//   orb_harris_kernel    Harris response of the candidates that survived the first selection
//   orb_angle_kernel     intensity-centroid orientation (one wave per keypoint)
//   orb_describe_kernel  256 rotated intensity tests on the blurred level (one thread per descriptor byte)
#include "orb_kernels.hpp"

#include <float.h>

namespace esfm {

__device__ __forceinline__ int orb_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
__device__ __forceinline__ int orb_round_f(float v) { return (int)rintf(v); }

__global__ __launch_bounds__(256) void orb_resize_kernel(const uint8_t *__restrict__ src, int sr, int sc, uint8_t *__restrict__ dst, int dr, int dc)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dc) return;
    const double fxs = (double)sc / dc, fys = (double)sr / dr;
    double fx = (x + 0.5) * fxs - 0.5, fy = (y + 0.5) * fys - 0.5;
    int ix = (int)floor(fx), iy = (int)floor(fy);
    fx -= ix; fy -= iy;
    if (ix < 0) { ix = 0; fx = 0; }
    if (ix >= sc - 1) { ix = sc - 1; fx = 0; }
    if (iy < 0) { iy = 0; fy = 0; }
    if (iy >= sr - 1) { iy = sr - 1; fy = 0; }
    const int ax = (int)rint(fx * 256.0), ay = (int)rint(fy * 256.0);
    const int ix1 = ix + 1 < sc ? ix + 1 : ix, iy1 = iy + 1 < sr ? iy + 1 : iy;
    const uint8_t *r0 = src + (size_t)iy * sc, *r1 = src + (size_t)iy1 * sc;
    const int h0 = r0[ix] * (256 - ax) + r0[ix1] * ax, h1 = r1[ix] * (256 - ax) + r1[ix1] * ax;
    dst[(size_t)y * dc + x] = (uint8_t)((h0 * (256 - ay) + h1 * ay + 32768) >> 16);
}

__global__ __launch_bounds__(256) void orb_blur_kernel(const OrbTables *__restrict__ tab, const uint8_t *__restrict__ src, int rows, int cols,
                                                       uint8_t *__restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= cols) return;
    int w[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) w[k] = tab->gauss[k];
    int xs[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) xs[k] = orb_reflect101(x + k - 3, cols);
    int v = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const uint8_t *r = src + (size_t)orb_reflect101(y + j - 3, rows) * cols;
        int h = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) h += w[k] * r[xs[k]];
        v += w[j] * h;
    }
    dst[(size_t)y * cols + x] = (uint8_t)((v + 32768) >> 16);
}

__device__ __forceinline__ int orb_level_of(const OrbLevels &L, long long e)
{
    int l = 0;
#pragma unroll
    for (int k = 1; k < kOrbLevels; ++k) l += (e >= L.offset[k]) ? 1 : 0;
    return l;
}

// score = 0 for a non-corner, else (largest threshold at which the pixel is still a FAST-9 corner): max over the 16 arcs of nine
// ring pixels of min |centre - ring| on one side, minus one
__global__ __launch_bounds__(256) void orb_fast_kernel(OrbLevels L, const uint8_t *__restrict__ pyr, uint8_t *__restrict__ score)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= L.total) return;
    const int l = orb_level_of(L, e);
    const int R = L.rows[l], C = L.cols[l];
    const long long o = e - L.offset[l];
    const int y = (int)(o / C), x = (int)(o % C);
    int out = 0;
    if (x >= 3 && y >= 3 && x < C - 3 && y < R - 3) {
        const uint8_t *p = pyr + L.offset[l] + (size_t)y * C + x;
        const int v = p[0];
        const int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
        const int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
        int d[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = v - p[dy[k] * C + dx[k]];
        // quick reject (any 9-arc contains one of every pair of opposite pixels ... at least two of the four compass points)
        int best = 0;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            int mn = d[s], mx = d[s];
#pragma unroll
            for (int k = 1; k < 9; ++k) { const int q = d[(s + k) & 15]; mn = min(mn, q); mx = max(mx, q); }
            best = max(best, max(mn, -mx));
        }
        out = best > kOrbFastThreshold ? best - 1 : 0;
    }
    score[e] = (uint8_t)out;
}

__global__ __launch_bounds__(256) void orb_nms_kernel(OrbLevels L, const uint8_t *__restrict__ score, OrbCand *__restrict__ cand,
                                                      int32_t *__restrict__ n_cand, int cap)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= L.total) return;
    const int l = orb_level_of(L, e);
    const int R = L.rows[l], C = L.cols[l];
    const long long o = e - L.offset[l];
    const int y = (int)(o / C), x = (int)(o % C);
    if (x < kOrbEdge || y < kOrbEdge || x >= C - kOrbEdge || y >= R - kOrbEdge) return;
    const uint8_t *p = score + e;
    const int s = p[0];
    if (s == 0) return;
    if (s > p[-1] && s > p[1] && s > p[-C - 1] && s > p[-C] && s > p[-C + 1] && s > p[C - 1] && s > p[C] && s > p[C + 1]) {
        const int slot = atomicAdd(n_cand, 1);
        if (slot < cap) { cand[slot].x = x; cand[slot].y = y; cand[slot].level = l; cand[slot].resp = (float)s; }
    }
}

__global__ __launch_bounds__(256) void orb_harris_kernel(OrbLevels L, const uint8_t *__restrict__ pyr, OrbCand *__restrict__ cand, int n)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const OrbCand c = cand[k];
    const int C = L.cols[c.level];
    const uint8_t *img = pyr + L.offset[c.level];
    int a = 0, b = 0, cc = 0;
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) {
            const uint8_t *p = img + (size_t)(c.y + dy) * C + c.x + dx;
            const int Ix = (p[1] - p[-1]) * 2 + (p[-C + 1] - p[-C - 1]) + (p[C + 1] - p[C - 1]);
            const int Iy = (p[C] - p[-C]) * 2 + (p[C - 1] - p[-C - 1]) + (p[C + 1] - p[-C + 1]);
            a += Ix * Ix; b += Iy * Iy; cc += Ix * Iy;
        }
    const float scale = 1.f / (4 * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    cand[k].resp = ((float)a * (float)b - (float)cc * (float)cc - 0.04f * ((float)a + (float)b) * ((float)a + (float)b)) * scale_sq_sq;
}

__device__ __forceinline__ float orb_fast_atan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846), p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846),
                p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846), p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// one wave per keypoint: lane v handles the row pair +-v of the disc (v = 0: the centre row); integer moments, any order
__global__ __launch_bounds__(64) void orb_angle_kernel(OrbLevels L, const OrbTables *__restrict__ tab, const uint8_t *__restrict__ pyr,
                                                       const OrbCand *__restrict__ cand, int n, float *__restrict__ angles)
{
    const int k = blockIdx.x;
    if (k >= n) return;
    const OrbCand c = cand[k];
    const int C = L.cols[c.level];
    const uint8_t *center = pyr + L.offset[c.level] + (size_t)c.y * C + c.x;
    const int v = threadIdx.x;
    int m01 = 0, m10 = 0;
    if (v == 0) {
        for (int u = -kOrbHalfPatch; u <= kOrbHalfPatch; ++u) m10 += u * center[u];
    } else if (v <= kOrbHalfPatch) {
        const int d = tab->umax[v];
        int vsum = 0;
        for (int u = -d; u <= d; ++u) {
            const int vp = center[u + v * C], vm = center[u - v * C];
            vsum += vp - vm;
            m10 += u * (vp + vm);
        }
        m01 = v * vsum;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m01 += __shfl_xor(m01, o); m10 += __shfl_xor(m10, o); }
    if (v == 0) angles[k] = orb_fast_atan2((float)m01, (float)m10);
}

__global__ __launch_bounds__(256) void orb_describe_kernel(OrbLevels L, const OrbTables *__restrict__ tab, const uint8_t *__restrict__ blurred,
                                                           const OrbKp *__restrict__ kps, int n, uint8_t *__restrict__ desc)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 32 * n) return;
    const int k = e >> 5, by = e & 31;
    const OrbKp kp = kps[k];
    const int R = L.rows[kp.level], C = L.cols[kp.level];
    const uint8_t *img = blurred + L.offset[kp.level];
    int val = 0;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
        const int8_t *pp = tab->pattern + 4 * (8 * by + bit);
        const float x0f = (float)pp[0] * kp.a - (float)pp[1] * kp.b, y0f = (float)pp[0] * kp.b + (float)pp[1] * kp.a;
        const float x1f = (float)pp[2] * kp.a - (float)pp[3] * kp.b, y1f = (float)pp[2] * kp.b + (float)pp[3] * kp.a;
        const int t0 = img[(size_t)orb_reflect101(kp.cy + orb_round_f(y0f), R) * C + orb_reflect101(kp.cx + orb_round_f(x0f), C)];
        const int t1 = img[(size_t)orb_reflect101(kp.cy + orb_round_f(y1f), R) * C + orb_reflect101(kp.cx + orb_round_f(x1f), C)];
        val |= (t0 < t1) << bit;
    }
    desc[e] = (uint8_t)val;
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
#define LAUNCH_OK() ESFM_HIP_TRY(hipGetLastError())

int launch_orb_resize(hipStream_t st, const uint8_t *src, int sr, int sc, uint8_t *dst, int dr, int dc)
{
    hipLaunchKernelGGL(orb_resize_kernel, dim3((dc + 255) / 256, dr), dim3(256), 0, st, src, sr, sc, dst, dr, dc);
    LAUNCH_OK();
    return ESFM_OK;
}
int launch_orb_blur(hipStream_t st, const OrbTables *tab, const uint8_t *src, int rows, int cols, uint8_t *dst)
{
    hipLaunchKernelGGL(orb_blur_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, st, tab, src, rows, cols, dst);
    LAUNCH_OK();
    return ESFM_OK;
}
int launch_orb_fast(hipStream_t st, const OrbLevels &L, const uint8_t *pyr, uint8_t *score, esfm_ctx *timing_ctx)
{
    KernelTimer tm(timing_ctx, ESFM_K_ORB_FAST);
    hipLaunchKernelGGL(orb_fast_kernel, dim3((unsigned)((L.total + 255) / 256)), dim3(256), 0, st, L, pyr, score);
    LAUNCH_OK();
    return ESFM_OK;
}
int launch_orb_nms(hipStream_t st, const OrbLevels &L, const uint8_t *score, OrbCand *cand, int32_t *n_cand, int cap)
{
    hipLaunchKernelGGL(orb_nms_kernel, dim3((unsigned)((L.total + 255) / 256)), dim3(256), 0, st, L, score, cand, n_cand, cap);
    LAUNCH_OK();
    return ESFM_OK;
}
int launch_orb_harris(hipStream_t st, const OrbLevels &L, const uint8_t *pyr, OrbCand *cand, int n)
{
    if (n <= 0) return ESFM_OK;
    hipLaunchKernelGGL(orb_harris_kernel, dim3((n + 255) / 256), dim3(256), 0, st, L, pyr, cand, n);
    LAUNCH_OK();
    return ESFM_OK;
}
int launch_orb_angles(hipStream_t st, const OrbLevels &L, const OrbTables *tab, const uint8_t *pyr, const OrbCand *cand, int n, float *angles)
{
    if (n <= 0) return ESFM_OK;
    hipLaunchKernelGGL(orb_angle_kernel, dim3(n), dim3(64), 0, st, L, tab, pyr, cand, n, angles);
    LAUNCH_OK();
    return ESFM_OK;
}
int launch_orb_describe(hipStream_t st, const OrbLevels &L, const OrbTables *tab, const uint8_t *blurred, const OrbKp *kps, int n, uint8_t *desc)
{
    if (n <= 0) return ESFM_OK;
    hipLaunchKernelGGL(orb_describe_kernel, dim3((32 * n + 255) / 256), dim3(256), 0, st, L, tab, blurred, kps, n, desc);
    LAUNCH_OK();
    return ESFM_OK;
}

}  // namespace esfm
