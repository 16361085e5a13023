// 3-D/2-D registration for gfx950 (MI355X): the device side of the replacement for cv::solvePnPRansac(..., SOLVEPNP_EPNP) as
// MotionEstimator::estimate2D3D_P3P_RANSAC calls it (reference cpp_code/src/estimate_motion.cpp:161-162; once per newly
// registered frame, cpp_code/test/sfm.cpp:288).  SURVEY.md section 8 row f-1.
//
//   pnp_solve_kernel    one thread per RANSAC hypothesis: EPnP on its 5 correspondences (epnp_core.hpp) -> R, t
//   pnp_score_kernel    one workgroup per hypothesis: inliers by squared reprojection distance (float, like the
//                       PnPRansacCallback::computeError / findInliers pair)
//   pnp_mask_kernel     inlier mask of the winning hypothesis
//   pnp_*_sums_kernel   the reductions over the inlier set that the final EPnP re-fit needs (second moments, M'M, the
//                       absolute-orientation sums and the reprojection errors of the three beta candidates); the fixed-size
//                       algebra between them (3 x 3 / 12 x 12 eigen-decompositions, betas) runs on the host from the same
//                       epnp_core.hpp routines
// The sample stream and the best-model / adaptive-iteration bookkeeping are replayed on the host exactly as for the
// essential-matrix RANSAC (ransac_api.cpp).
#ifdef ESFM_PNP_TRACE
// stage stamps of pnp_solve_kernel's chain (hypothesis 0 of workgroup 0 prints them): s_memrealtime ticks of 10 ns
#include <hip/hip_runtime.h>
__device__ unsigned long long g_pnp_mark[8];
#if defined(__HIP_DEVICE_COMPILE__)
#define EPNP_MARK(k) do { g_pnp_mark[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#endif
#include "pnp_kernels.hpp"

#include <float.h>

namespace esfm {

using epnp::Cam;

// A hypothesis is SIXTEEN lanes (round 4).  EPnP is a chain of small dense factorisations whose arithmetic and order are the host
// routines' (epnp_core.hpp), so that the oracle's iteration counts and masks are reproduced; all sixteen lanes run that chain
// redundantly -- same operands, same results, same stores -- except inside the one expensive step, the Jacobi diagonalisation of the
// 12 x 12 M'M (66 rotations per sweep, ~8 sweeps, 36 element updates per rotation), where lane r takes row / column r of a rotation's
// three update loops (Jacobi12Coop): the same operations on the same operands, three LDS round trips per rotation instead of 144.
// The four 12 x 12 work arrays (M'M, the eigenvectors, the iteration's A and V) live in LDS, one arena per hypothesis.
// History: one thread per hypothesis with the arrays as thread-private scratch memory 16.9 ms per launch of 1024 hypotheses (152 of
// the 180 ms of GPU time of a run_fountain_small.sh reconstruction); the arrays in LDS, still one thread: 9.9 ms.
constexpr int kPnpLanes = 16, kPnpHypPerBlock = 64 / kPnpLanes, kPnpWsStride = 4 * 144 + 1;
struct Jacobi12Coop {
    int l;            // lane of the hypothesis' group
    int sweep_cap;    // epnp::kJacobiSweeps = the routine's own cap; less: a first pass that gives up on the few matrices whose off-diagonal norm stalls
    bool *gave_up;    // ... and says so here (the caller discards the hypothesis and, if the RANSAC replay needs it, solves it again in full)
    __device__ __forceinline__ void operator()(double *A, double *V) const
    {
        constexpr int N = 12;
        // the group's lanes exchange rows and columns through the arena: DS operations of a wave execute in order; the fences keep the
        // compiler from moving accesses across (no instruction is emitted for a wavefront-scope fence)
        auto sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
        const bool mine = l < N;
        const int r = mine ? l : 0;
        sync();
        if (mine) for (int j = 0; j < N; ++j) V[r * N + j] = r == j ? 1.0 : 0.0;
        sync();
        for (int sweep = 0; sweep < epnp::kJacobiSweeps; ++sweep) {
            double off = 0.0, diag = 0.0;
            for (int i = 0; i < N; ++i) { diag += A[i * N + i] * A[i * N + i]; for (int j = i + 1; j < N; ++j) off += A[i * N + j] * A[i * N + j]; }
            if (off <= epnp::kJacobiOff * diag || off == 0.0) break;
            if (sweep >= sweep_cap) { *gave_up = true; break; }
            for (int p = 0; p < N - 1; ++p)
                for (int q = p + 1; q < N; ++q) {
                    const double apq = A[p * N + q];
                    if (apq == 0.0) continue;
                    const double th = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
                    double c, s;
                    epnp::jacobi_cs(th, c, s);
                    sync();
                    if (mine) { const double x = A[r * N + p], y = A[r * N + q]; A[r * N + p] = c * x - s * y; A[r * N + q] = s * x + c * y; }
                    sync();
                    if (mine) { const double x = A[p * N + r], y = A[q * N + r]; A[p * N + r] = c * x - s * y; A[q * N + r] = s * x + c * y; }
                    sync();
                    if (mine) { const double x = V[r * N + p], y = V[r * N + q]; V[r * N + p] = c * x - s * y; V[r * N + q] = s * x + c * y; }
                    sync();
                }
        }
    }
};

__global__ __launch_bounds__(64) void pnp_solve_kernel(PnpProblem pb, const float *__restrict__ p3, const float *__restrict__ p2,
                                                       const int32_t *__restrict__ samples, int n_hyp, double *__restrict__ poses,
                                                       int32_t *__restrict__ valid, int sweep_cap, int only_unfinished)
{
    __shared__ double ws_all[kPnpHypPerBlock * kPnpWsStride];
    const int grp = threadIdx.x / kPnpLanes, l = threadIdx.x % kPnpLanes;
    const int g = blockIdx.x * kPnpHypPerBlock + grp;
    if (g >= n_hyp) return;
    if (only_unfinished && valid[g] != kPnpUnfinished) return;      // the completion launch: everything else of the range is settled
    const int32_t *id = samples + 5 * (size_t)g;
    double pw[15], us[10];
    for (int k = 0; k < 5; ++k) {
        const int i = id[k];
        pw[3 * k] = (double)p3[3 * (size_t)i]; pw[3 * k + 1] = (double)p3[3 * (size_t)i + 1]; pw[3 * k + 2] = (double)p3[3 * (size_t)i + 2];
        us[2 * k] = (double)p2[2 * (size_t)i]; us[2 * k + 1] = (double)p2[2 * (size_t)i + 1];
    }
    const Cam cam = {pb.fu, pb.fv, pb.uc, pb.vc};
    double R[9], t[3];
#ifdef ESFM_PNP_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    bool gave_up = false;
    epnp::solve_small_ws<5>(cam, pw, us, R, t, ws_all + grp * kPnpWsStride, Jacobi12Coop{l, sweep_cap, &gave_up});
#ifdef ESFM_PNP_TRACE
    if (g == 0 && l == 0)
        printf("pnp_solve [10 ns]: control points %llu  M'M %llu  eig12 %llu  L/rho %llu  betas %llu  gauss-newton %llu  candidates %llu\n", g_pnp_mark[0] - tr0,
               g_pnp_mark[1] - g_pnp_mark[0], g_pnp_mark[2] - g_pnp_mark[1], g_pnp_mark[3] - g_pnp_mark[2], g_pnp_mark[4] - g_pnp_mark[3], g_pnp_mark[5] - g_pnp_mark[4],
               g_pnp_mark[6] - g_pnp_mark[5]);
#endif
    if (l != 0) return;
    if (gave_up) { valid[g] = kPnpUnfinished; return; }      // (what the chain made of the half-diagonalised matrix is discarded)
    bool ok = true;
    for (int k = 0; k < 9; ++k) ok = ok && isfinite(R[k]);
    for (int k = 0; k < 3; ++k) ok = ok && isfinite(t[k]);
    valid[g] = ok ? 1 : 0;
    double *dst = poses + 12 * (size_t)g;
    for (int k = 0; k < 9; ++k) dst[k] = R[k];
    for (int k = 0; k < 3; ++k) dst[9 + k] = t[k];
}

// projectPoints on float input returns Point2f; the residual and its squared norm are float arithmetic
// (PnPRansacCallback::computeError), cut at (float)(reprojectionError^2)
__device__ __forceinline__ bool pnp_inlier(const PnpProblem &pb, const double *P, const float *__restrict__ p3, const float *__restrict__ p2, int i)
{
    const double X = (double)p3[3 * (size_t)i], Y = (double)p3[3 * (size_t)i + 1], Z = (double)p3[3 * (size_t)i + 2];
    const double xc = P[0] * X + P[1] * Y + P[2] * Z + P[9], yc = P[3] * X + P[4] * Y + P[5] * Z + P[10], zc = P[6] * X + P[7] * Y + P[8] * Z + P[11];
    const double iz = zc != 0.0 ? 1.0 / zc : 1.0;
    const float u = (float)(pb.fu * (xc * iz) + pb.uc), v = (float)(pb.fv * (yc * iz) + pb.vc);
    const float dx = p2[2 * (size_t)i] - u, dy = p2[2 * (size_t)i + 1] - v;
    return dx * dx + dy * dy <= pb.thresh_sq;
}

__global__ __launch_bounds__(256) void pnp_score_kernel(PnpProblem pb, const float *__restrict__ p3, const float *__restrict__ p2,
                                                        const double *__restrict__ poses, const int32_t *__restrict__ valid, int32_t *__restrict__ counts,
                                                        int only_unscored)
{
    __shared__ int red[4];
    const int g = blockIdx.x;
    if (only_unscored && counts[g] >= 0) return;
    if (valid[g] == kPnpUnfinished) { if (threadIdx.x == 0) counts[g] = -1; return; }
    if (!valid[g]) { if (threadIdx.x == 0) counts[g] = 0; return; }
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = poses[12 * (size_t)g + k];
    int cnt = 0;
    for (int i = threadIdx.x; i < pb.n; i += 256) cnt += pnp_inlier(pb, P, p3, p2, i) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[g] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void pnp_mask_kernel(PnpProblem pb, const float *__restrict__ p3, const float *__restrict__ p2,
                                                       const double *__restrict__ pose, uint8_t *__restrict__ mask)
{
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = pose[k];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < pb.n; i += gridDim.x * 256) mask[i] = pnp_inlier(pb, P, p3, p2, i) ? 1 : 0;
}

// K per-thread partial sums -> out[K], one workgroup of 256, fixed reduction order (deterministic)
template <int K> __device__ __forceinline__ void block_reduce_store(double (&acc)[K], double *__restrict__ out)
{
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < K; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (threadIdx.x == 0) out[k] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

// out[0..2] = sum pw, out[3..11] = sum pw pw' over the masked correspondences, out[12] = their number
__global__ __launch_bounds__(256) void pnp_moment_sums_kernel(PnpProblem pb, const float *__restrict__ p3, const uint8_t *__restrict__ mask,
                                                              double *__restrict__ out)
{
    double acc[13];
    for (int k = 0; k < 13; ++k) acc[k] = 0.0;
    for (int i = threadIdx.x; i < pb.n; i += 256) {
        if (!mask[i]) continue;
        const double w[3] = {(double)p3[3 * (size_t)i], (double)p3[3 * (size_t)i + 1], (double)p3[3 * (size_t)i + 2]};
        for (int a = 0; a < 3; ++a) { acc[a] += w[a]; for (int b = 0; b < 3; ++b) acc[3 + 3 * a + b] += w[a] * w[b]; }
        acc[12] += 1.0;
    }
    block_reduce_store<13>(acc, out);
}

// geo = c0 (3) | CCi (9).  out = M'M upper triangle, row-major packed (78)
__global__ __launch_bounds__(256) void pnp_mtm_sums_kernel(PnpProblem pb, const float *__restrict__ p3, const float *__restrict__ p2,
                                                           const uint8_t *__restrict__ mask, const double *__restrict__ geo, double *__restrict__ out)
{
    double acc[78];
    for (int k = 0; k < 78; ++k) acc[k] = 0.0;
    const Cam cam = {pb.fu, pb.fv, pb.uc, pb.vc};
    double c0[3], CCi[9];
    for (int k = 0; k < 3; ++k) c0[k] = geo[k];
    for (int k = 0; k < 9; ++k) CCi[k] = geo[3 + k];
    for (int i = threadIdx.x; i < pb.n; i += 256) {
        if (!mask[i]) continue;
        const double w[3] = {(double)p3[3 * (size_t)i], (double)p3[3 * (size_t)i + 1], (double)p3[3 * (size_t)i + 2]};
        double as[4], m1[12], m2[12];
        epnp::alphas_of(c0, CCi, w, as);
        epnp::m_rows(cam, as, (double)p2[2 * (size_t)i], (double)p2[2 * (size_t)i + 1], m1, m2);
        int e = 0;
#pragma unroll
        for (int a = 0; a < 12; ++a)
#pragma unroll
            for (int b = a; b < 12; ++b) acc[e++] += m1[a] * m1[b] + m2[a] * m2[b];
    }
    block_reduce_store<78>(acc, out);
}

// geo = c0 | CCi | ccs of the 3 candidates (3 x 12) | sign of each (3).  out[c][0..2] = sum pc, out[c][3..11] = sum pc pw'
__global__ __launch_bounds__(256) void pnp_rt_sums_kernel(PnpProblem pb, const float *__restrict__ p3, const uint8_t *__restrict__ mask,
                                                          const double *__restrict__ geo, double *__restrict__ out)
{
    double acc[36];
    for (int k = 0; k < 36; ++k) acc[k] = 0.0;
    double c0[3], CCi[9];
    for (int k = 0; k < 3; ++k) c0[k] = geo[k];
    for (int k = 0; k < 9; ++k) CCi[k] = geo[3 + k];
    for (int i = threadIdx.x; i < pb.n; i += 256) {
        if (!mask[i]) continue;
        const double w[3] = {(double)p3[3 * (size_t)i], (double)p3[3 * (size_t)i + 1], (double)p3[3 * (size_t)i + 2]};
        double as[4];
        epnp::alphas_of(c0, CCi, w, as);
        for (int c = 0; c < 3; ++c) {
            const double *cc = geo + 12 + 12 * c;
            const double sg = geo[48 + c];
            for (int a = 0; a < 3; ++a) {
                const double pc = sg * (as[0] * cc[a] + as[1] * cc[3 + a] + as[2] * cc[6 + a] + as[3] * cc[9 + a]);
                acc[12 * c + a] += pc;
                for (int b = 0; b < 3; ++b) acc[12 * c + 3 + 3 * a + b] += pc * w[b];
            }
        }
    }
    block_reduce_store<36>(acc, out);
}

// poses3 = 3 x (R | t); out[c] = sum of reprojection distances over the masked correspondences
__global__ __launch_bounds__(256) void pnp_reproj_sums_kernel(PnpProblem pb, const float *__restrict__ p3, const float *__restrict__ p2,
                                                              const uint8_t *__restrict__ mask, const double *__restrict__ poses3, double *__restrict__ out)
{
    double acc[3] = {0.0, 0.0, 0.0};
    const Cam cam = {pb.fu, pb.fv, pb.uc, pb.vc};
    for (int i = threadIdx.x; i < pb.n; i += 256) {
        if (!mask[i]) continue;
        const double w[3] = {(double)p3[3 * (size_t)i], (double)p3[3 * (size_t)i + 1], (double)p3[3 * (size_t)i + 2]};
        for (int c = 0; c < 3; ++c) acc[c] += epnp::reproj_dist(cam, poses3 + 12 * c, poses3 + 12 * c + 9, w, (double)p2[2 * (size_t)i], (double)p2[2 * (size_t)i + 1]);
    }
    block_reduce_store<3>(acc, out);
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
#define LAUNCH_OK() ESFM_HIP_TRY(hipGetLastError())

int launch_pnp_chunk(hipStream_t st, const PnpProblem &pb, const float *p3, const float *p2, const int32_t *samples, int n_hyp, double *poses,
                     int32_t *valid, int32_t *counts, int sweep_cap, bool only_unfinished, esfm_ctx *timing_ctx)
{
    if (n_hyp <= 0) return ESFM_OK;
    KernelTimer tm(timing_ctx, ESFM_K_RANSAC);
    hipLaunchKernelGGL(pnp_solve_kernel, dim3((n_hyp + kPnpHypPerBlock - 1) / kPnpHypPerBlock), dim3(64), 0, st, pb, p3, p2, samples, n_hyp, poses, valid,
                       sweep_cap, only_unfinished ? 1 : 0);
    LAUNCH_OK();
    hipLaunchKernelGGL(pnp_score_kernel, dim3(n_hyp), dim3(256), 0, st, pb, p3, p2, poses, valid, counts, only_unfinished ? 1 : 0);
    LAUNCH_OK();
    return ESFM_OK;
}

int launch_pnp_mask(hipStream_t st, const PnpProblem &pb, const float *p3, const float *p2, const double *pose, uint8_t *mask)
{
    if (pb.n <= 0) return ESFM_OK;
    hipLaunchKernelGGL(pnp_mask_kernel, dim3(std::min((pb.n + 255) / 256, 1024)), dim3(256), 0, st, pb, p3, p2, pose, mask);
    LAUNCH_OK();
    return ESFM_OK;
}

int launch_pnp_moment_sums(hipStream_t st, const PnpProblem &pb, const float *p3, const uint8_t *mask, double *out)
{
    hipLaunchKernelGGL(pnp_moment_sums_kernel, dim3(1), dim3(256), 0, st, pb, p3, mask, out);
    LAUNCH_OK();
    return ESFM_OK;
}

int launch_pnp_mtm_sums(hipStream_t st, const PnpProblem &pb, const float *p3, const float *p2, const uint8_t *mask, const double *geo, double *out)
{
    hipLaunchKernelGGL(pnp_mtm_sums_kernel, dim3(1), dim3(256), 0, st, pb, p3, p2, mask, geo, out);
    LAUNCH_OK();
    return ESFM_OK;
}

int launch_pnp_rt_sums(hipStream_t st, const PnpProblem &pb, const float *p3, const uint8_t *mask, const double *geo, double *out)
{
    hipLaunchKernelGGL(pnp_rt_sums_kernel, dim3(1), dim3(256), 0, st, pb, p3, mask, geo, out);
    LAUNCH_OK();
    return ESFM_OK;
}

int launch_pnp_reproj_sums(hipStream_t st, const PnpProblem &pb, const float *p3, const float *p2, const uint8_t *mask, const double *poses3, double *out)
{
    hipLaunchKernelGGL(pnp_reproj_sums_kernel, dim3(1), dim3(256), 0, st, pb, p3, p2, mask, poses3, out);
    LAUNCH_OK();
    return ESFM_OK;
}

}  // namespace esfm
