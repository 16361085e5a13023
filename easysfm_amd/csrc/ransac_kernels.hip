// Two-view geometric verification for gfx950 (MI355X): the device side of the replacement for
// cv::findEssentialMat(..., CV_RANSAC, ...) + cv::recoverPose as MotionEstimator::estimate2D2D_E5P_RANSAC calls them
// (reference cpp_code/src/estimate_motion.cpp:49-67; once per matched image pair, cpp_code/test/sfm.cpp:165).
// SURVEY.md section 8 row f-1.
//
// RANSAC hypotheses are data-parallel: the sample stream of cv::RNG is replayed on the host (it is a 64-bit
// multiply-with-carry recurrence, sequential by nature and a few microseconds long), every hypothesis of a chunk of
// iterations of every pair is then solved and scored concurrently, and the host replays OpenCV's sequential
// best-model / adaptive-iteration-count bookkeeping on the inlier counts -- the result is the one the sequential loop
// reaches, independent of the chunk size.
//
//   essential_setup_kernel   one thread per (pair, iteration): 5-point kernel up to the degree-10 polynomial
//   essential_roots_kernel   sixteen lanes per (pair, iteration): its roots -> up to 10 essential matrices
//   essential_score_kernel   one workgroup per (pair, iteration): Sampson inlier count of each of its models
//   essential_mask_kernel    one workgroup per pair: inlier mask of the winning model
//   pose_cheirality_kernel   one workgroup per (pair, candidate pose): f64 DLT triangulation + cheirality test per point
//
// All arithmetic in f64 like OpenCV; the error is rounded to float before the threshold test like
// RANSACPointSetRegistrator::findInliers.
#include "ransac_kernels.hpp"
#include "five_point_core.hpp"

#include <float.h>
#include <math.h>

namespace esfm {

// The 5-point kernel's arithmetic is five_point_core.hpp (host + device; restated to the letter by oracle/ransac_ref.c): the kernels
// below only decide which lane runs which piece of it.
using fivept::kSetupDet;
using fivept::kSetupN;
using fivept::kSetupP;
using fivept::kSetupQ;
using fivept::kSetupR;
constexpr int kSetupLanes = 32;
using LdsVec = fivept::Store<kSetupLanes>;              // entry e of this lane: lds[e * kSetupLanes + lane]

__device__ __forceinline__ void normalise_pt(const RansacPair &pr, const float2 *__restrict__ p1, const float2 *__restrict__ p2, int i,
                                             double &x1, double &y1, double &x2, double &y2)
{
    const float2 a = p1[pr.first + i], b = p2[pr.first + i];
    x1 = ((double)a.x - pr.cx) / pr.fx; y1 = ((double)a.y - pr.cy) / pr.fy;
    x2 = ((double)b.x - pr.cx) / pr.fx; y2 = ((double)b.y - pr.cy) / pr.fy;
}

// value of lane J of the caller's row of 16 lanes: one v_mov_b64_dpp (row_newbcast) instead of the two ds_bpermute_b32 and their LDS
// round trip that __shfl(v, J, 16) costs.  (The s_nop covers the DPP read-after-VALU-write hazard, which the compiler does not track
// through inline asm.)
#ifdef ESFM_DK_HIST
// timing / diagnosis build: histogram of Durand-Kerner sweeps per hypothesis (scratch/dk_hist.py)
__device__ unsigned int g_dk_hist[301];
extern "C" int esfm_debug_dk_hist(unsigned int *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dk_hist), sizeof(g_dk_hist)); }
#endif
template <int J>
__device__ __forceinline__ double row16_bcast(double v)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
    return r;
}
template <int J>
__device__ __forceinline__ void dk_factor(double re, double im, int i, double &dr, double &di)
{
    double tr = dr, ti = di;
    fivept::dk_times(tr, ti, re - row16_bcast<J>(re), im - row16_bcast<J>(im));
    dr = J == i ? dr : tr; di = J == i ? di : ti;
}

// one thread per (pair, iteration of this chunk): the polynomial system of its sample -> models[90 g ..] (86 values);
// n_models[g] = 1 when there is one, -1 when the slot is idle or the sample degenerate (essential_roots_kernel turns both into counts)
__global__ __launch_bounds__(kSetupLanes) void essential_setup_kernel(const RansacPair *__restrict__ pairs, int n_pairs, const float2 *__restrict__ p1,
                                                                      const float2 *__restrict__ p2, const int32_t *__restrict__ samples, int chunk,
                                                                      double *__restrict__ models, int32_t *__restrict__ n_models)
{
    __shared__ double lds[200 * kSetupLanes];             // 51 KB: three workgroups per CU, 19 200 hypotheses resident at once
    const int g = blockIdx.x * kSetupLanes + threadIdx.x;
    if (g >= n_pairs * chunk) return;
    const int pi = g / chunk;
    const RansacPair pr = pairs[pi];
    const int32_t *id = samples + 5 * (size_t)g;
    if (!pr.active || id[0] < 0) { n_models[g] = -1; return; }
    double q1[10], q2[10];
    for (int k = 0; k < 5; ++k) normalise_pt(pr, p1, p2, id[k], q1[2 * k], q1[2 * k + 1], q2[2 * k], q2[2 * k + 1]);
    const LdsVec mine{lds + threadIdx.x};
    fivept::null_space(q1, q2, mine);
    n_models[g] = fivept::determinant_polynomial(models + 90 * (size_t)g, mine) ? 1 : -1;
}

// the same for samples given directly as normalised coordinates (esfm_five_point_models: the solver alone, q[20 g ..] = q1[10], q2[10])
__global__ __launch_bounds__(kSetupLanes) void essential_setup_samples_kernel(const double *__restrict__ q, int n, double *__restrict__ models,
                                                                              int32_t *__restrict__ n_models)
{
    __shared__ double lds[200 * kSetupLanes];
    const int g = blockIdx.x * kSetupLanes + threadIdx.x;
    if (g >= n) return;
    double q1[10], q2[10];
    for (int k = 0; k < 10; ++k) { q1[k] = q[20 * (size_t)g + k]; q2[k] = q[20 * (size_t)g + 10 + k]; }
    const LdsVec mine{lds + threadIdx.x};
    fivept::null_space(q1, q2, mine);
    n_models[g] = fivept::determinant_polynomial(models + 90 * (size_t)g, mine) ? 1 : -1;
}

// Roots of the degree-10 determinant and the models they give: SIXTEEN LANES PER HYPOTHESIS, lane i = root estimate i (ten of them).
// Until round 3 this was the tail of a one-lane-per-hypothesis kernel whose Durand-Kerner loop updated the ten estimates one after the
// other, 7.7 us per sweep, and whose launch took as long as its slowest lane: clusters and multiple roots converge linearly and run
// into the cap of 300 sweeps, so 19 200 hypotheses -- 300 waves, one per CU, 63 lanes of most of them idle -- cost 2.3 ms whatever the
// average (timing-only builds: cap 80: 0.95 ms, 40: 0.80 ms, 20: 0.72 ms; results change below ~100).  Here a sweep updates all ten
// estimates at once from the previous sweep's values (Weierstrass / Durand-Kerner in its simultaneous form: the same fixed points,
// the same quadratic convergence at simple roots): p(z_i) by Horner in every lane, the other estimates through DPP row broadcasts, one
// division per lane.  A group stops when every estimate is at rest (step <= 1e-13 of its magnitude, or down at its evaluation noise;
// or after 300 sweeps) and is frozen from then on, so a hypothesis' result does not depend on which others share its wave.  Then lane i polishes its estimate
// on the real axis if it is real to 1e-8 (Newton), back-substitutes (x, y from the null vector of B(z)), forms E and takes the
// output slot given by its rank in (E[0][0], z, i) among the group's valid lanes -- the order the sequential code produced by
// sorting the roots, then the models.
// (dbg, or NULL: 32 doubles per hypothesis for the stage tests -- cc[10], the estimates' re[10], im[10] after the iteration, its sweeps)
__global__ __launch_bounds__(256) void essential_roots_kernel(int n, double *__restrict__ models, int32_t *__restrict__ n_models, double *__restrict__ dbg)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int g = t >> 4, i = t & 15, grp = (threadIdx.x & 63) >> 4;
    const bool in = g < n;
    const size_t gc = in ? (size_t)g : (size_t)(n - 1);
    const double *w = models + 90 * gc;
    const bool have_poly = in && n_models[gc] > 0;
    const double c10 = have_poly ? w[kSetupDet + 10] : 0.0;
    const bool have = have_poly && fabs(c10) > 0.0;
    double cc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) cc[k] = have ? w[kSetupDet + k] / c10 : 0.0;
    double re, im;
    fivept::start_point(fivept::start_radius(cc), i < 10 ? i : 0, re, im);
    bool active = have;                                   // (uniform over the group)
    fivept::Rest rest;
    int my_sweeps = 0;
    for (int it = 0; it < fivept::kMaxSweeps; ++it) {
        if (!__any(active)) break;
        my_sweeps += active ? 1 : 0;
        double pr, pim, qr, qi;
        fivept::poly_eval(cc, re, im, pr, pim);
        double dr = 1.0, di = 0.0;                        // prod_{j != i} (z_i - z_j), j ascending
        dk_factor<0>(re, im, i, dr, di); dk_factor<1>(re, im, i, dr, di); dk_factor<2>(re, im, i, dr, di); dk_factor<3>(re, im, i, dr, di);
        dk_factor<4>(re, im, i, dr, di); dk_factor<5>(re, im, i, dr, di); dk_factor<6>(re, im, i, dr, di); dk_factor<7>(re, im, i, dr, di);
        dk_factor<8>(re, im, i, dr, di); dk_factor<9>(re, im, i, dr, di);
        fivept::dk_step(pr, pim, dr, di, qr, qi);
        const bool upd = active && i < 10;
        re = upd ? re - qr : re; im = upd ? im - qi : im;
        const bool still = fivept::dk_moving(rest, qr, qi, re, im) && upd;      // (the at-rest rule: five_point_core.hpp)
        const unsigned long long moving = __ballot(still);
        if (((moving >> (16 * grp)) & 0xffffull) == 0ull) active = false;
    }
#ifdef ESFM_DK_HIST
    if (have && i == 0) atomicAdd(&g_dk_hist[my_sweeps], 1u);
#endif
    if (dbg && in && i < 10) {
        double *d = dbg + 32 * (size_t)g;
        d[i] = cc[i]; d[10 + i] = re; d[20 + i] = im;
        if (i == 0) d[30] = have ? (double)my_sweeps : -1.0;
    }
    // the estimate's model: Newton on the real axis, B(z), its null vector, E
    double Ev[9], z;
    const bool valid = fivept::model_from_root(cc, w, re, im, Ev, z) && have && i < 10;
    // output slot: rank in (E[0][0], z, i) among the group's valid lanes
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const double k0 = __shfl(Ev[0], j, 16), kz = __shfl(z, j, 16);
        const int vj = __shfl((int)valid, j, 16);
        rank += (vj && fivept::model_precedes(k0, kz, j, Ev[0], z, i)) ? 1 : 0;
    }
    const unsigned long long vm = __ballot(valid);
    const int count = __popcll((vm >> (16 * grp)) & 0xffffull);
    // (every lane of the group is past its reads of the setup values: the models take their place)
    if (valid) {
        double *dst = models + 90 * gc + 9 * rank;
#pragma unroll
        for (int a = 0; a < 9; ++a) dst[a] = Ev[a];
    }
    if (in && i == 0) n_models[g] = have ? count : 0;
}

__device__ __forceinline__ bool sampson_inlier(const double *E, double x1, double y1, double x2, double y2, float t)
{
    const double Ex0 = E[0] * x1 + E[1] * y1 + E[2], Ex1 = E[3] * x1 + E[4] * y1 + E[5], Ex2 = E[6] * x1 + E[7] * y1 + E[8];
    const double Et0 = E[0] * x2 + E[3] * y2 + E[6], Et1 = E[1] * x2 + E[4] * y2 + E[7];
    const double v = x2 * Ex0 + y2 * Ex1 + Ex2;
    const float err = (float)(v * v / (Ex0 * Ex0 + Ex1 * Ex1 + Et0 * Et0 + Et1 * Et1));
    return err <= t;
}

// the correspondences in normalised coordinates (x1, y1, x2, y2), once per call: one workgroup per pair
__global__ __launch_bounds__(256) void essential_normalise_kernel(const RansacPair *__restrict__ pairs, const float2 *__restrict__ p1,
                                                                  const float2 *__restrict__ p2, double4 *__restrict__ npts)
{
    const RansacPair pr = pairs[blockIdx.x];
    for (int i = threadIdx.x; i < pr.count; i += 256) {
        double x1, y1, x2, y2;
        normalise_pt(pr, p1, p2, i, x1, y1, x2, y2);
        npts[pr.first + i] = make_double4(x1, y1, x2, y2);
    }
}

int launch_essential_normalise(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, double *npts)
{
    if (n_pairs <= 0) return ESFM_OK;
    hipLaunchKernelGGL(essential_normalise_kernel, dim3(n_pairs), dim3(256), 0, st, pairs, reinterpret_cast<const float2 *>(p1),
                       reinterpret_cast<const float2 *>(p2), reinterpret_cast<double4 *>(npts));
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

// one workgroup per (pair, iteration): counts[g][m] = inliers of model m
__global__ __launch_bounds__(256) void essential_score_kernel(const RansacPair *__restrict__ pairs, const double4 *__restrict__ npts, int chunk,
                                                              const double *__restrict__ models, const int32_t *__restrict__ n_models,
                                                              int32_t *__restrict__ counts)
{
    __shared__ int red[10][4];
    __shared__ double sE[90];
    const int g = blockIdx.x;
    const int nm = __builtin_amdgcn_readfirstlane(n_models[g]);
    if (nm <= 0) return;
    const RansacPair pr = pairs[g / chunk];
    for (int k = threadIdx.x; k < 9 * nm; k += 256) sE[k] = models[90 * (size_t)g + k];
    if (threadIdx.x < 40) red[threadIdx.x >> 2][threadIdx.x & 3] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Four correspondences per thread and sweep in registers (normalised once per call by essential_normalise_kernel: the four f64
    // divisions per correspondence were repeated for every hypothesis of every round), then the hypothesis' models one after the other
    // in a REAL loop over the nm of them, each with its nine entries in registers.  (Until round 3: correspondences outside, the ten
    // model slots unrolled inside under `m < nm` -- every workgroup evaluated all ten slots under lane masks, nine LDS reads per
    // slot and correspondence: 131 us per round against 79.)
    for (int i0 = threadIdx.x; i0 < pr.count; i0 += 4 * 256) {
        double4 v[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { ok[u] = i0 + 256 * u < pr.count; v[u] = ok[u] ? npts[pr.first + i0 + 256 * u] : make_double4(0.0, 0.0, 0.0, 0.0); }
#pragma unroll 1
        for (int m = 0; m < nm; ++m) {
            double E[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) E[k] = sE[9 * m + k];
            int c = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) c += (ok[u] && sampson_inlier(E, v[u].x, v[u].y, v[u].z, v[u].w, pr.thresh_sq)) ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
            if (lane == 0) red[m][wave] += c;              // (this wave's own slot)
        }
    }
    __syncthreads();
    if (threadIdx.x < 10) counts[10 * (size_t)g + threadIdx.x] = threadIdx.x < nm ? red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3] : 0;
}

// one workgroup per pair: inlier mask of its best model (best[9 * pair])
__global__ __launch_bounds__(256) void essential_mask_kernel(const RansacPair *__restrict__ pairs, const float2 *__restrict__ p1,
                                                             const float2 *__restrict__ p2, const double *__restrict__ best,
                                                             uint8_t *__restrict__ mask)
{
    const RansacPair pr = pairs[blockIdx.x];
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = best[9 * (size_t)blockIdx.x + k];
    for (int i = threadIdx.x; i < pr.count; i += 256) {
        double x1, y1, x2, y2;
        normalise_pt(pr, p1, p2, i, x1, y1, x2, y2);
        mask[pr.first + i] = sampson_inlier(E, x1, y1, x2, y2, pr.thresh_sq) ? 1 : 0;
    }
}

// smallest right singular vector of the 4 x 4 DLT system of ([I|0], P) in double (cv::triangulatePoints on CV_64F input)
__device__ void triangulate_f64(const double *P, double x0, double y0, double x1, double y1, double X[4])
{
    double A[4][4];
    A[0][0] = -1.0; A[0][1] = 0.0; A[0][2] = x0; A[0][3] = 0.0;
    A[1][0] = 0.0; A[1][1] = -1.0; A[1][2] = y0; A[1][3] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { A[2][k] = x1 * P[8 + k] - P[k]; A[3][k] = y1 * P[8 + k] - P[4 + k]; }
    double M[4][4], V[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) s += A[r][p] * A[r][q];
            M[p][q] = s; V[p][q] = p == q ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 100; ++sweep) {          // (cap and stopping rule: the CPU restatement's jacobi_eig to the letter)
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            diag += M[p][p] * M[p][p];
#pragma unroll
            for (int q = p + 1; q < 4; ++q) off += M[p][q] * M[p][q];
        }
        if (off <= 1e-40 * diag || off == 0.0) break;
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
                for (int r = 0; r < 4; ++r) { const double mp = M[r][p], mq = M[r][q]; M[r][p] = c * mp - s * mq; M[r][q] = s * mp + c * mq; }
#pragma unroll
                for (int r = 0; r < 4; ++r) { const double mp = M[p][r], mq = M[q][r]; M[p][r] = c * mp - s * mq; M[q][r] = s * mp + c * mq; }
#pragma unroll
                for (int r = 0; r < 4; ++r) { const double vp = V[r][p], vq = V[r][q]; V[r][p] = c * vp - s * vq; V[r][q] = s * vp + c * vq; }
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
        X[r] = v;
    }
}

// grid (4, n_pairs): candidate pose c of pair p.  poses[p][c] = 3 x 4 [R | t].  cand_mask[c][point], good[p][c].
__global__ __launch_bounds__(256) void pose_cheirality_kernel(const RansacPair *__restrict__ pairs, const float2 *__restrict__ p1,
                                                              const float2 *__restrict__ p2, const double *__restrict__ poses,
                                                              const uint8_t *__restrict__ in_mask, int n_total, uint8_t *__restrict__ cand_mask,
                                                              int32_t *__restrict__ good)
{
    __shared__ int red[4];
    const int c = blockIdx.x, pi = blockIdx.y;
    const RansacPair pr = pairs[pi];
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = poses[12 * (4 * (size_t)pi + c) + k];
    int cnt = 0;
    for (int i = threadIdx.x; i < pr.count; i += 256) {
        double x0, y0, x1, y1;
        normalise_pt(pr, p1, p2, i, x0, y0, x1, y1);
        double Q[4];
        triangulate_f64(P, x0, y0, x1, y1, Q);
        bool m = Q[2] * Q[3] > 0.0;
        const double X = Q[0] / Q[3], Y = Q[1] / Q[3], Z = Q[2] / Q[3];
        m = m && (Z < pr.dist_thresh);
        const double Z2 = P[8] * X + P[9] * Y + P[10] * Z + P[11];
        m = m && (Z2 > 0.0) && (Z2 < pr.dist_thresh);
        if (in_mask) m = m && in_mask[pr.first + i] != 0;
        cand_mask[(size_t)c * n_total + pr.first + i] = m ? 1 : 0;
        cnt += m ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) good[4 * pi + c] = red[0] + red[1] + red[2] + red[3];
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
int launch_essential_chunk(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, const double *npts,
                           const int32_t *samples, int chunk, double *models, int32_t *n_models, int32_t *counts, esfm_ctx *timing_ctx)
{
    const int n = n_pairs * chunk;
    if (n <= 0) return ESFM_OK;
    KernelTimer tm(timing_ctx, ESFM_K_RANSAC);
    hipLaunchKernelGGL(essential_setup_kernel, dim3((n + kSetupLanes - 1) / kSetupLanes), dim3(kSetupLanes), 0, st, pairs, n_pairs, reinterpret_cast<const float2 *>(p1),
                       reinterpret_cast<const float2 *>(p2), samples, chunk, models, n_models);
    ESFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(essential_roots_kernel, dim3((16 * n + 255) / 256), dim3(256), 0, st, n, models, n_models, (double *)nullptr);
    ESFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(essential_score_kernel, dim3(n), dim3(256), 0, st, pairs, reinterpret_cast<const double4 *>(npts), chunk, models, n_models, counts);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_five_point_setup_samples(hipStream_t st, const double *q, int n, double *models, int32_t *n_models)
{
    if (n <= 0) return ESFM_OK;
    hipLaunchKernelGGL(essential_setup_samples_kernel, dim3((n + kSetupLanes - 1) / kSetupLanes), dim3(kSetupLanes), 0, st, q, n, models, n_models);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_five_point_roots(hipStream_t st, int n, double *models, int32_t *n_models, double *dbg)
{
    if (n <= 0) return ESFM_OK;
    hipLaunchKernelGGL(essential_roots_kernel, dim3((16 * n + 255) / 256), dim3(256), 0, st, n, models, n_models, dbg);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

__global__ __launch_bounds__(256) void essential_take_best_kernel(const int32_t *__restrict__ take, int n_take, const double *__restrict__ models,
                                                                  double *__restrict__ best)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 9 * n_take) return;
    const int e = t / 9, a = t % 9;
    best[9 * (size_t)take[3 * e] + a] = models[90 * (size_t)take[3 * e + 1] + 9 * (size_t)take[3 * e + 2] + a];
}

int launch_essential_take_best(hipStream_t st, const int32_t *take, int n_take, const double *models, double *best)
{
    if (n_take <= 0) return ESFM_OK;
    hipLaunchKernelGGL(essential_take_best_kernel, dim3((9 * n_take + 255) / 256), dim3(256), 0, st, take, n_take, models, best);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_essential_mask(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, const double *best, uint8_t *mask)
{
    if (n_pairs <= 0) return ESFM_OK;
    hipLaunchKernelGGL(essential_mask_kernel, dim3(n_pairs), dim3(256), 0, st, pairs, reinterpret_cast<const float2 *>(p1),
                       reinterpret_cast<const float2 *>(p2), best, mask);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_pose_cheirality(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, const double *poses,
                           const uint8_t *in_mask, int n_total, uint8_t *cand_mask, int32_t *good)
{
    if (n_pairs <= 0) return ESFM_OK;
    hipLaunchKernelGGL(pose_cheirality_kernel, dim3(4, n_pairs), dim3(256), 0, st, pairs, reinterpret_cast<const float2 *>(p1),
                       reinterpret_cast<const float2 *>(p2), poses, in_mask, n_total, cand_mask, good);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm
