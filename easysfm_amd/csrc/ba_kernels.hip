// Bundle-adjustment kernels for gfx950 (MI355X): the device side of the replacement for
// BundleAdjustment::solveBA's ceres::Solve (reference cpp_code/src/ba.cpp:132-212) with the cost
// functor ReprojectErrorTerm_fixcalib (cpp_code/include/ba.h:108-164).  All arithmetic is f64,
// like Ceres; observations and intrinsics are f32 inputs, like the reference (ba.h:162-163).
//
//   ba_linearize_kernel    residual + analytic Jacobian + Cauchy corrector per observation
//                          (the "Jacobian sweep": streams 16 B in, 160 B out per observation),
//                          plus the per-camera F'F / F'r sums
//   ba_point_prep_kernel   per point: E'E, E'r, (E'E + D^2)^-1
//   ba_schur_kernel        per observation: point-block Schur complement into the reduced system
//   ba_chol_solve_kernel   dense Cholesky of the reduced camera system + both triangular solves
//   ba_backsub_*_kernel    back-substitution (observation-parallel, 2 passes), candidate point, model cost change
//   ba_cost_kernel         robustified cost of a parameter vector (candidate evaluation)
//
// The LM control flow (accept/reject, radius) lives in ba_api.cpp.  DESIGN.md "Bundle adjustment".
#include "ba_kernels.hpp"

#include <float.h>

namespace esfm {

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double *lds /*>= 4 doubles*/)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += lds[w];
    return s;
}

// value of lane `l` (wave-uniform index) as a scalar: two v_readlane_b32 instead of the LDS round trip of __shfl
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v)
{
    // non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}

// Sum of entry e over n_slabs per-workgroup slabs, computed by a 32 x 8 thread tile: thread (ent, grp)
// adds slabs grp, grp+8, ... (independent loads, coalesced across ent), then the 8 partial sums are
// combined in a fixed order through LDS.  Deterministic; returns the total to the grp == 0 threads.
constexpr int kRedEnt = 32, kRedGrp = 8;
__device__ __forceinline__ double slab_sum(const double *__restrict__ slabs, int per, int n_slabs, int e, double *lds /*[256]*/)
{
    const int grp = threadIdx.x / kRedEnt;
    double v0 = 0.0, v1 = 0.0;
    if (e < per) {
        int b = grp;
        for (; b + kRedGrp < n_slabs; b += 2 * kRedGrp) {
            v0 += slabs[(size_t)b * per + e];
            v1 += slabs[(size_t)(b + kRedGrp) * per + e];
        }
        if (b < n_slabs) v0 += slabs[(size_t)b * per + e];
    }
    lds[threadIdx.x] = v0 + v1;
    __syncthreads();
    double t = 0.0;
    if (grp == 0)
        for (int g = 0; g < kRedGrp; ++g) t += lds[g * kRedEnt + (threadIdx.x % kRedEnt)];
    return t;
}

// ceres::CauchyLoss::Evaluate [upstream]; a <= 0 selects the trivial (squared) loss.
__device__ __forceinline__ void loss_eval(double a, double s, double &rho0, double &rho1)
{
    if (a <= 0.0) { rho0 = s; rho1 = 1.0; return; }
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c, inv = 1.0 / sum;
    rho0 = b * log(sum);
    rho1 = inv > DBL_MIN ? inv : DBL_MIN;
}

// Rotate-and-translate of ReprojectErrorTerm_fixcalib (ba.h:131-135): p = AngleAxisRotatePoint(a, X) + t,
// with ceres' two branches [upstream rotation.h].  Optionally the derivatives dp/dX (R) and dp/da (Ja):
// with alpha = sin/theta, beta = (1-cos)/theta^2,
//   p = cos X + alpha a x X + beta a (a.X)
//   dp/da_k = -alpha a_k X + gamma a_k (a x X) + alpha (e_k x X) + delta a_k (a.X) a + beta (e_k (a.X) + a X_k)
//   gamma = (cos - alpha)/theta^2, delta = (alpha - 2 beta)/theta^2.
template <bool DERIV>
__device__ __forceinline__ void transform_point(const double *__restrict__ cam, const double *__restrict__ X, double p[3],
                                                double R[9], double Ja[9])
{
    const double a0 = cam[0], a1 = cam[1], a2 = cam[2];
    const double X0 = X[0], X1 = X[1], X2 = X[2];
    const double theta2 = a0 * a0 + a1 * a1 + a2 * a2;
    const double c0 = a1 * X2 - a2 * X1, c1 = a2 * X0 - a0 * X2, c2 = a0 * X1 - a1 * X0;  // a x X
    if (theta2 > DBL_EPSILON) {
        const double theta = sqrt(theta2);
        double s, c;
        sincos(theta, &s, &c);
        const double ti = 1.0 / theta;
        const double w0 = a0 * ti, w1 = a1 * ti, w2 = a2 * ti;
        const double tmp = (w0 * X0 + w1 * X1 + w2 * X2) * (1.0 - c);
        p[0] = X0 * c + (w1 * X2 - w2 * X1) * s + w0 * tmp;
        p[1] = X1 * c + (w2 * X0 - w0 * X2) * s + w1 * tmp;
        p[2] = X2 * c + (w0 * X1 - w1 * X0) * s + w2 * tmp;
        if (DERIV) {
            const double ti2 = ti * ti;
            const double alpha = s * ti, beta = (1.0 - c) * ti2, gamma = (c - alpha) * ti2, delta = (alpha - 2.0 * beta) * ti2;
            const double aX = a0 * X0 + a1 * X1 + a2 * X2;
            R[0] = c + beta * a0 * a0;       R[1] = -alpha * a2 + beta * a0 * a1; R[2] = alpha * a1 + beta * a0 * a2;
            R[3] = alpha * a2 + beta * a1 * a0; R[4] = c + beta * a1 * a1;       R[5] = -alpha * a0 + beta * a1 * a2;
            R[6] = -alpha * a1 + beta * a2 * a0; R[7] = alpha * a0 + beta * a2 * a1; R[8] = c + beta * a2 * a2;
            const double av[3] = {a0, a1, a2}, Xv[3] = {X0, X1, X2}, cv[3] = {c0, c1, c2};
            // e_k x X, k = 0,1,2 (columns)
            const double ex[3][3] = {{0.0, -X2, X1}, {X2, 0.0, -X0}, {-X1, X0, 0.0}};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double ak = av[k];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    double v = -alpha * ak * Xv[i] + gamma * ak * cv[i] + alpha * ex[k][i] + delta * ak * aX * av[i] + beta * av[i] * Xv[k];
                    if (i == k) v += beta * aX;
                    Ja[3 * i + k] = v;
                }
            }
        }
    } else {
        p[0] = X0 + c0; p[1] = X1 + c1; p[2] = X2 + c2;
        if (DERIV) {
            R[0] = 1.0; R[1] = -a2; R[2] = a1; R[3] = a2; R[4] = 1.0; R[5] = -a0; R[6] = -a1; R[7] = a0; R[8] = 1.0;
            Ja[0] = 0.0; Ja[1] = X2;  Ja[2] = -X1;
            Ja[3] = -X2; Ja[4] = 0.0; Ja[5] = X0;
            Ja[6] = X1;  Ja[7] = -X0; Ja[8] = 0.0;
        }
    }
    p[0] += cam[3]; p[1] += cam[4]; p[2] += cam[5];
}

// fx, cx, fy, cy of camera c: the camera's float calibration (ba.h:142-143), or the shared free block, which is
// stored behind the real cameras of the camera-side parameter vector `cams` (ba.h:199-202).
__device__ __forceinline__ void load_intrinsics(const BADev &d, const double *__restrict__ cams, int c, double in4[4])
{
    if (d.has_calib) {
        const double *kp = cams + 6 * (size_t)d.n_real_cam;
        in4[0] = kp[0]; in4[1] = kp[1]; in4[2] = kp[2]; in4[3] = kp[3];
    } else {
        const float4 K = d.K4[c];
        in4[0] = (double)K.x; in4[1] = (double)K.y; in4[2] = (double)K.z; in4[3] = (double)K.w;
    }
}

// Residual and analytic Jacobian of the reprojection functor (ba.h:113-153 / 170-216) at (cam, X):
// r = uv - (x fx + cx, y fy + cy), x = p0/p2, y = p1/p2, p = R(a) X + t.  Jc[2][6], Jp[2][3]; xn, yn = x, y.
__device__ __forceinline__ void reproject_jac(const double cam[6], const double X[3], const double in4[4], float2 uv,
                                              double &r0, double &r1, double Jc[12], double Jp[6], double &xn, double &yn)
{
    double pt[3], R[9], Ja[9];
    transform_point<true>(cam, X, pt, R, Ja);
    const double iz = 1.0 / pt[2];
    const double x = pt[0] * iz, y = pt[1] * iz;
    const double fx = in4[0], cx = in4[1], fy = in4[2], cy = in4[3];
    r0 = (double)uv.x - (x * fx + cx);
    r1 = (double)uv.y - (y * fy + cy);
    // d r / d p  (rows)
    const double g0[3] = {-fx * iz, 0.0, fx * x * iz};
    const double g1[3] = {0.0, -fy * iz, fy * y * iz};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        Jc[j] = g0[0] * Ja[j] + g0[1] * Ja[3 + j] + g0[2] * Ja[6 + j];
        Jc[6 + j] = g1[0] * Ja[j] + g1[1] * Ja[3 + j] + g1[2] * Ja[6 + j];
        Jc[3 + j] = g0[j];
        Jc[9 + j] = g1[j];
        Jp[j] = g0[0] * R[j] + g0[1] * R[3 + j] + g0[2] * R[6 + j];
        Jp[3 + j] = g1[0] * R[j] + g1[1] * R[3 + j] + g1[2] * R[6 + j];
    }
    xn = x; yn = y;
}

// ---------------------------------------------------------------------------------------------
// Jacobian sweep.  PRIV: per-camera sums go through an LDS-private copy first (n_cam * 27 doubles),
// so global f64 atomics are one per camera entry per workgroup instead of 27 per observation.
constexpr int kLinThreads = 512;

template <bool PRIV, bool CALIB>
__global__ __launch_bounds__(kLinThreads) void ba_linearize_kernel(BADev d, double cauchy_a, int use_scaling, double *__restrict__ slabs)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [8] reduction scratch, then PRIV: [n_cam*27]
    double *red = lds;
    double *priv = lds + 8;
    const int tid = threadIdx.x;
    const int n_obs = d.n_obs;
    if (PRIV) {
        for (int e = tid; e < d.n_cam * 27; e += kLinThreads) priv[e] = 0.0;
        __syncthreads();
    }
    double cost = 0.0, bad = 0.0;
    double kacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = blockIdx.x * kLinThreads + tid; k < n_obs; k += gridDim.x * kLinThreads) {
        const int c = d.obs_cam[k], p = d.obs_pt[k];
        const float2 uv = d.obs_uv[k];
        double in4[4];
        load_intrinsics(d, d.x_c, c, in4);
        double cam[6], X[3];
#pragma unroll
        for (int i = 0; i < 6; ++i) cam[i] = d.x_c[6 * (size_t)c + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) X[i] = d.x_p[3 * (size_t)p + i];
        double r0, r1, Jc[12], Jp[6], xn, yn;
        reproject_jac(cam, X, in4, uv, r0, r1, Jc, Jp, xn, yn);
        // intrinsics columns (ba.h:199-206): u = x fx + cx, v = y fy + cy
        double Jk[4] = {-xn, -1.0, -yn, -1.0};
        const double s = r0 * r0 + r1 * r1;
        bool fin = isfinite(s);
#pragma unroll
        for (int i = 0; i < 12; ++i) fin = fin && isfinite(Jc[i]);
#pragma unroll
        for (int i = 0; i < 6; ++i) fin = fin && isfinite(Jp[i]);
        fin = fin && isfinite(xn) && isfinite(yn);
        if (!fin) {
            bad += 1.0; r0 = r1 = 0.0;
#pragma unroll
            for (int i = 0; i < 12; ++i) Jc[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) Jp[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) Jk[i] = 0.0;
        } else {
            double rho0, rho1;
            loss_eval(cauchy_a, s, rho0, rho1);
            cost += 0.5 * rho0;
            const double sq = sqrt(rho1);  // corrector, rho'' <= 0 branch [upstream corrector.cc]
            r0 *= sq; r1 *= sq;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double sc = use_scaling ? sq * d.scale_c[6 * (size_t)c + i] : sq;
                Jc[i] *= sc; Jc[6 + i] *= sc;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double sp = use_scaling ? sq * d.scale_p[3 * (size_t)p + i] : sq;
                Jp[i] *= sp; Jp[3 + i] *= sp;
            }
            if (CALIB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Jk[i] *= use_scaling ? sq * d.scale_c[6 * (size_t)d.n_real_cam + i] : sq;
            }
        }
        if (CALIB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) d.Jk[(size_t)i * n_obs + k] = Jk[i];
            // G'G (upper triangle of the 4x4: rows fx, cx | fy, cy never mix) and G'r, kept in registers over the sweep
            kacc[0] += Jk[0] * Jk[0]; kacc[1] += Jk[0] * Jk[1]; kacc[2] += Jk[1] * Jk[1];
            kacc[3] += Jk[2] * Jk[2]; kacc[4] += Jk[2] * Jk[3]; kacc[5] += Jk[3] * Jk[3];
            kacc[6] += Jk[0] * r0; kacc[7] += Jk[1] * r0; kacc[8] += Jk[2] * r1; kacc[9] += Jk[3] * r1;
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) d.Jc[(size_t)i * n_obs + k] = Jc[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) d.Jp[(size_t)i * n_obs + k] = Jp[i];
        d.res[k] = r0; d.res[(size_t)n_obs + k] = r1;
        // per-camera sums: F'F upper triangle (21) and F'r (6)
        int e = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = a; b < 6; ++b) {
                const double v = Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
                if (PRIV) atomicAdd(&priv[c * 27 + e], v);
                else { atomicAdd(&d.camacc[36 * (size_t)c + 6 * a + b], v); if (a != b) atomicAdd(&d.camacc[36 * (size_t)c + 6 * b + a], v); }
                ++e;
            }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double v = Jc[a] * r0 + Jc[6 + a] * r1;
            if (PRIV) atomicAdd(&priv[c * 27 + 21 + a], v);
            else atomicAdd(&d.camacc[36 * (size_t)d.n_cam + 6 * (size_t)c + a], v);
        }
    }
    const double cs = block_sum(cost, red);
    const double bs = block_sum(bad, red);
    if (tid == 0) { atomicAdd(&d.scal[SC_COST], cs); if (bs > 0.0) atomicAdd(&d.scal[SC_LIN_BAD], bs); }
    if (CALIB) {
        // the intrinsics block's F'F / F'r entries: every observation hits the same 10 sums, so they are reduced
        // over the workgroup instead of through atomics.  Packed index of (a, b), a <= b: a*6 - a(a-1)/2 + b - a.
        const int kc = d.n_real_cam;
        const int slot[10] = {0, 1, 6, 11, 12, 15, 21, 22, 23, 24};
        const int ra[10] = {0, 0, 1, 2, 2, 3, 0, 1, 2, 3}, rb[10] = {0, 1, 1, 2, 3, 3, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const double v = block_sum(kacc[q], red);
            if (tid != 0) continue;
            if (PRIV) priv[kc * 27 + slot[q]] += v;
            else if (q < 6) {
                atomicAdd(&d.camacc[36 * (size_t)kc + 6 * ra[q] + rb[q]], v);
                if (ra[q] != rb[q]) atomicAdd(&d.camacc[36 * (size_t)kc + 6 * rb[q] + ra[q]], v);
            } else atomicAdd(&d.camacc[36 * (size_t)d.n_cam + 6 * (size_t)kc + ra[q]], v);
        }
    }
    if (PRIV) {
        // one coalesced slab per workgroup; ba_camacc_reduce_kernel sums them in a fixed order
        __syncthreads();
        double *out = slabs + (size_t)blockIdx.x * d.n_cam * 27;
        for (int e = tid; e < d.n_cam * 27; e += kLinThreads) out[e] = priv[e];
    }
}

// camacc = sum over slabs, unpacked: F'F (36 per camera, both triangles) | F'r (6 per camera)
__global__ __launch_bounds__(256) void ba_camacc_reduce_kernel(BADev d, const double *__restrict__ slabs, int n_slabs)
{
    __shared__ double lds[256];
    const int e = blockIdx.x * kRedEnt + (threadIdx.x % kRedEnt);
    const int per = d.n_cam * 27;
    const double v = slab_sum(slabs, per, n_slabs, e, lds);
    if (threadIdx.x >= kRedEnt || e >= per) return;
    const int c = e / 27, q = e % 27;
    if (q >= 21) { d.camacc[36 * (size_t)d.n_cam + 6 * (size_t)c + (q - 21)] = v; return; }
    int a = 0, rem = q;
    while (rem >= 6 - a) { rem -= 6 - a; ++a; }
    const int b2 = a + rem;
    d.camacc[36 * (size_t)c + 6 * a + b2] = v;
    d.camacc[36 * (size_t)c + 6 * b2 + a] = v;
}

// ---------------------------------------------------------------------------------------------
// Per point: E'E, E'r (only when the Jacobian is fresh), then M^-1 = (E'E + clamp(diag)/radius)^-1
// via a 3x3 Cholesky (ceres InvertPSDMatrix), M^-1 E'r, and the point part of max|gradient|.
__global__ __launch_bounds__(64) void ba_point_prep_kernel(BADev d, double radius, double min_diag, double max_diag, int fresh)
{
    __shared__ double red[8];
    const int p = blockIdx.x * 64 + threadIdx.x;
    double gmax = 0.0, sing = 0.0;
    if (p < d.n_pt) {
        const int b = d.pt_start[p], e = d.pt_start[p + 1];
        if (e > b) {
            double A[6], g[3];
            if (fresh) {
#pragma unroll
                for (int i = 0; i < 6; ++i) A[i] = 0.0;
                g[0] = g[1] = g[2] = 0.0;
                const size_t n = d.n_obs;
                for (int k = b; k < e; ++k) {
                    const double j0 = d.Jp[k], j1 = d.Jp[n + k], j2 = d.Jp[2 * n + k];
                    const double j3 = d.Jp[3 * n + k], j4 = d.Jp[4 * n + k], j5 = d.Jp[5 * n + k];
                    const double r0 = d.res[k], r1 = d.res[n + k];
                    A[0] += j0 * j0 + j3 * j3; A[1] += j0 * j1 + j3 * j4; A[2] += j0 * j2 + j3 * j5;
                    A[3] += j1 * j1 + j4 * j4; A[4] += j1 * j2 + j4 * j5; A[5] += j2 * j2 + j5 * j5;
                    g[0] += j0 * r0 + j3 * r1; g[1] += j1 * r0 + j4 * r1; g[2] += j2 * r0 + j5 * r1;
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) d.EtE[6 * (size_t)p + i] = A[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    d.Etr[3 * (size_t)p + i] = g[i];
                    gmax = fmax(gmax, fabs(g[i] / d.scale_p[3 * (size_t)p + i]));  // gradient of the unscaled problem
                }
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i) A[i] = d.EtE[6 * (size_t)p + i];
#pragma unroll
                for (int i = 0; i < 3; ++i) g[i] = d.Etr[3 * (size_t)p + i];
            }
            // M = E'E + D^2, D^2 = clamp(diag(E'E)) / radius  (levenberg_marquardt_strategy.cc)
            const double m00 = A[0] + fmin(fmax(A[0], min_diag), max_diag) / radius;
            const double m11 = A[3] + fmin(fmax(A[3], min_diag), max_diag) / radius;
            const double m22 = A[5] + fmin(fmax(A[5], min_diag), max_diag) / radius;
            const double m10 = A[1], m20 = A[2], m21 = A[4];
            double Mi[6] = {0, 0, 0, 0, 0, 0};
            bool ok = m00 > 0.0;
            const double l00 = sqrt(m00);
            const double l10 = m10 / l00, l20 = m20 / l00;
            const double t11 = m11 - l10 * l10;
            ok = ok && (t11 > 0.0);
            const double l11 = sqrt(t11);
            const double l21 = (m21 - l20 * l10) / l11;
            const double t22 = m22 - l20 * l20 - l21 * l21;
            ok = ok && (t22 > 0.0);
            const double l22 = sqrt(t22);
            if (ok) {
                const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
                const double i10 = -l10 * i00 * i11;
                const double i21 = -l21 * i11 * i22;
                const double i20 = -(l20 * i00 + l21 * i10) * i22;
                Mi[0] = i00 * i00 + i10 * i10 + i20 * i20;  // xx
                Mi[1] = i10 * i11 + i20 * i21;              // xy
                Mi[2] = i20 * i22;                          // xz
                Mi[3] = i11 * i11 + i21 * i21;              // yy
                Mi[4] = i21 * i22;                          // yz
                Mi[5] = i22 * i22;                          // zz
            } else {
                sing = 1.0;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) d.Minv[6 * (size_t)p + i] = Mi[i];
            d.Aig[3 * (size_t)p + 0] = Mi[0] * g[0] + Mi[1] * g[1] + Mi[2] * g[2];
            d.Aig[3 * (size_t)p + 1] = Mi[1] * g[0] + Mi[3] * g[1] + Mi[4] * g[2];
            d.Aig[3 * (size_t)p + 2] = Mi[2] * g[0] + Mi[4] * g[1] + Mi[5] * g[2];
        }
    }
    if (fresh) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, o));
        if ((threadIdx.x & 63) == 0 && gmax > 0.0) atomic_max_nonneg(&d.scal[SC_GMAX], gmax);
    }
    const double ss = block_sum(sing, red);
    if (threadIdx.x == 0 && ss > 0.0) atomicAdd(&d.scal[SC_PT_SINGULAR], ss);
}

// Jacobi scaling 1/(1 + sqrt(squared column norm)) from the unscaled linearisation
// (trust_region_minimizer.cc, iteration 0).  Point norms = diag(E'E); camera norms = diag(F'F).
__global__ void ba_jacobi_scaling_kernel(BADev d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * d.n_pt) {
        const int p = i / 3, a = i % 3;
        const int di = a == 0 ? 0 : (a == 1 ? 3 : 5);
        const bool active = d.pt_start[p + 1] > d.pt_start[p];
        d.scale_p[i] = active ? 1.0 / (1.0 + sqrt(d.EtE[6 * (size_t)p + di])) : 1.0;
    }
    if (i < 6 * d.n_cam) {
        const int c = i / 6, a = i % 6;
        d.scale_c[i] = 1.0 / (1.0 + sqrt(d.camacc[36 * (size_t)c + 7 * a]));
    }
}

// Camera part of max|gradient| (needs the all-reduced F'r).
__global__ void ba_camera_gradient_kernel(BADev d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double g = 0.0;
    if (i < 6 * d.n_cam) {
        g = d.camacc[36 * (size_t)d.n_cam + i] / d.scale_c[i];
        if (d.constrained) {
            // bounded problems: norm of x - Plus(x, -gradient)  (trust_region_minimizer.cc, projected gradient step)
            const double x = d.x_c[i];
            g = x - fmin(fmax(x - g, d.lo_c[i]), d.up_c[i]);
        }
        g = fabs(g);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) g = fmax(g, __shfl_xor(g, o));
    if ((threadIdx.x & 63) == 0 && g > 0.0) atomic_max_nonneg(&d.scal[SC_GMAX], g);
}

// ---------------------------------------------------------------------------------------------
// Point-block Schur complement (schur_eliminator_impl.h [upstream]).  Thread i owns observation i of
// point p: W_i = F_i'E_i, Y_i = W_i M^-1; rhs_corr[c_i] -= W_i M^-1 E'r; and for every observation j of
// the same point with camera(j) <= camera(i):  S[c_i][c_j] -= Y_i W_j'.  Only blocks on or below the
// block diagonal are produced (the factorisation reads the lower triangle).
__global__ __launch_bounds__(256) void ba_schur_kernel(BADev d)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n_obs) return;
    const size_t n_obs = d.n_obs;
    const int n = 6 * d.n_cam;
    const int p = d.obs_pt[i], ci = d.obs_cam[i];
    double Jc[12], Jp[6];
#pragma unroll
    for (int a = 0; a < 12; ++a) Jc[a] = d.Jc[a * n_obs + i];
#pragma unroll
    for (int a = 0; a < 6; ++a) Jp[a] = d.Jp[a * n_obs + i];
    const double *Mi = d.Minv + 6 * (size_t)p;
    const double M[9] = {Mi[0], Mi[1], Mi[2], Mi[1], Mi[3], Mi[4], Mi[2], Mi[4], Mi[5]};
    const double ag0 = d.Aig[3 * (size_t)p], ag1 = d.Aig[3 * (size_t)p + 1], ag2 = d.Aig[3 * (size_t)p + 2];
    double Y[18];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const double w0 = Jc[a] * Jp[0] + Jc[6 + a] * Jp[3];
        const double w1 = Jc[a] * Jp[1] + Jc[6 + a] * Jp[4];
        const double w2 = Jc[a] * Jp[2] + Jc[6 + a] * Jp[5];
        atomicAdd(&d.red[(size_t)n * n + 6 * ci + a], -(w0 * ag0 + w1 * ag1 + w2 * ag2));
        Y[3 * a + 0] = w0 * M[0] + w1 * M[3] + w2 * M[6];
        Y[3 * a + 1] = w0 * M[1] + w1 * M[4] + w2 * M[7];
        Y[3 * a + 2] = w0 * M[2] + w1 * M[5] + w2 * M[8];
    }
    const int b = d.pt_start[p], e = d.pt_start[p + 1];
    for (int j = b; j < e; ++j) {
        const int cj = d.obs_cam[j];
        if (cj > ci) continue;
        double Fj[12], Ej[6];
#pragma unroll
        for (int a = 0; a < 12; ++a) Fj[a] = d.Jc[a * n_obs + j];
#pragma unroll
        for (int a = 0; a < 6; ++a) Ej[a] = d.Jp[a * n_obs + j];
        double *Sb = d.red + (size_t)(6 * ci) * n + 6 * cj;
#pragma unroll
        for (int c2 = 0; c2 < 6; ++c2) {
            const double w0 = Fj[c2] * Ej[0] + Fj[6 + c2] * Ej[3];
            const double w1 = Fj[c2] * Ej[1] + Fj[6 + c2] * Ej[4];
            const double w2 = Fj[c2] * Ej[2] + Fj[6 + c2] * Ej[5];
#pragma unroll
            for (int a = 0; a < 6; ++a)
                atomicAdd(&Sb[(size_t)a * n + c2], -(Y[3 * a] * w0 + Y[3 * a + 1] * w1 + Y[3 * a + 2] * w2));
        }
    }
}

// Small-camera-count variant (the metric configuration: 25 cameras): the whole lower-block-triangular
// Schur matrix fits in LDS (n_cam (n_cam+1)/2 blocks of 36 doubles + the 6 n_cam right-hand side), so
// each workgroup accumulates its observations' contributions with LDS f64 atomics (ds_add_f64) and
// then stores its private copy as one coalesced slab; ba_schur_reduce_kernel sums the slabs in a fixed
// order and unpacks them into the n x n layout.  No global atomics.
__global__ __launch_bounds__(1024) void ba_schur_lds_kernel(BADev d, double *__restrict__ slabs, int slab_doubles)
{
    extern __shared__ __attribute__((aligned(16))) double sl[];   // [nblk*36] blocks, then [n] rhs_corr
    const int tid = threadIdx.x;
    const int n = 6 * d.n_cam;
    const int nblk = d.n_cam * (d.n_cam + 1) / 2;
    for (int e = tid; e < slab_doubles; e += 1024) sl[e] = 0.0;
    __syncthreads();
    double *srhs = sl + (size_t)nblk * 36;
    const size_t n_obs = d.n_obs;
    for (int i = blockIdx.x * 1024 + tid; i < d.n_obs; i += gridDim.x * 1024) {
        const int p = d.obs_pt[i], ci = d.obs_cam[i];
        double Jc[12], Jp[6];
#pragma unroll
        for (int a = 0; a < 12; ++a) Jc[a] = d.Jc[a * n_obs + i];
#pragma unroll
        for (int a = 0; a < 6; ++a) Jp[a] = d.Jp[a * n_obs + i];
        const double *Mi = d.Minv + 6 * (size_t)p;
        const double M[9] = {Mi[0], Mi[1], Mi[2], Mi[1], Mi[3], Mi[4], Mi[2], Mi[4], Mi[5]};
        const double ag0 = d.Aig[3 * (size_t)p], ag1 = d.Aig[3 * (size_t)p + 1], ag2 = d.Aig[3 * (size_t)p + 2];
        double Y[18];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double w0 = Jc[a] * Jp[0] + Jc[6 + a] * Jp[3];
            const double w1 = Jc[a] * Jp[1] + Jc[6 + a] * Jp[4];
            const double w2 = Jc[a] * Jp[2] + Jc[6 + a] * Jp[5];
            atomicAdd(&srhs[6 * ci + a], -(w0 * ag0 + w1 * ag1 + w2 * ag2));
            Y[3 * a + 0] = w0 * M[0] + w1 * M[3] + w2 * M[6];
            Y[3 * a + 1] = w0 * M[1] + w1 * M[4] + w2 * M[7];
            Y[3 * a + 2] = w0 * M[2] + w1 * M[5] + w2 * M[8];
        }
        const int b = d.pt_start[p], e = d.pt_start[p + 1];
        for (int j = b; j < e; ++j) {
            const int cj = d.obs_cam[j];
            if (cj > ci) continue;
            double Fj[12], Ej[6];
#pragma unroll
            for (int a = 0; a < 12; ++a) Fj[a] = d.Jc[a * n_obs + j];
#pragma unroll
            for (int a = 0; a < 6; ++a) Ej[a] = d.Jp[a * n_obs + j];
            double *Sb = sl + (size_t)(ci * (ci + 1) / 2 + cj) * 36;
#pragma unroll
            for (int c2 = 0; c2 < 6; ++c2) {
                const double w0 = Fj[c2] * Ej[0] + Fj[6 + c2] * Ej[3];
                const double w1 = Fj[c2] * Ej[1] + Fj[6 + c2] * Ej[4];
                const double w2 = Fj[c2] * Ej[2] + Fj[6 + c2] * Ej[5];
#pragma unroll
                for (int a = 0; a < 6; ++a)
                    atomicAdd(&Sb[a * 6 + c2], -(Y[3 * a] * w0 + Y[3 * a + 1] * w1 + Y[3 * a + 2] * w2));
            }
        }
    }
    __syncthreads();
    double *out = slabs + (size_t)blockIdx.x * slab_doubles;
    for (int e = tid; e < slab_doubles; e += 1024) out[e] = sl[e];
    (void)n;
}

// Large camera counts: the Schur matrix does not fit LDS, but a workgroup that walks points ordered by their
// lowest camera only touches a narrow band of it.  Each workgroup owns one chunk of that point order and an
// LDS window of kWinCams consecutive cameras starting at the chunk's lowest one: contributions whose two
// cameras fall inside the window are accumulated with LDS f64 atomics, the few that do not (wide baselines,
// ring wrap-around) go straight to global f64 atomics; the window is flushed once at the end.
constexpr int kWinCams = 28;
constexpr int kWinBlocks = kWinCams * (kWinCams + 1) / 2;

__global__ __launch_bounds__(1024) void ba_schur_window_kernel(BADev d)
{
    extern __shared__ __attribute__((aligned(16))) double sl[];   // [kWinBlocks*36] blocks, then [6*kWinCams] rhs_corr
    const int tid = threadIdx.x;
    const int n = 6 * d.n_cam;
    const int chunk = blockIdx.x;
    const int s0 = d.chunk_slot[chunk], s1 = d.chunk_slot[chunk + 1];
    const int cw = d.chunk_cam0[chunk];
    constexpr int kWinDoubles = kWinBlocks * 36 + 6 * kWinCams;
    for (int e = tid; e < kWinDoubles; e += 1024) sl[e] = 0.0;
    __syncthreads();
    double *srhs = sl + kWinBlocks * 36;
    const size_t n_obs = d.n_obs;
    for (int s = s0 + tid; s < s1; s += 1024) {
        const int i = d.slot_obs[s];
        const int p = d.obs_pt[i], ci = d.obs_cam[i];
        double Jc[12], Jp[6];
#pragma unroll
        for (int a = 0; a < 12; ++a) Jc[a] = d.Jc[a * n_obs + i];
#pragma unroll
        for (int a = 0; a < 6; ++a) Jp[a] = d.Jp[a * n_obs + i];
        const double *Mi = d.Minv + 6 * (size_t)p;
        const double M[9] = {Mi[0], Mi[1], Mi[2], Mi[1], Mi[3], Mi[4], Mi[2], Mi[4], Mi[5]};
        const double ag0 = d.Aig[3 * (size_t)p], ag1 = d.Aig[3 * (size_t)p + 1], ag2 = d.Aig[3 * (size_t)p + 2];
        const int wi = ci - cw;
        const bool in_i = wi >= 0 && wi < kWinCams;
        double Y[18];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double w0 = Jc[a] * Jp[0] + Jc[6 + a] * Jp[3];
            const double w1 = Jc[a] * Jp[1] + Jc[6 + a] * Jp[4];
            const double w2 = Jc[a] * Jp[2] + Jc[6 + a] * Jp[5];
            const double rv = -(w0 * ag0 + w1 * ag1 + w2 * ag2);
            if (in_i) atomicAdd(&srhs[6 * wi + a], rv); else atomicAdd(&d.red[(size_t)n * n + 6 * ci + a], rv);
            Y[3 * a + 0] = w0 * M[0] + w1 * M[3] + w2 * M[6];
            Y[3 * a + 1] = w0 * M[1] + w1 * M[4] + w2 * M[7];
            Y[3 * a + 2] = w0 * M[2] + w1 * M[5] + w2 * M[8];
        }
        const int b = d.pt_start[p], e = d.pt_start[p + 1];
        for (int j = b; j < e; ++j) {
            const int cj = d.obs_cam[j];
            if (cj > ci) continue;
            double Fj[12], Ej[6];
#pragma unroll
            for (int a = 0; a < 12; ++a) Fj[a] = d.Jc[a * n_obs + j];
#pragma unroll
            for (int a = 0; a < 6; ++a) Ej[a] = d.Jp[a * n_obs + j];
            const int wj = cj - cw;
            const bool in_w = in_i && wj >= 0;   // cj <= ci < cw + kWinCams
            double *Sl = sl + (size_t)(wi * (wi + 1) / 2 + wj) * 36;
            double *Sg = d.red + (size_t)(6 * ci) * n + 6 * cj;
#pragma unroll
            for (int c2 = 0; c2 < 6; ++c2) {
                const double w0 = Fj[c2] * Ej[0] + Fj[6 + c2] * Ej[3];
                const double w1 = Fj[c2] * Ej[1] + Fj[6 + c2] * Ej[4];
                const double w2 = Fj[c2] * Ej[2] + Fj[6 + c2] * Ej[5];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const double v = -(Y[3 * a] * w0 + Y[3 * a + 1] * w1 + Y[3 * a + 2] * w2);
                    if (in_w) atomicAdd(&Sl[a * 6 + c2], v); else atomicAdd(&Sg[(size_t)a * n + c2], v);
                }
            }
        }
    }
    __syncthreads();
    // flush the window
    for (int e = tid; e < kWinBlocks * 36; e += 1024) {
        const double v = sl[e];
        if (v == 0.0) continue;
        const int blk = e / 36, r = e % 36;
        int wi = (int)((sqrt(8.0 * (double)blk + 1.0) - 1.0) * 0.5);
        while ((wi + 1) * (wi + 2) / 2 <= blk) ++wi;
        while (wi * (wi + 1) / 2 > blk) --wi;
        const int wj = blk - wi * (wi + 1) / 2;
        const int ci = cw + wi, cj = cw + wj;
        if (ci < d.n_cam) atomicAdd(&d.red[(size_t)(6 * ci + r / 6) * n + 6 * cj + r % 6], v);
    }
    for (int e = tid; e < 6 * kWinCams; e += 1024) {
        const double v = srhs[e];
        if (v != 0.0 && cw + e / 6 < d.n_cam) atomicAdd(&d.red[(size_t)n * n + 6 * cw + e], v);
    }
}

// Free intrinsics: the block row of the reduced system that belongs to fx, cx, fy, cy (block index n_real_cam).
// One thread per point p.  With G_i the intrinsics columns of observation i (2 x 4, two non-zeros per row),
//   Wk_p = sum_i G_i'E_i (4x3), Yk = Wk_p M^-1:
//   S[k][k]   -= Yk Wk_p'            rhs[k] -= Wk_p M^-1 E'r
//   S[k][c_j] -= Yk W_j'  + G_j'F_j  for every observation j of p  (the second term is the F'F cross block)
// The camera-indexed part goes through an LDS copy of the block row (4 x 6 n_real_cam) when it fits, the 14 sums
// every point shares through a workgroup reduction; both are then added into d.red.  Runs after ba_schur.
__global__ __launch_bounds__(256) void ba_schur_calib_kernel(BADev d, int use_lds)
{
    extern __shared__ __attribute__((aligned(16))) double sl[];   // [8] reduction scratch, then use_lds: [4 * 6 n_real_cam]
    double *red = sl;
    double *rowblk = sl + 8;
    const int tid = threadIdx.x;
    const int nr = d.n_real_cam, n = 6 * d.n_cam, kap = 6 * nr, roww = 6 * nr;
    if (use_lds) {
        for (int e = tid; e < 4 * roww; e += 256) rowblk[e] = 0.0;
        __syncthreads();
    }
    double acc[14];
#pragma unroll
    for (int q = 0; q < 14; ++q) acc[q] = 0.0;
    const int p = blockIdx.x * 256 + tid;
    if (p < d.n_pt) {
        const int b = d.pt_start[p], e = d.pt_start[p + 1];
        if (e > b) {
            const size_t no = d.n_obs;
            double Wk[4][3];
#pragma unroll
            for (int a = 0; a < 4; ++a) Wk[a][0] = Wk[a][1] = Wk[a][2] = 0.0;
            for (int k = b; k < e; ++k) {
                const double k0 = d.Jk[k], k1 = d.Jk[no + k], k2 = d.Jk[2 * no + k], k3 = d.Jk[3 * no + k];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const double e0 = d.Jp[m * no + k], e1 = d.Jp[(3 + m) * no + k];
                    Wk[0][m] += k0 * e0; Wk[1][m] += k1 * e0; Wk[2][m] += k2 * e1; Wk[3][m] += k3 * e1;
                }
            }
            const double *Mi = d.Minv + 6 * (size_t)p;
            const double M[9] = {Mi[0], Mi[1], Mi[2], Mi[1], Mi[3], Mi[4], Mi[2], Mi[4], Mi[5]};
            const double ag[3] = {d.Aig[3 * (size_t)p], d.Aig[3 * (size_t)p + 1], d.Aig[3 * (size_t)p + 2]};
            double Yk[4][3];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int m = 0; m < 3; ++m) Yk[a][m] = Wk[a][0] * M[m] + Wk[a][1] * M[3 + m] + Wk[a][2] * M[6 + m];
                acc[10 + a] = -(Wk[a][0] * ag[0] + Wk[a][1] * ag[1] + Wk[a][2] * ag[2]);
            }
            {
                int q = 0;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b2 = 0; b2 <= a; ++b2) acc[q++] = -(Yk[a][0] * Wk[b2][0] + Yk[a][1] * Wk[b2][1] + Yk[a][2] * Wk[b2][2]);
            }
            for (int k = b; k < e; ++k) {
                const int c = d.obs_cam[k];
                const double kk[4] = {d.Jk[k], d.Jk[no + k], d.Jk[2 * no + k], d.Jk[3 * no + k]};
                double Ej[6];
#pragma unroll
                for (int m = 0; m < 6; ++m) Ej[m] = d.Jp[m * no + k];
#pragma unroll
                for (int c2 = 0; c2 < 6; ++c2) {
                    const double f0 = d.Jc[c2 * no + k], f1 = d.Jc[(6 + c2) * no + k];
                    const double w0 = f0 * Ej[0] + f1 * Ej[3], w1 = f0 * Ej[1] + f1 * Ej[4], w2 = f0 * Ej[2] + f1 * Ej[5];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const double v = (a < 2 ? kk[a] * f0 : kk[a] * f1) - (Yk[a][0] * w0 + Yk[a][1] * w1 + Yk[a][2] * w2);
                        if (use_lds) atomicAdd(&rowblk[a * roww + 6 * c + c2], v);
                        else atomicAdd(&d.red[(size_t)(kap + a) * n + 6 * c + c2], v);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 14; ++q) {
        const double v = block_sum(acc[q], red);
        if (tid != 0 || v == 0.0) continue;
        if (q >= 10) { atomicAdd(&d.red[(size_t)n * n + kap + (q - 10)], v); continue; }
        int a = 0, rem = q;
        while (rem > a) { rem -= a + 1; ++a; }
        atomicAdd(&d.red[(size_t)(kap + a) * n + kap + rem], v);
    }
    if (use_lds) {
        __syncthreads();
        for (int e = tid; e < 4 * roww; e += 256) {
            const double v = rowblk[e];
            if (v != 0.0) atomicAdd(&d.red[(size_t)(kap + e / roww) * n + (e % roww)], v);
        }
    }
}

// Sum the per-workgroup slabs (fixed order) and scatter into red = S_schur (n x n) | rhs_corr (n).
__global__ __launch_bounds__(256) void ba_schur_reduce_kernel(BADev d, const double *__restrict__ slabs, int slab_doubles, int n_slabs)
{
    __shared__ double lds[256];
    const int e = blockIdx.x * kRedEnt + (threadIdx.x % kRedEnt);
    const double s = slab_sum(slabs, slab_doubles, n_slabs, e, lds);
    if (threadIdx.x >= kRedEnt || e >= slab_doubles) return;
    const int n = 6 * d.n_cam;
    const int nblk = d.n_cam * (d.n_cam + 1) / 2;
    if (e >= nblk * 36) { d.red[(size_t)n * n + (e - nblk * 36)] = s; return; }
    const int blk = e / 36, r = e % 36;
    int ci = (int)((sqrt(8.0 * (double)blk + 1.0) - 1.0) * 0.5);
    while ((ci + 1) * (ci + 2) / 2 <= blk) ++ci;
    while (ci * (ci + 1) / 2 > blk) --ci;
    const int cj = blk - ci * (ci + 1) / 2;
    d.red[(size_t)(6 * ci + r / 6) * n + 6 * cj + r % 6] = s;
}

// ---------------------------------------------------------------------------------------------
// Dense solve of the reduced camera system (DENSE_SCHUR's Cholesky):
//   (F'F + D_c^2 + S_schur) y = F'r + rhs_corr
// One workgroup, right-looking blocked Cholesky on a packed lower matrix in LDS with the right-hand side carried
// as row n (so the forward substitution comes for free), then the backward substitution.
constexpr int kCholThreads = 1024;
constexpr int kCholNB = 8;  // panel width

// Per panel of kCholNB columns (3 barriers):
//   B1  wave 0 factors the nb x nb diagonal block in registers (lane r = block row r, __shfl)
//   B2  every row below (and the rhs row) solves against the block's transpose
//   C   all waves apply the rank-nb update to the trailing matrix in 4 x 4 register tiles: per panel column 8 LDS loads
//       feed 16 independent FMAs (the left-looking form this replaces issued 2 dependent loads per FMA and was
//       LDS-latency bound: 161 us at n = 150)
// The backward substitution is blocked the same way.
template <bool LDS_STORE>
__global__ __launch_bounds__(kCholThreads) void ba_chol_solve_kernel(BADev d, double radius, double min_diag, double max_diag)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = 6 * d.n_cam;
    const size_t tot = (size_t)(n + 1) * (n + 2) / 2;
    double *L = LDS_STORE ? smem : d.chol;            // packed lower, (n+1) rows; row n = rhs
    double *rd = LDS_STORE ? (smem + tot) : smem;     // [n] reciprocal diagonal of L
    // failure flag kept inside the dynamic region (a static __shared__ object in front of it would
    // shift the f64 array off its 8-byte alignment)
    volatile double *failp = rd + n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *failp = 0.0;
#ifdef ESFM_CHOL_PROFILE
    double prof[6] = {0, 0, 0, 0, 0, 0};
    long long tprev = wall_clock64();
#define CHOL_MARK(slot) do { const long long tn__ = wall_clock64(); prof[slot] += (double)(tn__ - tprev); tprev = tn__; } while (0)
#else
#define CHOL_MARK(slot) do { } while (0)
#endif
    const double *S = d.red;
    const double *rc = d.red + (size_t)n * n;
    const double *FtF = d.camacc;
    const double *Ftr = d.camacc + 36 * (size_t)d.n_cam;
    // assemble: L = F'F + D_c^2 + S_schur (lower), row n = F'r + rhs_corr
    for (size_t e = tid; e < tot; e += kCholThreads) {
        int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
        while ((size_t)(i + 1) * (i + 2) / 2 <= e) ++i;
        while ((size_t)i * (i + 1) / 2 > e) --i;
        const int k = (int)(e - (size_t)i * (i + 1) / 2);
        double v;
        if (i < n) {
            v = S[(size_t)i * n + k];
            if (i / 6 == k / 6) {
                const int c = i / 6;
                v += FtF[36 * (size_t)c + 6 * (i % 6) + (k % 6)];
                if (i == k) v += fmin(fmax(FtF[36 * (size_t)c + 7 * (i % 6)], min_diag), max_diag) / radius;
            }
        } else {
            v = (k < n) ? (Ftr[k] + rc[k]) : 0.0;
        }
        L[e] = v;
    }
    __syncthreads();
    CHOL_MARK(0);
    auto row = [&](int i) -> double * { return L + (size_t)i * (i + 1) / 2; };

    for (int j0 = 0; j0 < n; j0 += kCholNB) {
        const int nb = min(kCholNB, n - j0);
        // B1: wave 0 factors the nb x nb diagonal block in registers (lane r = block row r)
        if (wave == 0) {
            const int r = lane;
            double a[kCholNB];
#pragma unroll
            for (int c = 0; c < kCholNB; ++c) a[c] = (r < nb && c <= r) ? row(j0 + r)[j0 + c] : 0.0;
#pragma unroll
            for (int c = 0; c < kCholNB; ++c) {
                if (c < nb) {
                    const double piv = readlane_f64(a[c], c);
                    if (!(piv > 0.0) || !isfinite(piv)) { if (lane == 0) *failp = 1.0; }
                    const double rinv = rsqrt(piv > 0.0 ? piv : 1.0);
                    a[c] = (r == c) ? piv * rinv : a[c] * rinv;
                    if (lane == c) rd[j0 + c] = rinv;
#pragma unroll
                    for (int c2 = c + 1; c2 < kCholNB; ++c2) {
                        const double l2 = readlane_f64(a[c], c2);   // L[j0+c2][j0+c]
                        if (r >= c2) a[c2] -= a[c] * l2;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < kCholNB; ++c)
                if (r < nb && c <= r) row(j0 + r)[j0 + c] = a[c];
        }
        __syncthreads();
        CHOL_MARK(1);
        // B2: every row below the block (and the rhs row) solves against the block's transpose
        for (int i = j0 + nb + tid; i <= n; i += kCholThreads) {
            double *Li = row(i);
            double x[kCholNB];
#pragma unroll
            for (int c = 0; c < kCholNB; ++c) x[c] = c < nb ? Li[j0 + c] : 0.0;
#pragma unroll
            for (int c = 0; c < kCholNB; ++c) {
                if (c < nb) {
                    const double *Lc = row(j0 + c) + j0;
                    double v = x[c];
#pragma unroll
                    for (int c1 = 0; c1 < c; ++c1) v -= x[c1] * Lc[c1];
                    x[c] = v * rd[j0 + c];
                    Li[j0 + c] = x[c];
                }
            }
        }
        __syncthreads();
        CHOL_MARK(2);
        // C: trailing update A[i][j] -= sum_c L[i][j0+c] L[j][j0+c] over j0+nb <= j <= i <= n, j < n, in 4 x 4 tiles
        {
            const int base = j0 + nb;
            const int R = (n + 1 - base + 3) / 4;           // tile rows (the rhs row n included)
            const int ntile = R * (R + 1) / 2;
            for (int t = tid; t < ntile; t += kCholThreads) {
                int ta = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while ((ta + 1) * (ta + 2) / 2 <= t) ++ta;
                while (ta * (ta + 1) / 2 > t) --ta;
                const int tb = t - ta * (ta + 1) / 2;
                const int i0 = base + 4 * ta, c0 = base + 4 * tb;
                double acc[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.0;
                const double *ri[4], *rj[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ri[r] = row(min(i0 + r, n)) + j0;        // clamped rows are masked at the store
                    rj[r] = row(min(c0 + r, n)) + j0;
                }
#pragma unroll
                for (int c = 0; c < kCholNB; ++c) {
                    if (c < nb) {
                        double li[4], lj[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { li[r] = ri[r][c]; lj[r] = rj[r][c]; }
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[4 * r + q] += li[r] * lj[q];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + r;
                    if (i > n) continue;
                    double *Ai = row(i);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int j = c0 + q;
                        if (j <= i && j < n) Ai[j] -= acc[4 * r + q];
                    }
                }
            }
        }
        __syncthreads();
        CHOL_MARK(3);
    }
    // backward substitution L' y = z (z = row n), panels from the bottom
    double *z = row(n);
    const int last = ((n - 1) / kCholNB) * kCholNB;
    for (int j0 = last; j0 >= 0; j0 -= kCholNB) {
        const int nb = min(kCholNB, n - j0);
        if (wave == 0) {
            // the nb x nb block in registers: lane k holds z[j0+k] and column k of the block (L[j0+i][j0+k], i > k)
            const int kk = lane;
            double zk = kk < nb ? z[j0 + kk] : 0.0;
            double col[kCholNB];
#pragma unroll
            for (int i = 0; i < kCholNB; ++i) col[i] = (i < nb && kk < i) ? row(j0 + i)[j0 + kk] : 0.0;
            const double rdk = kk < nb ? rd[j0 + kk] : 0.0;
#pragma unroll
            for (int i = kCholNB - 1; i >= 0; --i) {
                if (i < nb) {
                    const double yi = readlane_f64(zk * rdk, i);
                    zk = (kk == i) ? yi : zk - col[i] * yi;
                }
            }
            if (kk < nb) z[j0 + kk] = zk;
            if (!LDS_STORE) __threadfence_block();
        }
        __syncthreads();
        for (int k = tid; k < j0; k += kCholThreads) {
            double s = 0.0;
            for (int i = j0; i < j0 + nb; ++i) s += row(i)[k] * z[i];
            z[k] -= s;
        }
        __syncthreads();
    }
    CHOL_MARK(4);
    const bool fail = *failp != 0.0;
    for (int i = tid; i < n; i += kCholThreads) d.y_c[i] = fail ? 0.0 : z[i];
    if (tid == 0 && fail) d.scal[SC_CHOL_FAIL] = 1.0;
#ifdef ESFM_CHOL_PROFILE
    CHOL_MARK(5);
    if (tid == 0) for (int q = 0; q < 6; ++q) atomicAdd(&d.chol[q], prof[q]);
#endif
#undef CHOL_MARK
}

// ---------------------------------------------------------------------------------------------
// Candidate cameras: x + (-y) .* scaling for cameras that have observations; step/candidate norms.
__global__ __launch_bounds__(256) void ba_camera_step_kernel(BADev d)
{
    __shared__ double red[8];
    double ssq = 0.0, csq = 0.0, dmax = 0.0;
    for (int i = threadIdx.x; i < 6 * d.n_cam; i += 256) {
        const bool active = d.cam_nobs[i / 6] > 0.0;
        const double x = d.x_c[i];
        const double dl = active ? (-d.y_c[i]) * d.scale_c[i] : 0.0;
        double cnd = active ? x + dl : x;
        if (d.constrained) {
            // ParameterBlock::Plus projects onto the box (lower bound first) [upstream parameter_block.h]
            if (active) cnd = fmin(fmax(cnd, d.lo_c[i]), d.up_c[i]);
            d.delta_c[i] = dl;
            dmax = fmax(dmax, fabs(dl));
        }
        d.cand_c[i] = cnd;
        if (active) { const double df = x - cnd; ssq += df * df; csq += cnd * cnd; }
    }
    const double a = block_sum(ssq, red);
    const double b = block_sum(csq, red);
    if (threadIdx.x == 0) { d.scal[SC_STEP_SQ_CAM] = a; d.scal[SC_CAND_SQ_CAM] = b; }
    if (d.constrained) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o));
        if ((threadIdx.x & 63) == 0 && dmax > 0.0) atomic_max_nonneg(&d.scal[SC_DMAX], dmax);
    }
}

// Back-substitution, point-parallel variant (one thread per point; fewer, fatter threads: faster below ~1M
// observations, where the observation-parallel passes are launch/atomic bound): y_p = M^-1 (E'r - sum_i E_i'F_i y_c), step = -y, candidate point,
// and this point's share of model_cost_change = -sum (J s).(r + J s / 2)  (trust_region_minimizer.cc).
__global__ __launch_bounds__(64) void ba_backsub_point_kernel(BADev d)
{
    __shared__ double red[8];
    const int p = blockIdx.x * 64 + threadIdx.x;
    double mc = 0.0, ssq = 0.0, csq = 0.0, gd = 0.0, dmax = 0.0;
    double yk[4] = {0.0, 0.0, 0.0, 0.0};
    if (d.has_calib) {
#pragma unroll
        for (int a = 0; a < 4; ++a) yk[a] = d.y_c[6 * d.n_real_cam + a];
    }
    if (p < d.n_pt) {
        const int b = d.pt_start[p], e = d.pt_start[p + 1];
        const size_t n = d.n_obs;
        double xp[3] = {d.x_p[3 * (size_t)p], d.x_p[3 * (size_t)p + 1], d.x_p[3 * (size_t)p + 2]};
        if (e > b) {
            double g[3] = {d.Etr[3 * (size_t)p], d.Etr[3 * (size_t)p + 1], d.Etr[3 * (size_t)p + 2]};
            // F_i y_c of the first kKeep observations stays in registers: the second pass below needs its negative
            // (-(a + b) == (-a) + (-b) exactly), which saves re-reading the camera Jacobian
            constexpr int kKeep = 8;
            double keep0[kKeep], keep1[kKeep];
            for (int k = b; k < e; ++k) {
                const int c = d.obs_cam[k];
                double f0 = 0.0, f1 = 0.0;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const double yc = d.y_c[6 * c + a];
                    f0 += d.Jc[a * n + k] * yc; f1 += d.Jc[(6 + a) * n + k] * yc;
                }
                if (d.has_calib) {
                    f0 += d.Jk[k] * yk[0] + d.Jk[n + k] * yk[1];
                    f1 += d.Jk[2 * n + k] * yk[2] + d.Jk[3 * n + k] * yk[3];
                }
#pragma unroll
                for (int q = 0; q < kKeep; ++q) if (k - b == q) { keep0[q] = f0; keep1[q] = f1; }
#pragma unroll
                for (int a = 0; a < 3; ++a) g[a] -= d.Jp[a * n + k] * f0 + d.Jp[(3 + a) * n + k] * f1;
            }
            const double *Mi = d.Minv + 6 * (size_t)p;
            const double sp[3] = {-(Mi[0] * g[0] + Mi[1] * g[1] + Mi[2] * g[2]),
                                  -(Mi[1] * g[0] + Mi[3] * g[1] + Mi[4] * g[2]),
                                  -(Mi[2] * g[0] + Mi[4] * g[1] + Mi[5] * g[2])};
            for (int k = b; k < e; ++k) {
                double m0 = 0.0, m1 = 0.0;
                if (k - b < kKeep) {
#pragma unroll
                    for (int q = 0; q < kKeep; ++q) if (k - b == q) { m0 = -keep0[q]; m1 = -keep1[q]; }
                } else {
                    const int c = d.obs_cam[k];
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        const double sc = -d.y_c[6 * c + a];
                        m0 += d.Jc[a * n + k] * sc; m1 += d.Jc[(6 + a) * n + k] * sc;
                    }
                    if (d.has_calib) {
                        m0 -= d.Jk[k] * yk[0] + d.Jk[n + k] * yk[1];
                        m1 -= d.Jk[2 * n + k] * yk[2] + d.Jk[3 * n + k] * yk[3];
                    }
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) { m0 += d.Jp[a * n + k] * sp[a]; m1 += d.Jp[(3 + a) * n + k] * sp[a]; }
                const double r0 = d.res[k], r1 = d.res[n + k];
                mc -= m0 * (r0 + m0 / 2.0) + m1 * (r1 + m1 / 2.0);
                gd += m0 * r0 + m1 * r1;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double dl = sp[a] * d.scale_p[3 * (size_t)p + a];
                const double cnd = xp[a] + dl;
                const double df = xp[a] - cnd;
                ssq += df * df; csq += cnd * cnd;
                d.cand_p[3 * (size_t)p + a] = cnd;
                if (d.constrained) { d.delta_p[3 * (size_t)p + a] = dl; dmax = fmax(dmax, fabs(dl)); }
            }
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                d.cand_p[3 * (size_t)p + a] = xp[a];
                if (d.constrained) d.delta_p[3 * (size_t)p + a] = 0.0;
            }
        }
    }
    const double s0 = block_sum(mc, red);
    const double s1 = block_sum(ssq, red);
    const double s2 = block_sum(csq, red);
    if (threadIdx.x == 0) {
        atomicAdd(&d.scal[SC_MODEL_CHANGE], s0);
        atomicAdd(&d.scal[SC_STEP_SQ_PT], s1);
        atomicAdd(&d.scal[SC_CAND_SQ_PT], s2);
    }
    if (d.constrained) {
        const double s3 = block_sum(gd, red);
        if (threadIdx.x == 0) atomicAdd(&d.scal[SC_GDOTD], s3);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o));
        if ((threadIdx.x & 63) == 0 && dmax > 0.0) atomic_max_nonneg(&d.scal[SC_DMAX], dmax);
    }
}

// Back-substitution, observation-parallel so that every J access is a coalesced SoA stream:
//   pass 1 (ba_backsub_accum_kernel)  gE[p] += E_k' (F_k y_c)           3 f64 atomics per observation
//   pass 2 (ba_backsub_apply_kernel)  s_p = -M^-1 (E'r - gE[p]) (recomputed per observation, 12 cached loads),
//                                     model_cost_change -= (J s).(r + J s / 2)  (trust_region_minimizer.cc),
//                                     and the first observation of each point writes the candidate point.
__global__ __launch_bounds__(256) void ba_backsub_accum_kernel(BADev d)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= d.n_obs) return;
    const size_t n = d.n_obs;
    const int c = d.obs_cam[k], p = d.obs_pt[k];
    double f0 = 0.0, f1 = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const double yc = d.y_c[6 * c + a];
        f0 += d.Jc[a * n + k] * yc; f1 += d.Jc[(6 + a) * n + k] * yc;
    }
    if (d.has_calib) {
        const double *yk = d.y_c + 6 * d.n_real_cam;
        f0 += d.Jk[k] * yk[0] + d.Jk[n + k] * yk[1];
        f1 += d.Jk[2 * n + k] * yk[2] + d.Jk[3 * n + k] * yk[3];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) atomicAdd(&d.gE[3 * (size_t)p + a], d.Jp[a * n + k] * f0 + d.Jp[(3 + a) * n + k] * f1);
}

__global__ __launch_bounds__(256) void ba_backsub_apply_kernel(BADev d)
{
    __shared__ double red[8];
    const int k = blockIdx.x * 256 + threadIdx.x;
    double mc = 0.0, ssq = 0.0, csq = 0.0, gd = 0.0, dmax = 0.0;
    if (k < d.n_obs) {
        const size_t n = d.n_obs;
        const int c = d.obs_cam[k], p = d.obs_pt[k];
        const double g0 = d.Etr[3 * (size_t)p] - d.gE[3 * (size_t)p], g1 = d.Etr[3 * (size_t)p + 1] - d.gE[3 * (size_t)p + 1],
                     g2 = d.Etr[3 * (size_t)p + 2] - d.gE[3 * (size_t)p + 2];
        const double *Mi = d.Minv + 6 * (size_t)p;
        const double sp[3] = {-(Mi[0] * g0 + Mi[1] * g1 + Mi[2] * g2), -(Mi[1] * g0 + Mi[3] * g1 + Mi[4] * g2),
                              -(Mi[2] * g0 + Mi[4] * g1 + Mi[5] * g2)};
        double m0 = 0.0, m1 = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double sc = -d.y_c[6 * c + a];
            m0 += d.Jc[a * n + k] * sc; m1 += d.Jc[(6 + a) * n + k] * sc;
        }
        if (d.has_calib) {
            const double *yk = d.y_c + 6 * d.n_real_cam;
            m0 -= d.Jk[k] * yk[0] + d.Jk[n + k] * yk[1];
            m1 -= d.Jk[2 * n + k] * yk[2] + d.Jk[3 * n + k] * yk[3];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { m0 += d.Jp[a * n + k] * sp[a]; m1 += d.Jp[(3 + a) * n + k] * sp[a]; }
        const double r0 = d.res[k], r1 = d.res[n + k];
        mc = -(m0 * (r0 + m0 / 2.0) + m1 * (r1 + m1 / 2.0));
        gd = m0 * r0 + m1 * r1;
        if (k == d.pt_start[p]) {   // one writer per point
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double xp = d.x_p[3 * (size_t)p + a];
                const double dl = sp[a] * d.scale_p[3 * (size_t)p + a];
                const double cnd = xp + dl;
                const double df = xp - cnd;
                ssq += df * df; csq += cnd * cnd;
                d.cand_p[3 * (size_t)p + a] = cnd;
                if (d.constrained) { d.delta_p[3 * (size_t)p + a] = dl; dmax = fmax(dmax, fabs(dl)); }
            }
        }
    }
    const double s0 = block_sum(mc, red);
    const double s1 = block_sum(ssq, red);
    const double s2 = block_sum(csq, red);
    if (threadIdx.x == 0) {
        atomicAdd(&d.scal[SC_MODEL_CHANGE], s0);
        atomicAdd(&d.scal[SC_STEP_SQ_PT], s1);
        atomicAdd(&d.scal[SC_CAND_SQ_PT], s2);
    }
    if (d.constrained) {
        const double s3 = block_sum(gd, red);
        if (threadIdx.x == 0) atomicAdd(&d.scal[SC_GDOTD], s3);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o));
        if ((threadIdx.x & 63) == 0 && dmax > 0.0) atomic_max_nonneg(&d.scal[SC_DMAX], dmax);
    }
}

// Robustified cost 1/2 sum rho(|r|^2) of (cams, pts) over this rank's observations.
// SLOPE: also d/dt cost(Plus(x, t delta)) at this point = gradient . delta = sum rho' r.(J delta), the derivative
// the Armijo search's cubic interpolation uses [upstream line_search.cc LineSearchFunction::Evaluate].
template <bool SLOPE>
__global__ __launch_bounds__(256) void ba_cost_kernel(BADev d, const double *__restrict__ cams, const double *__restrict__ pts,
                                                      double cauchy_a, int slot, int bad_slot)
{
    __shared__ double red[8];
    double cost = 0.0, bad = 0.0, slope = 0.0;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < d.n_obs; k += gridDim.x * 256) {
        const int c = d.obs_cam[k], p = d.obs_pt[k];
        const float2 uv = d.obs_uv[k];
        double in4[4];
        load_intrinsics(d, cams, c, in4);
        double cam[6], X[3];
#pragma unroll
        for (int i = 0; i < 6; ++i) cam[i] = cams[6 * (size_t)c + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) X[i] = pts[3 * (size_t)p + i];
        double r0, r1, m0 = 0.0, m1 = 0.0;
        if (SLOPE) {
            double Jc[12], Jp[6], xn, yn;
            reproject_jac(cam, X, in4, uv, r0, r1, Jc, Jp, xn, yn);
#pragma unroll
            for (int a = 0; a < 6; ++a) { const double dl = d.delta_c[6 * (size_t)c + a]; m0 += Jc[a] * dl; m1 += Jc[6 + a] * dl; }
#pragma unroll
            for (int a = 0; a < 3; ++a) { const double dl = d.delta_p[3 * (size_t)p + a]; m0 += Jp[a] * dl; m1 += Jp[3 + a] * dl; }
            if (d.has_calib) {
                const double *dk = d.delta_c + 6 * (size_t)d.n_real_cam;
                m0 -= xn * dk[0] + dk[1];
                m1 -= yn * dk[2] + dk[3];
            }
        } else {
            double pt[3];
            transform_point<false>(cam, X, pt, nullptr, nullptr);
            const double x = pt[0] / pt[2], y = pt[1] / pt[2];
            r0 = (double)uv.x - (x * in4[0] + in4[1]);
            r1 = (double)uv.y - (y * in4[2] + in4[3]);
        }
        const double s = r0 * r0 + r1 * r1;
        if (!isfinite(s)) { bad += 1.0; continue; }
        double rho0, rho1;
        loss_eval(cauchy_a, s, rho0, rho1);
        cost += 0.5 * rho0;
        if (SLOPE) slope += rho1 * (r0 * m0 + r1 * m1);
    }
    const double cs = block_sum(cost, red);
    const double bs = block_sum(bad, red);
    if (threadIdx.x == 0) { atomicAdd(&d.scal[slot], cs); if (bs > 0.0) atomicAdd(&d.scal[bad_slot], bs); }
    if (SLOPE) {
        const double ss = block_sum(slope, red);
        if (threadIdx.x == 0) atomicAdd(&d.scal[SC_LS_GRAD], ss);
    }
}

// candidate = Plus(x, t delta): cameras (projected onto the box) by workgroup 0, points by all; step / candidate norms.
__global__ __launch_bounds__(256) void ba_take_step_kernel(BADev d, double t)
{
    __shared__ double red[8];
    double ssq = 0.0, csq = 0.0, ssc = 0.0, csc = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * d.n_pt; i += gridDim.x * 256) {
        const int p = i / 3;
        const double x = d.x_p[i];
        if (d.pt_start[p + 1] > d.pt_start[p]) {
            const double cnd = x + t * d.delta_p[i];
            const double df = x - cnd;
            ssq += df * df; csq += cnd * cnd;
            d.cand_p[i] = cnd;
        } else d.cand_p[i] = x;
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 6 * d.n_cam; i += 256) {
            const double x = d.x_c[i];
            if (d.cam_nobs[i / 6] > 0.0) {
                const double cnd = fmin(fmax(x + t * d.delta_c[i], d.lo_c[i]), d.up_c[i]);
                const double df = x - cnd;
                ssc += df * df; csc += cnd * cnd;
                d.cand_c[i] = cnd;
            } else d.cand_c[i] = x;
        }
    const double a = block_sum(ssq, red);
    const double b = block_sum(csq, red);
    const double c = block_sum(ssc, red);
    const double e = block_sum(csc, red);
    if (threadIdx.x == 0) {
        atomicAdd(&d.scal[SC_STEP_SQ_PT], a); atomicAdd(&d.scal[SC_CAND_SQ_PT], b);
        if (blockIdx.x == 0) { d.scal[SC_STEP_SQ_CAM] = c; d.scal[SC_CAND_SQ_CAM] = e; }
    }
}

// x_c <- projection onto the box (TrustRegionMinimizer::IterationZero: Plus(x, 0))
__global__ void ba_project_cameras_kernel(BADev d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * d.n_cam && d.cam_nobs[i / 6] > 0.0) d.x_c[i] = fmin(fmax(d.x_c[i], d.lo_c[i]), d.up_c[i]);
}

// |x|^2 over the parameter blocks that take part in the problem (Ceres drops unused blocks).
__global__ __launch_bounds__(256) void ba_param_sqnorm_kernel(BADev d)
{
    __shared__ double red[8];
    double sp = 0.0, sc = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * d.n_pt; i += gridDim.x * 256) {
        const int p = i / 3;
        if (d.pt_start[p + 1] > d.pt_start[p]) sp += d.x_p[i] * d.x_p[i];
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 6 * d.n_cam; i += 256)
            if (d.cam_nobs[i / 6] > 0.0) sc += d.x_c[i] * d.x_c[i];
    const double a = block_sum(sp, red);
    const double b = block_sum(sc, red);
    if (threadIdx.x == 0) { atomicAdd(&d.scal[SC_XNORM_SQ_PT], a); if (blockIdx.x == 0) d.scal[SC_XNORM_SQ_CAM] = b; }
}

// Multi-GPU merge of the point blocks: each rank owns the points it has observations for.
//   to_delta: x_p <- (owned ? x_p - x0_p : 0)   (then SUM all-reduce)
//   else    : x_p <- x0_p + x_p
__global__ void ba_points_delta_kernel(BADev d, int to_delta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * d.n_pt) return;
    const int p = i / 3;
    if (to_delta) d.x_p[i] = (d.pt_start[p + 1] > d.pt_start[p]) ? d.x_p[i] - d.x0_p[i] : 0.0;
    else d.x_p[i] = d.x0_p[i] + d.x_p[i];
}

// Multi-GPU exchange buffer of the reduced system ("packed-upper S + rhs"): only the blocks the Schur kernels
// write (row camera >= column camera) travel, row by row -- row i of block row bi holds 6 (bi + 1) doubles at
// 36 bi (bi + 1) / 2 + (i % 6) 6 (bi + 1) -- followed by the n right-hand-side entries: 36 Nc (Nc + 1) / 2 + 6 Nc doubles
// instead of (6 Nc)^2 + 6 Nc.  One workgroup per row (coalesced both ways), the last one moves the right-hand side.
__global__ __launch_bounds__(256) void ba_red_pack_kernel(BADev d, double *__restrict__ packed, int unpack)
{
    const int n = 6 * d.n_cam;
    const int i = blockIdx.x;
    const double *src; double *dst; int cnt;
    if (i < n) {
        const int bi = i / 6, r = i - 6 * bi;
        cnt = 6 * (bi + 1);
        double *full = d.red + (size_t)i * n;
        double *pk = packed + (size_t)18 * bi * (bi + 1) + (size_t)r * cnt;
        src = unpack ? pk : full; dst = unpack ? full : pk;
    } else {
        cnt = n;
        double *full = d.red + (size_t)n * n;
        double *pk = packed + (size_t)18 * d.n_cam * (d.n_cam + 1);
        src = unpack ? pk : full; dst = unpack ? full : pk;
    }
    for (int k = threadIdx.x; k < cnt; k += 256) dst[k] = src[k];
}

// ---------------------------------------------------------------------------------------------
static inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
#define LAUNCH_CHECK() ESFM_HIP_TRY(hipGetLastError())

int ba_red_pack(hipStream_t st, const BADev &d, double *packed, bool unpack)
{
    if (d.n_cam <= 0) return ESFM_OK;
    hipLaunchKernelGGL(ba_red_pack_kernel, dim3(6 * d.n_cam + 1), dim3(256), 0, st, d, packed, unpack ? 1 : 0);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_linearize(hipStream_t st, const BADev &d, int num_cu, double cauchy_a, bool use_scaling, esfm_ctx *timing_ctx)
{
    if (d.n_obs <= 0) { ESFM_HIP_TRY(hipMemsetAsync(d.camacc, 0, sizeof(double) * ba_camacc_doubles(d.n_cam), st)); return ESFM_OK; }
    const size_t priv_bytes = sizeof(double) * (8 + (size_t)d.n_cam * 27);
    const int grid = std::min(div_up(d.n_obs, kLinThreads), std::max(1, num_cu) * 2);
    const bool priv = priv_bytes <= 150 * 1024 && d.lin_slabs && (size_t)grid * d.n_cam * 27 <= d.lin_slab_cap;
    if (priv) {
        auto kern = d.has_calib ? &ba_linearize_kernel<true, true> : &ba_linearize_kernel<true, false>;
        ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)priv_bytes));
        {
            KernelTimer tm(timing_ctx, ESFM_K_BA_LINEARIZE);   // the Jacobian sweep alone (not the slab reduction)
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kLinThreads), priv_bytes, st, d, cauchy_a, use_scaling ? 1 : 0, d.lin_slabs);
        }
        LAUNCH_CHECK();
        hipLaunchKernelGGL(ba_camacc_reduce_kernel, dim3(div_up(d.n_cam * 27, kRedEnt)), dim3(256), 0, st, d, d.lin_slabs, grid);
    } else {
        ESFM_HIP_TRY(hipMemsetAsync(d.camacc, 0, sizeof(double) * ba_camacc_doubles(d.n_cam), st));
        KernelTimer tm(timing_ctx, ESFM_K_BA_LINEARIZE);
        auto kern = d.has_calib ? &ba_linearize_kernel<false, true> : &ba_linearize_kernel<false, false>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kLinThreads), sizeof(double) * 8, st, d, cauchy_a, use_scaling ? 1 : 0, (double *)nullptr);
    }
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_point_prep(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag, bool fresh)
{
    if (d.n_pt <= 0) return ESFM_OK;
    hipLaunchKernelGGL(ba_point_prep_kernel, dim3(div_up(d.n_pt, 64)), dim3(64), 0, st, d, radius, min_diag, max_diag, fresh ? 1 : 0);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_jacobi_scaling(hipStream_t st, const BADev &d)
{
    const int m = std::max(3 * d.n_pt, 6 * d.n_cam);
    if (m <= 0) return ESFM_OK;
    hipLaunchKernelGGL(ba_jacobi_scaling_kernel, dim3(div_up(m, 256)), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_camera_gradient(hipStream_t st, const BADev &d)
{
    if (d.n_cam <= 0) return ESFM_OK;
    hipLaunchKernelGGL(ba_camera_gradient_kernel, dim3(div_up(6 * d.n_cam, 256)), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_schur(hipStream_t st, const BADev &d, int num_cu, double *slabs, size_t slab_capacity_doubles)
{
    ESFM_HIP_TRY(hipMemsetAsync(d.red, 0, sizeof(double) * ba_red_doubles(d.n_cam), st));
    if (d.n_obs <= 0) return ESFM_OK;
    const int nblk = d.n_cam * (d.n_cam + 1) / 2;
    const int slab_doubles = nblk * 36 + 6 * d.n_cam;
    const size_t lds_bytes = sizeof(double) * (size_t)slab_doubles;
    const int n_slabs = std::max(1, std::min(num_cu, div_up(d.n_obs, 1024)));
    if (lds_bytes <= 156 * 1024 && slabs && (size_t)n_slabs * slab_doubles <= slab_capacity_doubles) {
        ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_schur_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(ba_schur_lds_kernel, dim3(n_slabs), dim3(1024), lds_bytes, st, d, slabs, slab_doubles);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(ba_schur_reduce_kernel, dim3(div_up(slab_doubles, kRedEnt)), dim3(256), 0, st, d, slabs, slab_doubles, n_slabs);
        LAUNCH_CHECK();
        return ESFM_OK;
    }
    if (d.n_chunks > 0 && d.slot_obs) {
        constexpr size_t win_bytes = sizeof(double) * (kWinBlocks * 36 + 6 * kWinCams);
        ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_schur_window_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_bytes));
        hipLaunchKernelGGL(ba_schur_window_kernel, dim3(d.n_chunks), dim3(1024), win_bytes, st, d);
        LAUNCH_CHECK();
        return ESFM_OK;
    }
    hipLaunchKernelGGL(ba_schur_kernel, dim3(div_up(d.n_obs, 256)), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_solve_reduced(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag)
{
    if (d.n_cam <= 0) return ESFM_OK;
    const int n = 6 * d.n_cam;
    const size_t bytes = sizeof(double) * ((size_t)(n + 1) * (n + 2) / 2 + n + 2);
    if (ba_chol_small_fits(d.n_cam)) return ba_solve_reduced_small(st, d, radius, min_diag, max_diag);
    if (bytes <= 150 * 1024) {
        ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_chol_solve_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL(ba_chol_solve_kernel<true>, dim3(1), dim3(kCholThreads), bytes, st, d, radius, min_diag, max_diag);
    } else {
        return ba_solve_reduced_large(st, d, radius, min_diag, max_diag);
    }
    LAUNCH_CHECK();
    return ESFM_OK;
}

// Scalar read-back without a stream synchronisation: one wave copies the scalar slots into pinned host memory, fences to system
// scope and then stores the sequence number the host is spinning on.
__global__ __launch_bounds__(64) void ba_publish_scalars_kernel(const double *__restrict__ scal, double *host, unsigned long long *flag,
                                                                unsigned long long seq)
{
    const int i = threadIdx.x;
    if (i < SC_COUNT) host[i] = scal[i];
    __threadfence_system();
    __syncthreads();
    if (i == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int ba_publish_scalars(hipStream_t st, const BADev &d, double *host, unsigned long long *flag, unsigned long long seq)
{
    hipLaunchKernelGGL(ba_publish_scalars_kernel, dim3(1), dim3(64), 0, st, d.scal, host, flag, seq);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_camera_step(hipStream_t st, const BADev &d)
{
    hipLaunchKernelGGL(ba_camera_step_kernel, dim3(1), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_backsub(hipStream_t st, const BADev &d)
{
    if (d.n_pt <= 0) return ESFM_OK;
    if (d.n_obs < (1 << 20)) {
        hipLaunchKernelGGL(ba_backsub_point_kernel, dim3(div_up(d.n_pt, 64)), dim3(64), 0, st, d);
        LAUNCH_CHECK();
        return ESFM_OK;
    }
    // points without observations keep their value; observed ones are overwritten by pass 2
    ESFM_HIP_TRY(hipMemcpyAsync(d.cand_p, d.x_p, sizeof(double) * 3 * (size_t)d.n_pt, hipMemcpyDeviceToDevice, st));
    ESFM_HIP_TRY(hipMemsetAsync(d.gE, 0, sizeof(double) * 3 * (size_t)d.n_pt, st));
    if (d.n_obs <= 0) return ESFM_OK;
    hipLaunchKernelGGL(ba_backsub_accum_kernel, dim3(div_up(d.n_obs, 256)), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(ba_backsub_apply_kernel, dim3(div_up(d.n_obs, 256)), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_cost(hipStream_t st, const BADev &d, int num_cu, const double *cams, const double *pts, double cauchy_a, int slot, int bad_slot,
            bool with_slope)
{
    if (d.n_obs <= 0) return ESFM_OK;
    const int grid = std::min(div_up(d.n_obs, 256), std::max(1, num_cu) * 8);
    if (with_slope) hipLaunchKernelGGL(ba_cost_kernel<true>, dim3(grid), dim3(256), 0, st, d, cams, pts, cauchy_a, slot, bad_slot);
    else hipLaunchKernelGGL(ba_cost_kernel<false>, dim3(grid), dim3(256), 0, st, d, cams, pts, cauchy_a, slot, bad_slot);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_take_step(hipStream_t st, const BADev &d, double t)
{
    const int grid = std::max(1, std::min(div_up(3LL * d.n_pt, 256), 1024));
    hipLaunchKernelGGL(ba_take_step_kernel, dim3(grid), dim3(256), 0, st, d, t);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_project_cameras(hipStream_t st, const BADev &d)
{
    if (d.n_cam <= 0) return ESFM_OK;
    hipLaunchKernelGGL(ba_project_cameras_kernel, dim3(div_up(6 * d.n_cam, 256)), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_schur_calib(hipStream_t st, const BADev &d)
{
    if (!d.has_calib || d.n_pt <= 0 || d.n_obs <= 0) return ESFM_OK;
    const size_t row_bytes = sizeof(double) * (8 + (size_t)4 * 6 * d.n_real_cam);
    const bool use_lds = row_bytes <= 64 * 1024;
    const size_t lds = use_lds ? row_bytes : sizeof(double) * 8;
    hipLaunchKernelGGL(ba_schur_calib_kernel, dim3(div_up(d.n_pt, 256)), dim3(256), lds, st, d, use_lds ? 1 : 0);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_param_sqnorm(hipStream_t st, const BADev &d)
{
    const int grid = std::max(1, std::min(div_up(3LL * d.n_pt, 256), 1024));
    hipLaunchKernelGGL(ba_param_sqnorm_kernel, dim3(grid), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_points_delta(hipStream_t st, const BADev &d, bool to_delta)
{
    if (d.n_pt <= 0) return ESFM_OK;
    hipLaunchKernelGGL(ba_points_delta_kernel, dim3(div_up(3LL * d.n_pt, 256)), dim3(256), 0, st, d, to_delta ? 1 : 0);
    LAUNCH_CHECK();
    return ESFM_OK;
}

}  // namespace esfm
