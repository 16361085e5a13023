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
#include <array>

#include "ba_kernels.hpp"
#include "ba_chol_sparse.hpp"

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
    // non-negative doubles order like their bit patterns.  A look first (L2-coherent; a value that is behind is only smaller): nearly every
    // caller loses, and thousands of read-modify-writes of ONE address queue up in one L2 channel -- ba_point_prep_chunk_kernel on BA-512,
    // 47 k waves: 151 us with the bare atomic, 54 us without its tail (round 3)
    unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= bits) return;
    atomicMax(a, bits);
}

// ba_schur_reduce_kernel sums entry e over the per-workgroup slabs with a 32 x 8 thread tile: thread (ent, grp) adds slabs grp,
// grp + 8, ... (independent loads, coalesced across ent), then the 8 partial sums are combined through LDS.
constexpr int kRedEnt = 32, kRedGrp = 8;

// Scalar sums across workgroups without f64 atomics (whose order of arrival differs from run to run): every workgroup leaves its
// block sums in d.scal_part[slot][base + blockIdx.x]; the next read-back (ba_publish_scalars / ba_scal_reduce, one workgroup) adds
// the pending partials of each slot in a fixed order -- one wave per slot, lane l takes entries l, l + 64, ..., then the shuffle tree --
// and adds the total to d.scal[slot].  The result is a function of the launch geometry only.  The host keeps the number of
// pending partials per slot (BADev::parts); `base` is where this launch's go.  vals are per-THREAD partials.
// (A first version let the last workgroup to take a ticket do the sum inside the producing kernel: the fence + ticket + tail cost
// ~10 us per kernel on the 25-camera problem, a tenth of the LM iteration.)
template <int N>
__device__ __forceinline__ void scal_commit(const BADev &d, const ScalBase &base, const int (&slots)[N], const double (&vals)[N], double *lds /* >= 8 */,
                                            int bidx = -1 /* the workgroup's index among those that commit; default blockIdx.x */)
{
    const int b = bidx < 0 ? (int)blockIdx.x : bidx;
    if (N == 1) {
        const double t = block_sum(vals[0], lds);
        if (threadIdx.x == 0) d.scal_part[(size_t)slots[0] * d.scal_cap + base.b[0] + b] = t;
        return;
    }
    // all N sums behind ONE pair of barriers (block_sum's shuffle tree and wave order per sum, hence its bits; N calls of it were 2 N
    // barriers at the end of every workgroup: twelve in the back-substitution)
    __shared__ double part[N][16];
    double t[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        t[q] = vals[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t[q] += __shfl_xor(t[q], o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) part[q][wave] = t[q];
    }
    __syncthreads();
    if (threadIdx.x < N) {
        const int q = threadIdx.x;
        double sum = 0.0;
        for (int w = 0; w < nw; ++w) sum += part[q][w];
        d.scal_part[(size_t)slots[q] * d.scal_cap + base.b[q] + b] = sum;
    }
}

// 1024 threads: wave w adds the pending partials of slot w (SC_SUM_COUNT <= 16) -- lane l takes entries l, l + 64, ... in order,
// 32 loads in flight, then the fixed shuffle tree -- so all slots cost one memory round trip together.
__device__ __forceinline__ void scal_reduce_pending(const double *scal_part, int scal_cap, double *scal, const ScalCounts &c)
{
    const int slot = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (slot >= SC_SUM_COUNT) return;
    const int n = c.n[slot];
    if (n == 0) return;
    const double *part = scal_part + (size_t)slot * scal_cap;
    double v = 0.0;
    // (32 loads in flight; the adds keep the order they have always had -- entries lane, lane + 64, ... -- so the sum's bits do not
    // depend on the depth: with 8 in flight BA-512's 12 k partials per slot were 23 dependent round trips, 17 us per read-back)
    for (int b = lane; b < n; b += 32 * 64) {
        double t[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) t[u] = (b + 64 * u < n) ? part[b + 64 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) v += t[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) scal[slot] += v;
}

__global__ __launch_bounds__(1024) void ba_scal_reduce_kernel(const double *__restrict__ scal_part, int scal_cap, double *scal, ScalCounts c)
{
    scal_reduce_pending(scal_part, scal_cap, scal, c);
}

// host side: where the partials of a launch of `grid` workgroups go, for each of its slots
template <int N>
static int scal_reserve(hipStream_t st, const BADev &d, const int (&slots)[N], int grid, ScalBase &base)
{
    if (grid > d.scal_cap) { set_error("BA: %d workgroups exceed the scalar partial capacity %d", grid, d.scal_cap); return ESFM_ERR_INVALID_ARG; }
    bool full = false;
    for (int q = 0; q < N; ++q) full = full || d.parts->n[slots[q]] + grid > d.scal_cap;
    if (full) { if (int rc = ba_scal_reduce(st, d)) return rc; }
    for (int q = 0; q < N; ++q) { base.b[q] = d.parts->n[slots[q]]; d.parts->n[slots[q]] += grid; }
    return ESFM_OK;
}

// Exact accumulation of the Schur complement in 64-bit fixed point (BADev::qexp): fx64(v, sh) = round(v 2^sh) as an integer.
// fx64_scaled: round-to-nearest-even of an already scaled x, |x| < 2^62, to int64 WITHOUT the compiler's f64 -> i64 lowering (which,
// with the ldexp, came to ~110 VALU instructions per entry: the Schur kernel was VALU bound on it).  Two magic-number additions:
// A = x + 1.5 2^84 rounds x to a multiple of 2^32, whose quotient sits in A's low mantissa dword; xl = x - (A - 1.5 2^84) is exact,
// |xl| <= 2^31, and B = xl + 1.5 2^52 holds round(xl) as a 64-bit two's complement offset of bits(1.5 2^52).  5 f64 + 1 int op.
__device__ __forceinline__ unsigned long long fx64_scaled(double x)
{
    const double M1 = 0x1.8p84, M2 = 0x1.8p52;
    const double A = __dadd_rn(x, M1);
    const double xl = __dsub_rn(x, __dsub_rn(A, M1));
    const double B = __dadd_rn(xl, M2);
    const unsigned int hi = (unsigned int)__double2loint(A) + ((unsigned int)__double2hiint(B) - 0x43380000u);
    return ((unsigned long long)hi << 32) | (unsigned int)__double2loint(B);
}
__device__ __forceinline__ unsigned long long fx64(double v, int sh) { return fx64_scaled(ldexp(v, sh)); }
__device__ __forceinline__ void fx_add(double *slot, double v, int sh) { atomicAdd(reinterpret_cast<unsigned long long *>(slot), fx64(v, sh)); }
__device__ __forceinline__ void fx_add_scaled(double *slot, double x) { atomicAdd(reinterpret_cast<unsigned long long *>(slot), fx64_scaled(x)); }

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

// W_i = F_i'E_i (6 x 3, row-major) of observation i, re-formed from its 18 Jacobian entries.  (Round 2 had the sweep store W -- 144 B
// per observation -- and the Schur kernels load it: as many loads here, 36 multiply-adds less.  Measured in round 3: without the
// store the sweep of BA-512 takes 262 us instead of 369, the Schur kernels 927 against 928 us; BA-25 29.2 against 30.7 and 41.5
// against 40.6 us.  The array is gone: 432 MB less written per sweep at BA-512, 912 MB -> 480 MB of linearisation resident.)
__device__ __forceinline__ void load_W(const BADev &d, size_t n_obs, int i, double W[18])
{
    double Jc[12], Jp[6];
#pragma unroll
    for (int q = 0; q < 12; ++q) Jc[q] = d.Jc[q * n_obs + i];
#pragma unroll
    for (int q = 0; q < 6; ++q) Jp[q] = d.Jp[q * n_obs + i];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int m = 0; m < 3; ++m) W[3 * a + m] = Jc[a] * Jp[m] + Jc[6 + a] * Jp[3 + m];
}

// e with sqrt(diag(F'F)_i) < 2^e (BADev::qexp)
__device__ __forceinline__ int qexp_of(double diag)
{
    const double sd = sqrt(fabs(diag));
    return (sd > 1e-120 && sd < 1e120) ? ilogb(sd) + 1 : -400;   // a zero column only ever contributes zeros
}

#ifdef ESFM_LIN_TRACE
// timing-only build (scratch/build_variant_ba.sh NAME -DESFM_LIN_TRACE): s_memrealtime ticks (10 ns) summed over the waves of ba_linearize_kernel:
// [0] prologue (LDS clear, exponents, first loads), [1] the loop, [2] everything behind it, [3] waves, [4] loop iterations
__device__ unsigned long long g_lin_trace[8];
__device__ unsigned long long g_lin_wave[512 * 8 * 5];
extern "C" int esfm_debug_lin_waves(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lin_wave), sizeof(g_lin_wave)); }
extern "C" int esfm_debug_lin_trace(unsigned long long *out, int reset)
{
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lin_trace), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lin_trace), sizeof(g_lin_trace));
}
#endif

// ---------------------------------------------------------------------------------------------
// Jacobian sweep: 16 B in, 160 B out per observation (+ 32 B with free intrinsics), and the per-camera sums F'F / F'r.
//
// PRIV (the 27 sums of every camera fit in LDS): each workgroup accumulates its observations' contributions in LDS as 64-bit
// FIXED-POINT integers -- integer adds are associative, so the order in which waves reach the LDS does not matter (round 1 used
// ds_add_f64 here and was not reproducible) -- and stores its private copy as one coalesced slab of doubles;
// ba_camacc_reduce_kernel adds the slabs in a fixed order.  The fixed-point scale is per workgroup, per camera and per column:
// pass 1 streams the Jacobian out and leaves  m[c][a] = max_k (J_k[a]^2 + J_k[6+a]^2)  (and max |r_k|^2) and the number of
// observations n_c of each camera in LDS (u64 max of non-negative doubles and integer adds: order-independent);
// then  sum_k f_a f_b <= sqrt(n_c m[c][a]) sqrt(n_c m[c][b]) < 2^(ex[c][a] + ex[c][b])  bounds every partial sum and pass 2 re-reads
// the workgroup's own Jacobian rows (L2) and adds round(v 2^(60 - ex[c][a] - ex[c][b])).
// !PRIV: the sweep only; ba_camacc_chunk_kernel then forms the sums by gathering each camera's observations in order.
// (Round 3, BA-25, one observation per thread: the kernel takes 29 - 31 us whether it writes 160 or 304 B per observation, re-reads
// its rows in pass 2 or keeps them in registers, shares one set of LDS sums per workgroup or keeps one per wave -- every variant
// measured.  It is one dependent chain per workgroup: index load -> parameter gather -> sincos / divide in f64 -> stores -> LDS
// maxima -> barrier -> exponents -> barrier -> 27 LDS adds -> barrier -> slab.  43 MB in 30 us is 0.18 of the HBM roofline and
// not what limits it.)
// threads per workgroup of the sweep: 512 from 2^20 observations on, 256 below (BA-25, 240 k observations, two per thread: 28.7 -> 22.8
// us -- finer workgroups get out of each other's phase and finish unevenly loaded CUs sooner; 128: 36 us, the slab count doubles;
// 1024: 33 us.  BA-512: 223 us with 512, 313 with 256)
// Round 5, BA-512 (two waves per SIMD -- 250 registers; the LDS sums of 512 cameras leave room for one workgroup per CU anyway): the
// loop's loads are issued one observation AHEAD of its stores (see the loop): 228 -> 187 us; ONE workgroup per CU instead of two in
// sequence (each paid a 9-us prologue -- LDS clear, exponents -- and an 8-us epilogue): -> 161 - 167 us.  What the loop waits for is
// neither memory nor issue slots but ITSELF (-DESFM_LIN_TRACE, scratch/lin_trace.py: 5.6 us per iteration and wave = 13 400 cycles
// for 2 x 1 266 instructions): timing-only builds without the 20 stores AND the 35 LDS atomics take 177 us against 186; a build that
// takes sin / cos / 1/theta from a per-camera table -- 1 080 instructions instead of 1 266, bit-identical results -- takes exactly as
// long (5.58 against 5.60 us per iteration: measured, not kept); allowing fused multiply-adds removes 8 % of the instructions.  The
// f64 dependency chains of one observation (division -> Jacobian -> scaling -> products -> fixed point) at two waves per SIMD are the
// critical path; more waves would need the kernel in 128 registers.  A pure copy of the same 18 + 18 arrays runs at 5.9 TB/s in ANY
// layout (SoA of doubles as here, double2, tiles of 64: scratch/ubench/soa_stream.hip), so the layout is not what holds the sweep back.
constexpr int kLinThreadsLarge = 512, kLinThreadsSmall = 256;
constexpr int kLinSmallObs = 1 << 20;
constexpr int kLinLdsPerCam = 27 * 8 + 7 * 8 + 4 + 7 * 4;   // acc, maxima, count, exponents

#ifndef ESFM_LIN_OCC
#define ESFM_LIN_OCC 1
#endif
template <bool PRIV, bool CALIB, int kLinThreads>
__global__ __launch_bounds__(kLinThreads, kLinThreads == 256 ? ESFM_LIN_OCC : 1) void ba_linearize_kernel(BADev d, double cauchy_a, int use_scaling, ScalBase sbase, double *__restrict__ slabs,
                                                                   int prov_rexp)
{
    // prov_rexp != INT_MIN (PRIV, several observations per thread: BA-512): PROVISIONAL fixed-point exponents -- the previous
    // linearisation's global column exponents d.qexp plus one, and prov_rexp for the residual column (sqrt(2 cost) < 2^prov_rexp,
    // from the host: the accepted candidate's cost) -- let pass 1 quantise and add the rows while they are in registers.  The maxima
    // and counts are still collected, and after the loop every workgroup checks ITS bound sqrt(n_c max) < 2^exponent for every camera
    // and column: if one fails (a first linearisation, a problem that changed under the solver) the workgroup clears its sums and
    // runs the two-pass form below.  Either way the integers are exact sums on the grid the slab is converted with.  (Round 3: the
    // re-read of pass 2 was 336 MB of BA-512's sweep.)
#ifdef ESFM_LIN_TRACE
    const unsigned long long lt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long lt_iters = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *red = reinterpret_cast<double *>(lds_raw);                                  // [8] reduction scratch, [10] intrinsics-block sums
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(lds_raw) + 18;     // PRIV: [n_real_cam * 27]
    unsigned long long *mx = acc + (size_t)d.n_real_cam * 27;                           //       [n_real_cam * 7]
    int *cnt = reinterpret_cast<int *>(mx + (size_t)d.n_real_cam * 7);                  //       [n_real_cam]
    int *ex = cnt + d.n_real_cam;                                                       //       [n_real_cam * 7]
    const int tid = threadIdx.x;
    const int n_obs = d.n_obs;
    const bool single_launch = PRIV && (long long)gridDim.x * kLinThreads >= n_obs;
    const bool prov = PRIV && !single_launch && prov_rexp != INT_MIN;
    if (PRIV) {
        for (int e = tid; e < d.n_real_cam * 27; e += kLinThreads) acc[e] = 0ull;
        for (int e = tid; e < d.n_real_cam * 7; e += kLinThreads) {
            mx[e] = 0ull;
            if (prov) ex[e] = e % 7 == 6 ? prov_rexp : d.qexp[6 * (e / 7) + e % 7] + 1;
        }
        for (int e = tid; e < d.n_real_cam; e += kLinThreads) cnt[e] = 0;
        if (tid == 0) red[17] = 0.0;         // the guard's verdict (red[8..17] otherwise belong to the intrinsics block, which has no provisional path)
        __syncthreads();
    }
    // the 27 fixed-point adds of one observation's rows (every factor carries 2^(30 - exponent of its column))
    auto add_rows = [&](int c, const double (&Jrow)[12], double r0, double r1) {
        double J[12];
#pragma unroll
        for (int a = 0; a < 6; ++a) { const int sh = kFxBits / 2 - ex[c * 7 + a]; J[a] = ldexp(Jrow[a], sh); J[6 + a] = ldexp(Jrow[6 + a], sh); }
        const int shr = kFxBits / 2 - ex[c * 7 + 6];
        const double q0 = ldexp(r0, shr), q1 = ldexp(r1, shr);
        int e = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = a; b < 6; ++b) {
                atomicAdd(&acc[c * 27 + e], fx64_scaled(J[a] * J[b] + J[6 + a] * J[6 + b]));
                ++e;
            }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) atomicAdd(&acc[c * 27 + 21 + a], fx64_scaled(J[a] * q0 + J[6 + a] * q1));
    };
    double cost = 0.0, bad = 0.0;
    double kacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // PRIV, at most one observation per thread (the grid covers the observations: BA-25): pass 2 quantises the rows from REGISTERS
    // instead of re-reading the thread's own stores (112 B per observation fetched back: 33 MB of the 108 MB the kernel moved)
    const bool single = single_launch;
    double keepJ[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, keepR0 = 0.0, keepR1 = 0.0;
    int keepC = -1;
    // The loop is software-pipelined by hand.  A wave's loads and stores retire through ONE counter (vmcnt) in issue order: with
    // the next observation's index loads and parameter gathers issued AFTER this observation's 20 stores, "wait for the loads" meant
    // "wait until the stores have drained" -- two dependent round trips behind a write burst per iteration, 15 us of a 20-us iteration
    // idle at the two waves per SIMD the LDS sums leave room for.  Now the indices of observation k + 2 strides and the parameters of
    // k + stride are requested in front of the ARITHMETIC of k: they are ahead of its stores in the queue, their latency runs beside
    // 1 200 instructions, and the wait in front of the next iteration leaves the stores in flight.
    // (Unconditional loads at clamped indices: a load under `if (k + stride < n_obs)` merges with the old value, and the merge is a
    // register copy that has to wait for the load -- in front of the stores.  The camera's float calibration is converted where it
    // is used, for the same reason.)
    struct LinIdx { int c, p; float2 uv; };
    struct LinPar { float4 K; double kd[4], cam[6], X[3], sc[6], sp[3]; };
    auto load_idx = [&](int k, LinIdx &o) { o.c = d.obs_cam[k]; o.p = d.obs_pt[k]; o.uv = d.obs_uv[k]; };
    auto load_par = [&](const LinIdx &ix, LinPar &o) {
        if (CALIB) {         // (== d.has_calib: the free block rides behind the real cameras, load_intrinsics)
            const double *kp = d.x_c + 6 * (size_t)d.n_real_cam;
            o.kd[0] = kp[0]; o.kd[1] = kp[1]; o.kd[2] = kp[2]; o.kd[3] = kp[3];
        } else {
            o.K = d.K4[ix.c];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) o.cam[i] = d.x_c[6 * (size_t)ix.c + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) o.X[i] = d.x_p[3 * (size_t)ix.p + i];
        if (use_scaling) {
#pragma unroll
            for (int i = 0; i < 6; ++i) o.sc[i] = d.scale_c[6 * (size_t)ix.c + i];
#pragma unroll
            for (int i = 0; i < 3; ++i) o.sp[i] = d.scale_p[3 * (size_t)ix.p + i];
        }
    };
    double kscale[4] = {1.0, 1.0, 1.0, 1.0};      // the free intrinsics' column scales: the same for every observation, read once
    if (CALIB && use_scaling) {
#pragma unroll
        for (int i = 0; i < 4; ++i) kscale[i] = d.scale_c[6 * (size_t)d.n_real_cam + i];
    }
    const int kstride = gridDim.x * kLinThreads;
    const int k_first = blockIdx.x * kLinThreads + tid;
    auto clamped = [&](long long k) { return (int)(k < n_obs ? k : n_obs - 1); };
    LinIdx ix_cur = {0, 0, make_float2(0.f, 0.f)}, ix_nxt = ix_cur;
    LinPar par_cur = {};
    if (n_obs > 0) {
        load_idx(clamped(k_first), ix_cur); load_par(ix_cur, par_cur);
        load_idx(clamped((long long)k_first + kstride), ix_nxt);
        // (vmcnt(0) here, once: the compiler's wait-count bookkeeping merges the loop's two entries, and with these loads pending on
        // the way in it would wait for "all but two" memory operations in EVERY iteration -- the stores again)
        __builtin_amdgcn_s_waitcnt(0x0f70);
    }
#ifdef ESFM_LIN_TRACE
    const unsigned long long lt1 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int k = k_first; k < n_obs; k += kstride) {
#ifdef ESFM_LIN_TRACE
        ++lt_iters;
#endif
        // the next observation's parameters and the indices of the one after: requested in front of this observation's arithmetic
        // (and so ahead of its stores in the queue)
        // (with free intrinsics the kernel has no registers left for that: there the requests go out after the arithmetic, in front
        // of the stores)
        LinPar par_nxt;
        LinIdx ix_nn;
        if (!CALIB) {
            load_par(ix_nxt, par_nxt);
            load_idx(clamped((long long)k + 2 * (long long)kstride), ix_nn);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int c = ix_cur.c;
        const float2 uv = ix_cur.uv;
        double in4[4];
        if (CALIB) { in4[0] = par_cur.kd[0]; in4[1] = par_cur.kd[1]; in4[2] = par_cur.kd[2]; in4[3] = par_cur.kd[3]; }
        else { in4[0] = (double)par_cur.K.x; in4[1] = (double)par_cur.K.y; in4[2] = (double)par_cur.K.z; in4[3] = (double)par_cur.K.w; }
        double r0, r1, Jc[12], Jp[6], xn, yn;
        reproject_jac(par_cur.cam, par_cur.X, in4, uv, r0, r1, Jc, Jp, xn, yn);
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
                const double sc = use_scaling ? sq * par_cur.sc[i] : sq;
                Jc[i] *= sc; Jc[6 + i] *= sc;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double sp = use_scaling ? sq * par_cur.sp[i] : sq;
                Jp[i] *= sp; Jp[3 + i] *= sp;
            }
            if (CALIB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Jk[i] *= sq * kscale[i];
            }
        }
        if (CALIB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) d.Jk[(size_t)i * n_obs + k] = Jk[i];
            if (PRIV) {
                // G'G (upper triangle of the 4x4: rows fx, cx | fy, cy never mix) and G'r, kept in registers over the sweep
                kacc[0] += Jk[0] * Jk[0]; kacc[1] += Jk[0] * Jk[1]; kacc[2] += Jk[1] * Jk[1];
                kacc[3] += Jk[2] * Jk[2]; kacc[4] += Jk[2] * Jk[3]; kacc[5] += Jk[3] * Jk[3];
                kacc[6] += Jk[0] * r0; kacc[7] += Jk[1] * r0; kacc[8] += Jk[2] * r1; kacc[9] += Jk[3] * r1;
            }
        }
        if (CALIB) {
            load_par(ix_nxt, par_nxt);
            load_idx(clamped((long long)k + 2 * (long long)kstride), ix_nn);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) d.Jc[(size_t)i * n_obs + k] = Jc[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) d.Jp[(size_t)i * n_obs + k] = Jp[i];
        d.res[k] = r0; d.res[(size_t)n_obs + k] = r1;
        if (PRIV) {
            atomicAdd(&cnt[c], 1);
#pragma unroll
            for (int a = 0; a < 6; ++a) atomicMax(&mx[c * 7 + a], (unsigned long long)__double_as_longlong(Jc[a] * Jc[a] + Jc[6 + a] * Jc[6 + a]));
            atomicMax(&mx[c * 7 + 6], (unsigned long long)__double_as_longlong(r0 * r0 + r1 * r1));
            if (single) {
#pragma unroll
                for (int i = 0; i < 12; ++i) keepJ[i] = Jc[i];
                keepR0 = r0; keepR1 = r1; keepC = c;
            }
            if (prov && !CALIB) add_rows(c, Jc, r0, r1);
        }
        ix_cur = ix_nxt; ix_nxt = ix_nn; par_cur = par_nxt;
    }
#ifdef ESFM_LIN_TRACE
    const unsigned long long lt2 = __builtin_amdgcn_s_memrealtime();
    struct LinTraceEnd {
        unsigned long long t0, t1, t2, it;
        __device__ ~LinTraceEnd() {
            if ((threadIdx.x & 63) == 0) {
                const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
                atomicAdd(&g_lin_trace[0], t1 - t0); atomicAdd(&g_lin_trace[1], t2 - t1); atomicAdd(&g_lin_trace[2], t3 - t2);
                atomicAdd(&g_lin_trace[3], 1ull); atomicAdd(&g_lin_trace[4], it);
                if (blockIdx.x < 512) {       // per wave of the last launch: start, loop start, loop end, end (10-ns ticks), XCC id (scratch/lin_trace.py)
                    unsigned long long *w = &g_lin_wave[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 5];
                    w[0] = t0; w[1] = t1; w[2] = t2; w[3] = t3; w[4] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15;   // HW_REG_XCC_ID
                }
            }
        }
    } lin_trace_end{lt0, lt1, lt2, lt_iters};
#endif
    {
        const int slots[2] = {SC_COST, SC_LIN_BAD};
        const double vals[2] = {cost, bad};
        scal_commit<2>(d, sbase, slots, vals, red);
    }
    if (!PRIV) return;
    if (CALIB) {
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const double v = block_sum(kacc[q], red);     // a fixed tree
            if (tid == 0) red[8 + q] = v;
        }
    }
    __syncthreads();
    bool two_pass = !(prov && !CALIB);
    if (!two_pass) {
        // the guard: this workgroup's bound on every partial sum against the provisional grid
        bool viol = false;
        for (int e = tid; e < d.n_real_cam * 7; e += kLinThreads) {
            if (cnt[e / 7] == 0) continue;
            const double b = sqrt((double)cnt[e / 7] * __longlong_as_double((long long)mx[e]));
            const int need = (b > 1e-120 && b < 1e120) ? ilogb(b) + 1 : (b >= 1e120 || b != b ? INT_MAX : -400);
            viol = viol || need > ex[e];
        }
        if (viol) red[17] = 1.0;
        __syncthreads();
        two_pass = red[17] != 0.0;            // (workgroup-uniform)
        if (two_pass) {
            for (int e = tid; e < d.n_real_cam * 27; e += kLinThreads) acc[e] = 0ull;
        }
        __syncthreads();
    }
    if (two_pass) {
        for (int e = tid; e < d.n_real_cam * 7; e += kLinThreads) {
            const double b = sqrt((double)cnt[e / 7] * __longlong_as_double((long long)mx[e]));   // > sqrt(sum): every term is <= the maximum
            ex[e] = (b > 1e-120 && b < 1e120) ? ilogb(b) + 1 : -400;
        }
        __syncthreads();
    }
    for (int k = blockIdx.x * kLinThreads + tid; two_pass && k < n_obs; k += gridDim.x * kLinThreads) {
        int c;
        double J[12], r0, r1;
        if (single) {
            c = keepC; r0 = keepR0; r1 = keepR1;
#pragma unroll
            for (int i = 0; i < 12; ++i) J[i] = keepJ[i];
        } else {
            c = d.obs_cam[k];
#pragma unroll
            for (int i = 0; i < 12; ++i) J[i] = d.Jc[(size_t)i * n_obs + k];    // this thread's own stores
            r0 = d.res[k]; r1 = d.res[(size_t)n_obs + k];
        }
        add_rows(c, J, r0, r1);
    }
    __syncthreads();
    // one coalesced slab of doubles per workgroup; ba_camacc_reduce_kernel sums them in a fixed order
    double *out = slabs + (size_t)blockIdx.x * d.n_cam * 27;
    for (int e = tid; e < d.n_real_cam * 27; e += kLinThreads) {
        const int c = e / 27, q = e % 27;
        int a = 0, b2 = 6;                      // q >= 21: column a = q - 21 against the residual
        if (q >= 21) a = q - 21;
        else { int rem = q; while (rem >= 6 - a) { rem -= 6 - a; ++a; } b2 = a + rem; }
        out[e] = fx64_to_double(acc[e], kFxBits - ex[c * 7 + a] - ex[c * 7 + b2]);
    }
    if (CALIB) {
        // the intrinsics block rides as camera-side block n_real_cam.  Packed index of (a, b), a <= b: a*6 - a(a-1)/2 + b - a.
        const int slot[10] = {0, 1, 6, 11, 12, 15, 21, 22, 23, 24};
        if (tid < 27) {
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < 10; ++q) if (slot[q] == tid) v = red[8 + q];
            out[(size_t)d.n_real_cam * 27 + tid] = v;
        }
    }
}

// Camera part of max|gradient| (needs the all-reduced F'r): |F'r_i / scale_i|, or for bounded problems the norm of
// x - Plus(x, -gradient)  (trust_region_minimizer.cc, projected gradient step).
__device__ __forceinline__ double camera_gradient_entry(const BADev &d, int i, double ftr)
{
    double g = ftr / d.scale_c[i];
    if (d.constrained) {
        const double x = d.x_c[i];
        g = x - fmin(fmax(x - g, d.lo_c[i]), d.up_c[i]);
    }
    return fabs(g);
}

// Sum of entry e over n_slabs per-workgroup slabs, computed by a 32 x 8 thread tile: thread (ent, grp) adds slabs grp, grp + 8, ...
// (independent loads, coalesced across ent), then the 8 partial sums are combined in a fixed order through LDS.  Deterministic.
__device__ __forceinline__ void camacc_reduce_block(const BADev &d, const double *__restrict__ slabs, int n_slabs, int with_gradient, int bidx, double *lds /* 256 */)
{
    const int e = bidx * kRedEnt + (threadIdx.x % kRedEnt);
    const int per = d.n_cam * 27;
    const int grp = threadIdx.x / kRedEnt;
    // thread (ent, grp) adds slabs grp, grp + 8, ... in that order; 32 loads in flight per round (512 slabs: two round trips; the loop
    // used to wait for every load, ~1 us each, then for every eighth)
    double v0 = 0.0;
    if (e < per) {
        for (int b = grp; b < n_slabs; b += 32 * kRedGrp) {
            double t[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) t[u] = (b + u * kRedGrp < n_slabs) ? slabs[(size_t)(b + u * kRedGrp) * per + e] : 0.0;
#pragma unroll
            for (int u = 0; u < 32; ++u) v0 += t[u];
        }
    }
    lds[threadIdx.x] = v0;
    __syncthreads();
    if (threadIdx.x >= kRedEnt || e >= per) return;
    double v = 0.0;
    for (int g = 0; g < kRedGrp; ++g) v += lds[g * kRedEnt + threadIdx.x];
    const int c = e / 27, q = e % 27;
    if (q >= 21) {
        d.camacc[36 * (size_t)d.n_cam + 6 * (size_t)c + (q - 21)] = v;
        // one rank: this IS the gradient's camera part, max-reduced here instead of in a launch of its own
        if (with_gradient) { const double g = camera_gradient_entry(d, 6 * c + (q - 21), v); if (g > 0.0) atomic_max_nonneg(&d.scal[SC_GMAX], g); }
        return;
    }
    int a = 0, rem = q;
    while (rem >= 6 - a) { rem -= 6 - a; ++a; }
    const int b2 = a + rem;
    d.camacc[36 * (size_t)c + 6 * a + b2] = v;
    d.camacc[36 * (size_t)c + 6 * b2 + a] = v;
    if (a == b2) d.qexp[6 * c + a] = qexp_of(v);
}

__global__ __launch_bounds__(256) void ba_camacc_reduce_kernel(BADev d, const double *__restrict__ slabs, int n_slabs, int with_gradient)
{
    __shared__ double lds[256];
    camacc_reduce_block(d, slabs, n_slabs, with_gradient, blockIdx.x, lds);
}

// Per-camera sums from the stored Jacobian, chunk by chunk: ONE WAVE per chunk adds F'F (upper triangle, 21), F'r (6) -- and with
// free intrinsics the 10 sums of the shared block, G'G (6) and G'r (4) -- over the (up to kCamChunk = 256) observations
// cam_obs[beg, end) of ONE camera.  Lane l takes observations l, l + 64, l + 128, l + 192 in that order (all four index loads,
// then all Jacobian loads, are issued before the first use: the kernel is two memory round trips deep, not eight); the 37 sums
// then go through the wave shuffle tree.  A fixed association: bit-reproducible.
template <bool CALIB>
__global__ __launch_bounds__(64) void ba_camacc_chunk_kernel(BADev d)
{
    const int lane = threadIdx.x;
    const int beg = d.cchunk_beg[blockIdx.x], end = d.cchunk_end[blockIdx.x];
    const size_t n = d.n_obs;
    constexpr int U = kCamChunk / 64;
    int k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int idx = beg + lane + 64 * u; k[u] = idx < end ? d.cam_obs[idx] : -1; }
    double J[U][12], r0[U], r1[U], K[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int kk = k[u] < 0 ? 0 : k[u];
#pragma unroll
        for (int a = 0; a < 12; ++a) J[u][a] = d.Jc[a * n + kk];
        r0[u] = d.res[kk]; r1[u] = d.res[n + kk];
        if (CALIB) {
#pragma unroll
            for (int a = 0; a < 4; ++a) K[u][a] = d.Jk[a * n + kk];
        }
    }
    double acc[kCamPart];
#pragma unroll
    for (int q = 0; q < kCamPart; ++q) acc[q] = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (k[u] < 0) continue;
        int e = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) acc[e++] += J[u][a] * J[u][b] + J[u][6 + a] * J[u][6 + b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += J[u][a] * r0[u] + J[u][6 + a] * r1[u];
        if (CALIB) {
            const double k0 = K[u][0], k1 = K[u][1], k2 = K[u][2], k3 = K[u][3];
            acc[27] += k0 * k0; acc[28] += k0 * k1; acc[29] += k1 * k1; acc[30] += k2 * k2; acc[31] += k2 * k3; acc[32] += k3 * k3;
            acc[33] += k0 * r0[u]; acc[34] += k1 * r0[u]; acc[35] += k2 * r1[u]; acc[36] += k3 * r1[u];
        }
    }
    constexpr int NQ = CALIB ? kCamPart : 27;
    double mine = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double v = acc[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == q) mine = v;
    }
    if (lane < NQ) d.cam_part[(size_t)blockIdx.x * kCamPart + lane] = mine;
}

// camacc = F'F (36 per camera, both triangles) | F'r (6 per camera): a camera's chunk sums added in chunk order; the
// intrinsics block (free calibration) adds ALL chunks in order.  Every entry of camacc is written (cameras without observations get
// zeros), and the diagonal entries also leave their fixed-point exponent in qexp.
__global__ __launch_bounds__(256) void ba_camacc_final_kernel(BADev d)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int nr = d.n_real_cam;
    if (e < nr * 27) {
        const int c = e / 27, q = e % 27;
        double v = 0.0;
        for (int b = d.cam_chunk0[c]; b < d.cam_chunk0[c + 1]; ++b) v += d.cam_part[(size_t)b * kCamPart + q];
        if (q >= 21) { d.camacc[36 * (size_t)d.n_cam + 6 * (size_t)c + (q - 21)] = v; return; }
        int a = 0, rem = q;
        while (rem >= 6 - a) { rem -= 6 - a; ++a; }
        const int b2 = a + rem;
        d.camacc[36 * (size_t)c + 6 * a + b2] = v;
        d.camacc[36 * (size_t)c + 6 * b2 + a] = v;
        if (a == b2) d.qexp[6 * c + a] = qexp_of(v);
        return;
    }
    const int q = e - nr * 27;
    if (!d.has_calib || q >= 42) return;
    // the shared intrinsics block rides as camera-side block nr: 6 x 6 with the 4 x 4 part G'G (fx, cx | fy, cy never mix), then G'r
    const int kc = nr;
    if (q < 36) {
        const int a = q / 6, b2 = q % 6;
        const int lo = a < b2 ? a : b2, hi = a < b2 ? b2 : a;
        int src = -1;
        if (lo == 0 && hi == 0) src = 27; else if (lo == 0 && hi == 1) src = 28; else if (lo == 1 && hi == 1) src = 29;
        else if (lo == 2 && hi == 2) src = 30; else if (lo == 2 && hi == 3) src = 31; else if (lo == 3 && hi == 3) src = 32;
        double v = 0.0;
        if (src >= 0) for (int b = 0; b < d.n_cchunks; ++b) v += d.cam_part[(size_t)b * kCamPart + src];
        d.camacc[36 * (size_t)kc + q] = v;
        if (a == b2) d.qexp[6 * kc + a] = qexp_of(v);
    } else {
        const int a = q - 36;
        double v = 0.0;
        if (a < 4) for (int b = 0; b < d.n_cchunks; ++b) v += d.cam_part[(size_t)b * kCamPart + 33 + a];
        d.camacc[36 * (size_t)d.n_cam + 6 * (size_t)kc + a] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Per point: E'E, E'r (only when the Jacobian is fresh), then M^-1 = (E'E + clamp(diag)/radius)^-1
// via a 3x3 Cholesky (ceres InvertPSDMatrix), M^-1 E'r, and the point part of max|gradient|.
// M = E'E + D^2 of one point, its inverse (closed-form 3 x 3 Cholesky) and M^-1 E'r; returns 1.0 when M is not positive definite
__device__ __forceinline__ double point_block_invert(const BADev &d, int p, const double (&A)[6], const double (&g)[3], double radius, double min_diag,
                                                     double max_diag)
{
    double sing = 0.0;
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
    return sing;
}

// one thread = one point: (fresh) E'E and E'r from the track, then the damped 3 x 3 inverse; returns its |gradient| share and singular flag
__device__ __forceinline__ void point_prep_thread(const BADev &d, int p, double radius, double min_diag, double max_diag, int fresh, double &gmax, double &sing)
{
    gmax = 0.0; sing = 0.0;
    if (p >= d.n_pt) return;
    const int b = d.pt_start[p], e = d.pt_start[p + 1];
    if (e <= b) return;
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
    sing = point_block_invert(d, p, A, g, radius, min_diag, max_diag);
}

__global__ __launch_bounds__(64) void ba_point_prep_kernel(BADev d, double radius, double min_diag, double max_diag, int fresh, ScalBase sbase)
{
    __shared__ double red[8];
    double gmax, sing;
    point_prep_thread(d, blockIdx.x * 64 + threadIdx.x, radius, min_diag, max_diag, fresh, gmax, sing);
    if (fresh) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, o));
        if ((threadIdx.x & 63) == 0 && gmax > 0.0) atomic_max_nonneg(&d.scal[SC_GMAX], gmax);
    }
    const int slots[1] = {SC_PT_SINGULAR};
    const double vals[1] = {sing};
    scal_commit<1>(d, sbase, slots, vals, red);
}

// The per-point blocks of a fresh Jacobian AND the reduction of the sweep's per-camera slabs in one launch (one rank, small
// problems): the two are independent -- the slab sums are first needed by the reduced solve -- and each leaves most of the chip
// idle (22 workgroups for 25 cameras; a latency-bound walk over the tracks), so back to back they cost 8 + 12 us of a 180-us
// LM iteration on BA-25 and together 12.  The first n_red workgroups are ba_camacc_reduce_kernel's, the others take 256 points
// each (the singular-point count is an integer-valued sum and the gradient maximum a maximum: the wider workgroup changes no bit).
#ifndef ESFM_PREP_OCC
#define ESFM_PREP_OCC 1
#endif
__global__ __launch_bounds__(256, ESFM_PREP_OCC) void ba_point_prep_camacc_kernel(BADev d, double radius, double min_diag, double max_diag, ScalBase sbase,
                                                                   const double *__restrict__ slabs, int n_slabs, int with_gradient, int n_red)
{
    __shared__ double lds[256];
    if ((int)blockIdx.x < n_red) { camacc_reduce_block(d, slabs, n_slabs, with_gradient, blockIdx.x, lds); return; }
    const int pb = (int)blockIdx.x - n_red;
    double gmax, sing;
    point_prep_thread(d, pb * 256 + threadIdx.x, radius, min_diag, max_diag, 1, gmax, sing);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, o));
    if ((threadIdx.x & 63) == 0 && gmax > 0.0) atomic_max_nonneg(&d.scal[SC_GMAX], gmax);
    const int slots[1] = {SC_PT_SINGULAR};
    const double vals[1] = {sing};
    scal_commit<1>(d, sbase, slots, vals, lds, pb);
}

// The fresh-Jacobian variant for large problems, one workgroup per point chunk (pchunk_pt0, <= 256 observations): thread =
// observation forms its nine products from ONE coalesced read of its Jp / res rows -> LDS; thread = point adds them in observation
// order (the same order, hence the same bits, as ba_point_prep_kernel's walk over the track) and inverts its block.  Throughput
// against latency: on BA-512 (3M observations) the walk costs 0.27 ms, this form 0.15 ms; on BA-25 (240k) this form is
// slower (17.6 us against 12 us: the 3 x 3 inverse's sqrt / divide chain runs in one half-filled wave per chunk), so the
// launcher picks by size.  A single point with more observations than a chunk holds walks its track like the plain kernel.
__global__ __launch_bounds__(kPtChunkObs) void ba_point_prep_chunk_kernel(BADev d, double radius, double min_diag, double max_diag, ScalBase sbase)
{
    __shared__ double red[8];
    __shared__ double prod[kPtChunkObs][9];
    const int tid = threadIdx.x;
    const int4 ci4 = d.pchunk_info[blockIdx.x];          // (one load where p0 / p1, then pt_start[p0] / pt_start[p1], were two dependent ones)
    const int p0 = ci4.x, p1 = ci4.y, k0 = ci4.z, k1 = ci4.w;
    const int nobs = k1 - k0;
    const size_t n = d.n_obs;
    double gmax = 0.0, sing = 0.0;
    const bool small = nobs <= kPtChunkObs;
    // what the point phase needs from memory is requested NOW, in front of the rows (round 5): behind the barrier it was two more
    // dependent round trips in a workgroup whose whole life is five, with 9 of 10 threads idle
    const bool is_pt = tid < p1 - p0;
    int pt_b = 0, pt_e = 0;
    double pt_sc[3] = {1.0, 1.0, 1.0};
    if (is_pt) {
        pt_b = d.pt_start[p0 + tid]; pt_e = d.pt_start[p0 + tid + 1];
#pragma unroll
        for (int q = 0; q < 3; ++q) pt_sc[q] = d.scale_p[3 * (size_t)(p0 + tid) + q];
    }
    if (small && tid < nobs) {
        const int k = k0 + tid;
        const double j0 = d.Jp[k], j1 = d.Jp[n + k], j2 = d.Jp[2 * n + k];
        const double j3 = d.Jp[3 * n + k], j4 = d.Jp[4 * n + k], j5 = d.Jp[5 * n + k];
        const double r0 = d.res[k], r1 = d.res[n + k];
        prod[tid][0] = j0 * j0 + j3 * j3; prod[tid][1] = j0 * j1 + j3 * j4; prod[tid][2] = j0 * j2 + j3 * j5;
        prod[tid][3] = j1 * j1 + j4 * j4; prod[tid][4] = j1 * j2 + j4 * j5; prod[tid][5] = j2 * j2 + j5 * j5;
        prod[tid][6] = j0 * r0 + j3 * r1; prod[tid][7] = j1 * r0 + j4 * r1; prod[tid][8] = j2 * r0 + j5 * r1;
    }
    __syncthreads();
    if (is_pt) {
        const int p = p0 + tid;
        const int b = pt_b, e = pt_e;
        if (e > b) {
            double A[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
            if (small) {
                for (int o = b - k0; o < e - k0; ++o) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) A[q] += prod[o][q];
#pragma unroll
                    for (int q = 0; q < 3; ++q) g[q] += prod[o][6 + q];
                }
            } else {
                for (int k = b; k < e; ++k) {
                    const double j0 = d.Jp[k], j1 = d.Jp[n + k], j2 = d.Jp[2 * n + k];
                    const double j3 = d.Jp[3 * n + k], j4 = d.Jp[4 * n + k], j5 = d.Jp[5 * n + k];
                    const double r0 = d.res[k], r1 = d.res[n + k];
                    A[0] += j0 * j0 + j3 * j3; A[1] += j0 * j1 + j3 * j4; A[2] += j0 * j2 + j3 * j5;
                    A[3] += j1 * j1 + j4 * j4; A[4] += j1 * j2 + j4 * j5; A[5] += j2 * j2 + j5 * j5;
                    g[0] += j0 * r0 + j3 * r1; g[1] += j1 * r0 + j4 * r1; g[2] += j2 * r0 + j5 * r1;
                }
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) d.EtE[6 * (size_t)p + q] = A[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                d.Etr[3 * (size_t)p + q] = g[q];
                gmax = fmax(gmax, fabs(g[q] / pt_sc[q]));  // gradient of the unscaled problem
            }
            sing = point_block_invert(d, p, A, g, radius, min_diag, max_diag);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, o));
    if ((threadIdx.x & 63) == 0 && gmax > 0.0) atomic_max_nonneg(&d.scal[SC_GMAX], gmax);
    const int slots[1] = {SC_PT_SINGULAR};
    const double vals[1] = {sing};
    scal_commit<1>(d, sbase, slots, vals, red);
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

__global__ void ba_camera_gradient_kernel(BADev d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double g = 0.0;
    if (i < 6 * d.n_cam) g = camera_gradient_entry(d, i, d.camacc[36 * (size_t)d.n_cam + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) g = fmax(g, __shfl_xor(g, o));
    if ((threadIdx.x & 63) == 0 && g > 0.0) atomic_max_nonneg(&d.scal[SC_GMAX], g);
}

// ---------------------------------------------------------------------------------------------
// Point-block Schur complement (schur_eliminator_impl.h [upstream]).  Thread i owns observation i of
// point p: W_i = F_i'E_i, Y_i = W_i M^-1; rhs_corr[c_i] -= W_i M^-1 E'r; and for every observation j of
// the same point with camera(j) <= camera(i):  S[c_i][c_j] -= Y_i W_j'.  Only blocks on or below the
// block diagonal are produced (the factorisation reads the lower triangle).
// One pair of observations (i, j) of a point: block (c_i, c_j) of the Schur complement gets -Y_i W_j' (Y_i = W_i M^-1, rows scaled
// by 2^(60 - qexp[row]); W_j's columns by 2^-qexp[col] here), as fixed-point adds.  Entry (a, c2) goes to Sb[a * ra + c2 * rc]:
// (ra, rc) = (6, 1) normally, (1, 6) when the pair was met with c_j > c_i and the contribution belongs to the stored block
// (c_j, c_i) as its transpose -- run-time strides, so that a wave whose lanes disagree does not execute the loop twice.
// BOTH: a camera that sees the point twice -- the diagonal block is stored in full and takes the product and its transpose.
template <bool BOTH>
__device__ __forceinline__ void schur_pair(double *Sb, int ra, int rc, const double (&Y)[18], const double (&Wj)[18], const int *qe_j)
{
#pragma unroll
    for (int c2 = 0; c2 < 6; ++c2) {
        const int ej = -qe_j[c2];
        const double w0 = ldexp(Wj[3 * c2], ej), w1 = ldexp(Wj[3 * c2 + 1], ej), w2 = ldexp(Wj[3 * c2 + 2], ej);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const unsigned long long v = fx64_scaled(-fma(Y[3 * a + 2], w2, fma(Y[3 * a + 1], w1, Y[3 * a] * w0)));   // (explicit fma: the file is built with -ffp-contract=off)
            atomicAdd(reinterpret_cast<unsigned long long *>(&Sb[a * ra + c2 * rc]), v);
            if (BOTH) atomicAdd(reinterpret_cast<unsigned long long *>(&Sb[a * rc + c2 * ra]), v);
        }
    }
}

// list != NULL: the observations list[0 .. n_list) only (the wide tracks the windowed kernel leaves out)
__global__ __launch_bounds__(256) void ba_schur_kernel(BADev d, int rhs_exp, const int32_t *__restrict__ list, int n_list)
{
    const int li = blockIdx.x * 256 + threadIdx.x;
    if (li >= (list ? n_list : d.n_obs)) return;
    const int i = list ? list[li] : li;
    const size_t n_obs = d.n_obs;
    const int n = 6 * d.n_cam;
    const int p = d.obs_pt[i], ci = d.obs_cam[i];
    double W[18];
    load_W(d, n_obs, i, W);
    const double *Mi = d.Minv + 6 * (size_t)p;
    const double M[9] = {Mi[0], Mi[1], Mi[2], Mi[1], Mi[3], Mi[4], Mi[2], Mi[4], Mi[5]};
    const double ag0 = d.Aig[3 * (size_t)p], ag1 = d.Aig[3 * (size_t)p + 1], ag2 = d.Aig[3 * (size_t)p + 2];
    const int b = d.pt_start[p], T = d.pt_start[p + 1] - b, t = i - b;
    double Y[18];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const double w0 = W[3 * a], w1 = W[3 * a + 1], w2 = W[3 * a + 2];
        const int ei = kFxBits - d.qexp[6 * ci + a];
        fx_add(&d.red[(size_t)n * n + 6 * ci + a], -(w0 * ag0 + w1 * ag1 + w2 * ag2), ei - rhs_exp);
        Y[3 * a + 0] = ldexp(w0 * M[0] + w1 * M[3] + w2 * M[6], ei);
        Y[3 * a + 1] = ldexp(w0 * M[1] + w1 * M[4] + w2 * M[7], ei);
        Y[3 * a + 2] = ldexp(w0 * M[2] + w1 * M[5] + w2 * M[8], ei);
    }
    schur_pair<false>(d.red + (size_t)(6 * ci) * n + 6 * ci, n, 1, Y, W, d.qexp + 6 * ci);
    for (int s2 = 1; 2 * s2 <= T; ++s2) {
        if (2 * s2 == T && t < s2) continue;
        int q = t - s2; if (q < 0) q += T;
        const int j = b + q;
        const int cj = d.obs_cam[j];
        double Wj[18];
        load_W(d, n_obs, j, Wj);
        const int hi = cj > ci ? cj : ci, lo = cj > ci ? ci : cj;
        double *Sb = d.red + (size_t)(6 * hi) * n + 6 * lo;
        if (cj != ci) schur_pair<false>(Sb, cj > ci ? 1 : n, cj > ci ? n : 1, Y, Wj, d.qexp + 6 * cj);
        else schur_pair<true>(Sb, n, 1, Y, Wj, d.qexp + 6 * cj);
    }
}

// LDS pitch of a 6 x 6 block: 37 doubles, not 36.  The lanes of a wave add the same entry of DIFFERENT blocks; at pitch 36 (288 B)
// those addresses fall on 8 of the 64 banks (SQ_LDS_BANK_CONFLICT was two thirds of SQ_LDS_IDX_ACTIVE), at 37 on all of them.
constexpr int kSchurPitch = 37;

__global__ __launch_bounds__(1024) void ba_schur_lds_kernel(BADev d, double *__restrict__ slabs, int slab_doubles, int rhs_exp)
{
    extern __shared__ __attribute__((aligned(16))) double sl[];   // [nblk * kSchurPitch] blocks, [n] rhs_corr, then [n] ints: qexp
    const int tid = threadIdx.x;
    const int n = 6 * d.n_cam;
    const int nblk = d.n_cam * (d.n_cam + 1) / 2;
    const int lds_doubles = nblk * kSchurPitch + n;
    for (int e = tid; e < lds_doubles; e += blockDim.x) sl[e] = 0.0;
    double *srhs = sl + (size_t)nblk * kSchurPitch;
    int *qe = reinterpret_cast<int *>(sl + lds_doubles);
    for (int e = tid; e < n; e += blockDim.x) qe[e] = d.qexp[e];
    __syncthreads();
    const size_t n_obs = d.n_obs;
    for (int i = blockIdx.x * blockDim.x + tid; i < d.n_obs; i += gridDim.x * blockDim.x) {
        const int p = d.obs_pt[i], ci = d.obs_cam[i];
        double W[18];
        load_W(d, n_obs, i, W);
        const double *Mi = d.Minv + 6 * (size_t)p;
        const double M[9] = {Mi[0], Mi[1], Mi[2], Mi[1], Mi[3], Mi[4], Mi[2], Mi[4], Mi[5]};
        const double ag0 = d.Aig[3 * (size_t)p], ag1 = d.Aig[3 * (size_t)p + 1], ag2 = d.Aig[3 * (size_t)p + 2];
        const int b = d.pt_start[p], T = d.pt_start[p + 1] - b, t = i - b;
        double Y[18];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double w0 = W[3 * a], w1 = W[3 * a + 1], w2 = W[3 * a + 2];
            const int ei = kFxBits - qe[6 * ci + a];
            fx_add(&srhs[6 * ci + a], -(w0 * ag0 + w1 * ag1 + w2 * ag2), ei - rhs_exp);
            Y[3 * a + 0] = ldexp(w0 * M[0] + w1 * M[3] + w2 * M[6], ei);
            Y[3 * a + 1] = ldexp(w0 * M[1] + w1 * M[4] + w2 * M[7], ei);
            Y[3 * a + 2] = ldexp(w0 * M[2] + w1 * M[5] + w2 * M[8], ei);
        }
        // its own block, then the partners at distance s = 1 .. T/2 around the track (cyclically): every unordered pair of the
        // point's observations is met exactly once and every observation does the same number of pairs (walking j over the
        // whole track and keeping c_j <= c_i did 8 iterations for 4.5 pairs on an 8-observation track)
        schur_pair<false>(sl + (size_t)(ci * (ci + 1) / 2 + ci) * kSchurPitch, 6, 1, Y, W, qe + 6 * ci);
        for (int s2 = 1; 2 * s2 <= T; ++s2) {
            if (2 * s2 == T && t < s2) continue;              // the half-way partners meet once
            int q = t - s2; if (q < 0) q += T;
            const int j = b + q;
            const int cj = d.obs_cam[j];
            double Wj[18];
            load_W(d, n_obs, j, Wj);
            const int hi = cj > ci ? cj : ci, lo = cj > ci ? ci : cj;
            double *Sb = sl + (size_t)(hi * (hi + 1) / 2 + lo) * kSchurPitch;
            if (cj != ci) schur_pair<false>(Sb, cj > ci ? 1 : 6, cj > ci ? 6 : 1, Y, Wj, qe + 6 * cj);
            else schur_pair<true>(Sb, 6, 1, Y, Wj, qe + 6 * cj);
        }
    }
    __syncthreads();
    double *out = slabs + (size_t)blockIdx.x * slab_doubles;
    for (int e = tid; e < slab_doubles; e += blockDim.x) out[e] = e < nblk * 36 ? sl[(e / 36) * kSchurPitch + e % 36] : sl[e + nblk * (kSchurPitch - 36)];
}

// Large camera counts: the Schur matrix does not fit LDS, but a workgroup that walks points ordered by their
// lowest camera only touches a narrow band of it.  Each workgroup owns one chunk of that point order and an
// LDS window of kWinCams consecutive cameras starting at the chunk's lowest one: contributions whose two
// cameras fall inside the window are accumulated with LDS f64 atomics, the few that do not (wide baselines,
// ring wrap-around) go straight to global f64 atomics; the window is flushed once at the end.
constexpr int kWinCams = kSchurWinCams;
constexpr int kWinBlocks = kWinCams * (kWinCams + 1) / 2;

// rot: camera indices are rotated by rot (mod n_real_cam) before the window test -- the second pass, rot = n_real_cam / 2, takes
// the tracks that straddle the seam of a closed camera loop (cameras 508 .. 511, 0 .. 5 of a ring of 512): in true indices they
// are 500 cameras wide and would all go to same-address global atomics (measured: 2.1 ms for 2 % of BA-512's observations).
__global__ __launch_bounds__(1024) void ba_schur_window_kernel(BADev d, int rhs_exp, const int32_t *__restrict__ slot_obs,
                                                               const int32_t *__restrict__ chunk_slot, const int32_t *__restrict__ chunk_cam0, int rot)
{
    extern __shared__ __attribute__((aligned(16))) double sl[];   // [kWinBlocks * kSchurPitch] blocks, then [6*kWinCams] rhs_corr
    const int tid = threadIdx.x;
    const int n = 6 * d.n_cam, nrc = d.n_real_cam;
    const int chunk = blockIdx.x;
    const int s0 = chunk_slot[chunk], s1 = chunk_slot[chunk + 1];
    const int cw = chunk_cam0[chunk];                               // window base, in ROTATED camera indices
    auto rotated = [&](int c) { const int r = c + rot; return r >= nrc ? r - nrc : r; };
    constexpr int kWinDoubles = kWinBlocks * kSchurPitch + 6 * kWinCams;
    for (int e = tid; e < kWinDoubles; e += 1024) sl[e] = 0.0;
    __syncthreads();
    double *srhs = sl + kWinBlocks * kSchurPitch;
    const size_t n_obs = d.n_obs;
    for (int s = s0 + tid; s < s1; s += 1024) {
        const int i = slot_obs[s];
        const int p = d.obs_pt[i], ci = d.obs_cam[i];
        double W[18];
        load_W(d, n_obs, i, W);
        const double *Mi = d.Minv + 6 * (size_t)p;
        const double M[9] = {Mi[0], Mi[1], Mi[2], Mi[1], Mi[3], Mi[4], Mi[2], Mi[4], Mi[5]};
        const double ag0 = d.Aig[3 * (size_t)p], ag1 = d.Aig[3 * (size_t)p + 1], ag2 = d.Aig[3 * (size_t)p + 2];
        const int b = d.pt_start[p], T = d.pt_start[p + 1] - b, t = i - b;
        const int wi = rotated(ci) - cw;
        const bool in_i = wi >= 0 && wi < kWinCams;
        double Y[18];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double w0 = W[3 * a], w1 = W[3 * a + 1], w2 = W[3 * a + 2];
            const int ei = kFxBits - d.qexp[6 * ci + a];
            const unsigned long long rv = fx64(-(w0 * ag0 + w1 * ag1 + w2 * ag2), ei - rhs_exp);
            atomicAdd(reinterpret_cast<unsigned long long *>(in_i ? &srhs[6 * wi + a] : &d.red[(size_t)n * n + 6 * ci + a]), rv);
            Y[3 * a + 0] = ldexp(w0 * M[0] + w1 * M[3] + w2 * M[6], ei);
            Y[3 * a + 1] = ldexp(w0 * M[1] + w1 * M[4] + w2 * M[7], ei);
            Y[3 * a + 2] = ldexp(w0 * M[2] + w1 * M[5] + w2 * M[8], ei);
        }
        // its own block, then the partners at distance 1 .. T/2 around the track (see ba_schur_lds_kernel).  A pair's block sits
        // in the LDS window when both cameras do; there it is oriented by the ROTATED indices (row = the larger one) and turned
        // the right way round when the window is flushed.
        if (in_i) schur_pair<false>(sl + (size_t)(wi * (wi + 1) / 2 + wi) * kSchurPitch, 6, 1, Y, W, d.qexp + 6 * ci);
        else schur_pair<false>(d.red + (size_t)(6 * ci) * n + 6 * ci, n, 1, Y, W, d.qexp + 6 * ci);
        for (int s2 = 1; 2 * s2 <= T; ++s2) {
            if (2 * s2 == T && t < s2) continue;              // the half-way partners meet once
            int q = t - s2; if (q < 0) q += T;
            const int j = b + q;
            const int cj = d.obs_cam[j];
            double Wj[18];
            load_W(d, n_obs, j, Wj);
            const int wj = rotated(cj) - cw;
            const bool in_w = in_i && wj >= 0 && wj < kWinCams;
            // (two call sites on purpose: with the address space known the window gets ds_add_u64; one merged pointer made every
            // add a flat atomic)
            if (in_w) {
                const int wh = wj > wi ? wj : wi, wl = wj > wi ? wi : wj;
                double *Sb = sl + (size_t)(wh * (wh + 1) / 2 + wl) * kSchurPitch;
                if (wj != wi) schur_pair<false>(Sb, wj > wi ? 1 : 6, wj > wi ? 6 : 1, Y, Wj, d.qexp + 6 * cj);
                else schur_pair<true>(Sb, 6, 1, Y, Wj, d.qexp + 6 * cj);
            } else {
                const int hi = cj > ci ? cj : ci, lo = cj > ci ? ci : cj;
                double *Sb = d.red + (size_t)(6 * hi) * n + 6 * lo;
                if (cj != ci) schur_pair<false>(Sb, cj > ci ? 1 : n, cj > ci ? n : 1, Y, Wj, d.qexp + 6 * cj);
                else schur_pair<true>(Sb, n, 1, Y, Wj, d.qexp + 6 * cj);
            }
        }
    }
    __syncthreads();
    // flush the window: rotated-local (wi >= wj) -> true cameras; the stored triangle wants row camera >= column camera
    const unsigned long long *sq = reinterpret_cast<const unsigned long long *>(sl);   // same grid per entry as d.red: integer adds
    auto true_cam = [&](int w) { int c = cw + w - rot; if (c < 0) c += nrc; return c; };
    for (int e = tid; e < kWinBlocks * 36; e += 1024) {
        const int blk = e / 36, r = e % 36;
        const unsigned long long v = sq[blk * kSchurPitch + r];
        if (v == 0ull) continue;
        int wi = 0, wj = blk;
        while (wj > wi) { wj -= wi + 1; ++wi; }
        const int ci = true_cam(wi), cj = true_cam(wj);
        if (ci >= nrc || cj >= nrc) continue;
        const size_t at = ci >= cj ? (size_t)(6 * ci + r / 6) * n + 6 * cj + r % 6 : (size_t)(6 * cj + r % 6) * n + 6 * ci + r / 6;
        atomicAdd(reinterpret_cast<unsigned long long *>(&d.red[at]), v);
    }
    for (int e = tid; e < 6 * kWinCams; e += 1024) {
        const unsigned long long v = sq[kWinBlocks * kSchurPitch + e];
        if (v == 0ull) continue;
        const int c = true_cam(e / 6);
        if (c < nrc) atomicAdd(reinterpret_cast<unsigned long long *>(&d.red[(size_t)n * n + 6 * c + e % 6]), v);
    }
}

// ---------------------------------------------------------------------------------------------
// Point-block Schur complement on the f64 matrix cores (round 3).  For one point p with observations i = 1..T,
//     sum over the pairs (i, j) of p of  -Y_i W_j'  =  -(Yhat_p)(What_p)',      Yhat_p = [Y_i]_i (6 T x 3),  Y_i = W_i M_p^-1,
// a rank-3 update of the rows / columns of p's cameras -- and the sum over the points of a chunk whose cameras lie in one window of
// kSchurMfCams indices is a dense product  G = sum_p Yw_p Ww_p'  (80 x 80, rows = 6 (camera - window base) + a), symmetric because
// M^-1 is.  ba_schur_window_kernel adds the 36 entries of every pair with 64-bit LDS atomics, and consecutive points see the same
// cameras: up to 64 lanes on one address, 0.93 ms per pass over BA-512 (25 x what conflict-free LDS adds would take).  Here a wave
// takes a batch of whole points (<= 64 observations, one per lane): the lanes form W_i and Y_i of their observation (the same loads
// and arithmetic as the atomic kernels) and park them in a wave-private LDS table; then point by point the 16-row operand
// fragments are gathered from that table (a lane's row belongs to one camera slot: looked up in the point's slot -> lane table,
// zero if the point does not see it) and the lower block triangle of G is accumulated with v_mfma_f64_16x16x4_f64 -- K = 4 = the
// three columns of a point and a zero -- in registers for the whole chunk.  No atomics until the end: the four waves' tiles are
// added in wave order through LDS and the chunk's G goes into d.red as fixed-point integers like every other contribution (one
// atomic per entry and chunk), so the result does not depend on how workgroups are scheduled.
// Points with a camera twice, or spanning more than kSchurMfCams indices in both numberings, stay with the atomic kernels.
// Measured on BA-512 (3 M observations, 512 chunks): 276 us for the plain numbering + 50 us for the seam's rotated table, against
// 856 + 45 us of the atomic window kernel; with the matrix products AND the flush compiled out the launch still takes ~180 us --
// what is left is the lanes' own loads and arithmetic (432 MB of Jacobian rows, the point blocks, 36 LDS stores per lane).  Issuing a
// batch's loads one batch ahead changed nothing (345 against 341 us, 47 spilled registers) and neither did a scene whose points are
// numbered by lowest camera, i.e. contiguous reads (465 against 487 us on a slower box: the kernel also varies 340 - 490 us box to box).
constexpr int kMfRows = 80, kMfBR = 5, kMfYPitch = 37, kMfMaxPts = 16;
constexpr int kMfWaveDoubles = 65 * kMfYPitch + kMfMaxPts;     // Y | W table (36 doubles per observation, pitch 37; row 64 = zeros) + per-point slot masks and first lanes (ints)
constexpr size_t kMfLdsBytes = sizeof(double) * (4 * kMfWaveDoubles + 6 * kSchurMfCams + kMfRows / 2);     // ... + right-hand side (u64) + the window's exponents (ints)
static_assert(4 * kMfWaveDoubles >= kMfRows * (kMfRows + 1), "the staging area is reused for the 80 x 81 result");
static_assert(kSchurMfCams <= 16 && 6 * kSchurMfCams <= kMfRows, "slot masks are 16 bits; the window's rows fit the tile grid");

#ifndef ESFM_SCHUR_OCC
#define ESFM_SCHUR_OCC 2
#endif
#ifdef ESFM_SCHUR_TRACE
// timing-only build (scratch/build_variant_ba.sh NAME -DESFM_SCHUR_TRACE): s_memrealtime ticks (10 ns) per stage, summed over the waves:
// [0] waiting for a batch's rows, [1] W / Y into LDS, [2] matrix products, [3] batches, [4] the workgroup's sum + flush, [5] workgroups
__device__ unsigned long long g_schur_trace[8];
extern "C" int esfm_debug_schur_trace(unsigned long long *out, int reset)
{
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_schur_trace), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_schur_trace), sizeof(g_schur_trace));
}
#define SCH_T0() unsigned long long sch_t = __builtin_amdgcn_s_memrealtime()
#define SCH_T(q) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); sch_acc[q] += t_ - sch_t; sch_t = t_; } while (0)
#else
#define SCH_T0() do { } while (0)
#define SCH_T(q) do { } while (0)
#endif
// Both tables in ONE launch (round 5): workgroups [0, n_plain) take the plain numbering's chunks, the rest the seam's (camera indices
// rotated by `rot_seam`).  The seam's table is 2 % of BA-512's observations in ~100 short chunks whose fixed costs -- zeroing 120
// accumulator registers, the 80 x 81 sum through LDS, 3 240 fixed-point atomics -- made a launch of their own last 47 us; behind the
// plain chunks they fill the launch's tail.
struct SchurMfTables {
    const int32_t *slot_obs[2]; const int2 *slot_pc[2]; const int32_t *batch_slot[2]; const int32_t *chunk_batch0[2]; const int32_t *chunk_cam0[2];
    int n_plain, rot_seam;
};
__global__ __launch_bounds__(256, ESFM_SCHUR_OCC) void ba_schur_mfma_kernel(BADev d, int rhs_exp, SchurMfTables tabs)
{
    const int tb = (int)blockIdx.x >= tabs.n_plain ? 1 : 0;
    const int32_t *__restrict__ slot_obs = tabs.slot_obs[tb];
    const int2 *__restrict__ slot_pc = tabs.slot_pc[tb];
    const int32_t *__restrict__ batch_slot = tabs.batch_slot[tb];
    const int32_t *__restrict__ chunk_batch0 = tabs.chunk_batch0[tb];
    const int32_t *__restrict__ chunk_cam0 = tabs.chunk_cam0[tb];
    const int rot = tb ? tabs.rot_seam : 0;
    typedef double doublex4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) double sl[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *Yw = sl + (size_t)wave * kMfWaveDoubles;                               // [65][kMfYPitch]; row 64 stays zero
    unsigned int *pmask = reinterpret_cast<unsigned int *>(Yw + 65 * kMfYPitch);   // [kMfMaxPts] camera slots the point is seen from (bit = slot)
    int *pfirst = reinterpret_cast<int *>(pmask) + kMfMaxPts;                      // [kMfMaxPts] lane of the point's first observation
    unsigned long long *srhs = reinterpret_cast<unsigned long long *>(sl + 4 * (size_t)kMfWaveDoubles);   // [6 * kSchurMfCams]
    int *s_qe = reinterpret_cast<int *>(srhs + 6 * kSchurMfCams);                  // [kMfRows] fixed-point exponents of the window's rows
    const int n = 6 * d.n_cam, nrc = d.n_real_cam;
    const size_t n_obs = d.n_obs;
    const int chunk = (int)blockIdx.x - (tb ? tabs.n_plain : 0);
    const int cw = chunk_cam0[chunk];
    const int b0 = chunk_batch0[chunk], b1 = chunk_batch0[chunk + 1];
    auto rotated = [&](int c) { const int r = c + rot; return r >= nrc ? r - nrc : r; };
    auto true_cam = [&](int w) { int c = cw + w - rot; if (c < 0) c += nrc; return c; };
    for (int e = tid; e < 6 * kSchurMfCams; e += 256) srhs[e] = 0ull;
    // the exponents the flush converts with: fetched now, read from LDS after the batches (round 5: the flush's 25 entries per thread
    // each waited for two dependent global loads -- 58 us per workgroup, a fifth of the launch)
    if (tid < kMfRows) { const int c = true_cam(tid / 6); s_qe[tid] = (tid < 6 * kSchurMfCams && c < nrc) ? d.qexp[6 * c + tid % 6] : 0; }
    if (lane < kMfYPitch) Yw[64 * kMfYPitch + lane] = 0.0;
    __syncthreads();
    doublex4 acc[kMfBR * (kMfBR + 1) / 2];
#pragma unroll
    for (int t = 0; t < kMfBR * (kMfBR + 1) / 2; ++t) acc[t] = doublex4{0.0, 0.0, 0.0, 0.0};
    const int r16 = lane & 15, m4 = lane >> 4;          // this lane's row inside a block row, and its K index (point column; 3 = zero)
    // What a lane gathers for block row R does not depend on the point: camera slot sc = row / 6 and entry 3 a + column of the
    // observation's 36-double record.  WHICH observation is the point's business: a point's observations sit in consecutive lanes in
    // ascending slot order (the host builds the tables that way), so the one with slot sc is lane  first + popcount(mask & ((1 << sc) - 1))
    // if bit sc of the point's slot mask is set, and the zero row otherwise -- two words per point instead of a slot -> lane table
    // whose 80 entries per point had to be cleared, written and looked up (round 5).
    int fr_sc[kMfBR], fr_off[kMfBR]; unsigned int fr_low[kMfBR];
#pragma unroll
    for (int R = 0; R < kMfBR; ++R) {
        const int row = 16 * R + r16, sc = row / 6, a = row - 6 * sc;
        fr_sc[R] = sc < kSchurMfCams ? sc : 31;                      // bit 31 of a slot mask is never set
        fr_off[R] = 3 * a;
        fr_low[R] = (1u << (sc < kSchurMfCams ? sc : 0)) - 1u;
    }
    // A lane's observation, point and camera come from slot-ordered tables (contiguous, one load level) and are fetched one batch
    // AHEAD: the Jacobian rows / point block / exponents they address can then be requested the moment a batch starts -- one
    // round trip to memory per batch where the chain  slot -> observation -> point, camera -> rows  made three (8 us per batch).
    int nx_i = 0, nx_nob = 0; int2 nx_pc = make_int2(0, 0);
    auto fetch_ids = [&](int b) {
        if (b >= b1) { nx_nob = 0; return; }
        const int s0 = batch_slot[b];
        nx_nob = batch_slot[b + 1] - s0;
        const int sl_ = s0 + (lane < nx_nob ? lane : 0);
        nx_i = slot_obs[sl_]; nx_pc = slot_pc[sl_];
    };
    // ... and the ROWS of a batch are requested while the batch before it is in its matrix products (round 5; they land in the same
    // registers the products do not need: nothing is held twice).  Before that a wave waited 2.9 us per batch for them -- a third
    // of a batch's time, at the two waves per SIMD the accumulators leave room for.
    double Jc[12], Jp[6], Mi[6], ag[3];
    int qe[6];
    auto fetch_rows = [&]() {           // of the batch whose ids fetch_ids left in nx_*
        const int i = nx_i, p = nx_pc.x, ci = nx_pc.y;
#pragma unroll
        for (int q = 0; q < 12; ++q) Jc[q] = d.Jc[q * n_obs + i];
#pragma unroll
        for (int q = 0; q < 6; ++q) Jp[q] = d.Jp[q * n_obs + i];
#pragma unroll
        for (int q = 0; q < 6; ++q) Mi[q] = d.Minv[6 * (size_t)p + q];
#pragma unroll
        for (int q = 0; q < 3; ++q) ag[q] = d.Aig[3 * (size_t)p + q];
#pragma unroll
        for (int a = 0; a < 6; ++a) qe[a] = d.qexp[6 * ci + a];
    };
    fetch_ids(b0 + wave);
    fetch_rows();
#ifdef ESFM_SCHUR_TRACE
    unsigned long long sch_acc[5] = {0, 0, 0, 0, 0};
#endif
    SCH_T0();
    for (int b = b0 + wave; b < b1; b += 4) {
        const int nob = nx_nob;
        const bool valid = lane < nob;
        const int p = nx_pc.x, ci = nx_pc.y;
        fetch_ids(b + 4);
        const int slot = rotated(ci) - cw;                                          // 0 .. kSchurMfCams - 1 by construction
        // local index of the lane's point inside the batch (the batch is whole points, in order)
        const int pprev = __shfl_up(p, 1);
        const bool first = valid && (lane == 0 || pprev != p);
        const unsigned long long starts = __ballot(first);
        const int q = __popcll(starts & ((2ull << lane) - 1ull)) - 1;
        const int npts = __popcll(starts);
        double W[18];                   // W_i = F_i'E_i (load_W's arithmetic)
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int m = 0; m < 3; ++m) W[3 * a + m] = Jc[a] * Jp[m] + Jc[6 + a] * Jp[3 + m];
        const double M[9] = {Mi[0], Mi[1], Mi[2], Mi[1], Mi[3], Mi[4], Mi[2], Mi[4], Mi[5]};
        const double ag0 = ag[0], ag1 = ag[1], ag2 = ag[2];
        if (lane < kMfMaxPts) pmask[lane] = 0u;
#ifdef ESFM_SCHUR_TRACE
        __builtin_amdgcn_s_waitcnt(0x0f70);
        SCH_T(0);
#endif
        if (valid) {
            atomicOr(&pmask[q], 1u << slot);
            if (first) pfirst[q] = lane;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const double w0 = W[3 * a], w1 = W[3 * a + 1], w2 = W[3 * a + 2];
                const int ei = kFxBits - qe[a];
                atomicAdd(&srhs[6 * slot + a], fx64(-(w0 * ag0 + w1 * ag1 + w2 * ag2), ei - rhs_exp));
                double *y = Yw + lane * kMfYPitch + 3 * a;
                y[0] = w0 * M[0] + w1 * M[3] + w2 * M[6];
                y[1] = w0 * M[1] + w1 * M[4] + w2 * M[7];
                y[2] = w0 * M[2] + w1 * M[5] + w2 * M[8];
                y[18] = w0; y[19] = w1; y[20] = w2;
            }
        }
        // block rows the batch touches (wave-uniform)
        int smin = valid ? slot : kSchurMfCams, smax = valid ? slot : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { smin = min(smin, __shfl_xor(smin, o)); smax = max(smax, __shfl_xor(smax, o)); }
        const int R0 = __builtin_amdgcn_readfirstlane((6 * smin) / 16), R1 = __builtin_amdgcn_readfirstlane((6 * smax + 5) / 16);
        // The wave's own LDS writes are visible to its later reads (DS operations of a wave execute in order); the fence pair says so
        // to the COMPILER -- other lanes' stores above, a data-dependent gather below: without it nothing forbids hoisting the loads
        // (no instruction is emitted for a wavefront-scope fence).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        fetch_rows();                   // the next batch's (its ids were requested at the top; clamped to lane 0's slot past the end)
#ifdef ESFM_SCHUR_TRACE
        __builtin_amdgcn_s_waitcnt(0xc07f);
        SCH_T(1);
#endif
        // The K = 4 slices of one matrix instruction are four consecutive (point, column) pairs of the batch -- slice 4 t + k is column
        // (4 t + k) % 3 of point (4 t + k) / 3 -- not "three columns of one point and a zero": 3/4 of the instructions, and the matrix
        // pipe is what this loop waits for (a 16 x 16 x 4 f64 product occupies it for 64 cycles on this part -- its f64 matrix rate
        // equals its f64 vector rate -- and a ten-camera point touched 10 - 15 tiles: 2.6 us per batch of 6.4 points, 5.4 us per batch
        // and wave measured with two waves per SIMD).  A lane's slice decides its point: the point's mask and first lane come from
        // LDS by a per-lane address.  No branch: a slot the point does not have -- any slot of a block row the batch does not touch
        // is one -- and a slice past the batch's last point read the zero row.
        // (A second set of fragment registers to fetch the next step's rows under this step's products spilled at two waves per SIMD
        // and lost at one: measured, not kept.)
        const int nsteps = (3 * npts + 3) >> 2;
        for (int t = 0; t < nsteps; ++t) {
            const int sigma = 4 * t + m4;
            const int pq = (sigma * 43) >> 7;                    // sigma / 3 for sigma < 128
            const int col = sigma - 3 * pq;
            const bool live = pq < npts;
            const unsigned int pm = live ? pmask[pq & (kMfMaxPts - 1)] : 0u;
            const int pf = pfirst[pq & (kMfMaxPts - 1)];
            double fa[kMfBR], fb[kMfBR];
#pragma unroll
            for (int R = 0; R < kMfBR; ++R) {
                const int have = (int)((pm >> fr_sc[R]) & 1u);
                const int o = 64 + have * (pf + (int)__popc(pm & fr_low[R]) - 64);
                const double *y = Yw + o * kMfYPitch + fr_off[R] + col;
                fa[R] = y[0]; fb[R] = y[18];
            }
#pragma unroll
            for (int R = 0; R < kMfBR; ++R) {
                if (R < R0 || R > R1) continue;
#pragma unroll
                for (int C = 0; C <= R; ++C) {
                    if (C < R0) continue;
                    acc[R * (R + 1) / 2 + C] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[R], fb[C], acc[R * (R + 1) / 2 + C], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the table has been read before the next batch overwrites it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (... and the compiler may not sink this batch's reads below the next one's stores)
        __builtin_amdgcn_wave_barrier();
#ifdef ESFM_SCHUR_TRACE
        SCH_T(2); sch_acc[3] += 1;
#endif
    }
    // the chunk's G: the waves' tiles added in wave order (fixed), lower block triangle, in the staging area
    __syncthreads();
    double *G = sl;                                // [kMfRows][kMfRows + 1]
    for (int e = tid; e < kMfRows * (kMfRows + 1); e += 256) G[e] = 0.0;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int R = 0; R < kMfBR; ++R)
#pragma unroll
                for (int C = 0; C <= R; ++C)
#pragma unroll
                    for (int g = 0; g < 4; ++g) G[(16 * R + m4 + 4 * g) * (kMfRows + 1) + 16 * C + r16] += acc[R * (R + 1) / 2 + C][g];
        }
        __syncthreads();
    }
    // S -= G on the entry's fixed-point grid.  Window order (row >= col) is not camera order once the indices are rotated: the stored
    // triangle wants row camera >= column camera in TRUE indices, a diagonal camera block in full.
    unsigned long long *redq = reinterpret_cast<unsigned long long *>(d.red);
    for (int e = tid; e < kMfRows * kMfRows; e += 256) {
        const int row = e / kMfRows, col = e - row * kMfRows;
        if (col > row) continue;
        const int si = row / 6, a = row - 6 * si, sj = col / 6, b2 = col - 6 * sj;
        if (si >= kSchurMfCams) continue;
        const double v = G[row * (kMfRows + 1) + col];
        if (v == 0.0) continue;
        const int ci = true_cam(si), cj = true_cam(sj);
        if (ci >= nrc || cj >= nrc) continue;
        const int ri = 6 * ci + a, rj = 6 * cj + b2;
        const unsigned long long qv = fx64(-v, kFxBits - s_qe[row] - s_qe[col]);
        if (ci > cj) atomicAdd(&redq[(size_t)ri * n + rj], qv);
        else if (ci < cj) atomicAdd(&redq[(size_t)rj * n + ri], qv);
        else { atomicAdd(&redq[(size_t)ri * n + rj], qv); if (a != b2) atomicAdd(&redq[(size_t)rj * n + ri], qv); }
    }
    for (int e = tid; e < 6 * kSchurMfCams; e += 256) {
        const unsigned long long v = srhs[e];
        if (v == 0ull) continue;
        const int c = true_cam(e / 6);
        if (c < nrc) atomicAdd(&redq[(size_t)n * n + 6 * c + e % 6], v);
    }
#ifdef ESFM_SCHUR_TRACE
    SCH_T(4);
    if (lane == 0) {
        for (int q = 0; q < 5; ++q) atomicAdd(&g_schur_trace[q], sch_acc[q]);
        if (wave == 0) atomicAdd(&g_schur_trace[5], 1ull);
    }
#endif
}

// Free intrinsics: the block row of the reduced system that belongs to fx, cx, fy, cy (block index n_real_cam).
// One thread per point p.  With G_i the intrinsics columns of observation i (2 x 4, two non-zeros per row),
//   Wk_p = sum_i G_i'E_i (4x3), Yk = Wk_p M^-1:
//   S[k][k]   -= Yk Wk_p'            rhs[k] -= Wk_p M^-1 E'r
//   S[k][c_j] -= Yk W_j'  + G_j'F_j  for every observation j of p  (the second term is the F'F cross block)
// The camera-indexed part goes through an LDS copy of the block row (4 x 6 n_real_cam) when it fits, the 14 sums
// every point shares through a workgroup reduction; both are then added into d.red.  Runs after ba_schur.
__global__ __launch_bounds__(256) void ba_schur_calib_kernel(BADev d, int use_lds, int rhs_exp)
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
                    const int ej = d.qexp[6 * c + c2];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const double v = (a < 2 ? kk[a] * f0 : kk[a] * f1) - (Yk[a][0] * w0 + Yk[a][1] * w1 + Yk[a][2] * w2);
                        fx_add(use_lds ? &rowblk[a * roww + 6 * c + c2] : &d.red[(size_t)(kap + a) * n + 6 * c + c2], v, kFxBits - d.qexp[kap + a] - ej);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 14; ++q) {
        const double t = block_sum(acc[q], red);   // the workgroup's sum (fixed tree), rounded to the entry's grid, then exact adds
        if (tid != 0 || t == 0.0) continue;
        if (q >= 10) { fx_add(&d.red[(size_t)n * n + kap + (q - 10)], t, kFxBits - d.qexp[kap + (q - 10)] - rhs_exp); continue; }
        int a = 0, rem = q;
        while (rem > a) { rem -= a + 1; ++a; }
        fx_add(&d.red[(size_t)(kap + a) * n + kap + rem], t, kFxBits - d.qexp[kap + a] - d.qexp[kap + rem]);
    }
    if (use_lds) {
        __syncthreads();
        for (int e = tid; e < 4 * roww; e += 256) {
            const unsigned long long v = reinterpret_cast<const unsigned long long *>(rowblk)[e];
            if (v != 0ull) atomicAdd(reinterpret_cast<unsigned long long *>(&d.red[(size_t)(kap + e / roww) * n + (e % roww)]), v);
        }
    }
}

// Sum the per-workgroup slabs (fixed-point integers: exact in any order) and scatter into red = S_schur (n x n) | rhs_corr (n);
// TO_DOUBLE: converted to f64 on the way (no free intrinsics), else left in fixed point for ba_schur_calib_kernel to add to.
template <bool TO_DOUBLE>
__global__ __launch_bounds__(256) void ba_schur_reduce_kernel(BADev d, const double *__restrict__ slabs_f, int slab_doubles, int n_slabs, int rhs_exp)
{
    __shared__ unsigned long long lds[256];
    const unsigned long long *slabs = reinterpret_cast<const unsigned long long *>(slabs_f);
    const int e = blockIdx.x * kRedEnt + (threadIdx.x % kRedEnt);
    const int grp = threadIdx.x / kRedEnt;
    unsigned long long v0 = 0;
    if (e < slab_doubles) {
        for (int b = grp; b < n_slabs; b += 8 * kRedGrp) {      // eight loads in flight per round
            unsigned long long t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = (b + u * kRedGrp < n_slabs) ? slabs[(size_t)(b + u * kRedGrp) * slab_doubles + e] : 0ull;
#pragma unroll
            for (int u = 0; u < 8; ++u) v0 += t[u];
        }
    }
    lds[threadIdx.x] = v0;
    __syncthreads();
    if (threadIdx.x >= kRedEnt || e >= slab_doubles) return;
    unsigned long long t = 0;
    for (int g = 0; g < kRedGrp; ++g) t += lds[g * kRedEnt + threadIdx.x];
    const int n = 6 * d.n_cam;
    const int nblk = d.n_cam * (d.n_cam + 1) / 2;
    size_t dst; int sh;
    if (e >= nblk * 36) {
        const int r = e - nblk * 36;
        dst = (size_t)n * n + r; sh = kFxBits - d.qexp[r] - rhs_exp;
    } else {
        const int blk = e / 36, r = e % 36;
        int ci = (int)((sqrt(8.0 * (double)blk + 1.0) - 1.0) * 0.5);
        while ((ci + 1) * (ci + 2) / 2 <= blk) ++ci;
        while (ci * (ci + 1) / 2 > blk) --ci;
        const int cj = blk - ci * (ci + 1) / 2;
        const int row = 6 * ci + r / 6, col = 6 * cj + r % 6;
        dst = (size_t)row * n + col; sh = kFxBits - d.qexp[row] - d.qexp[col];
    }
    if (TO_DOUBLE) d.red[dst] = fx64_to_double(t, sh);
    else reinterpret_cast<unsigned long long *>(d.red)[dst] = t;
}

// red: fixed point -> f64 in place, block-lower triangle and right-hand side (everything else is zero in both encodings).
__global__ __launch_bounds__(256) void ba_schur_to_double_kernel(BADev d, int rhs_exp)
{
    const int n = 6 * d.n_cam;
    const int row = blockIdx.y;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (row == n) {
        if (col < n) d.red[(size_t)n * n + col] = fx64_to_double(reinterpret_cast<const unsigned long long *>(d.red)[(size_t)n * n + col], kFxBits - d.qexp[col] - rhs_exp);
        return;
    }
    if (col >= 6 * (row / 6 + 1)) return;
    const size_t at = (size_t)row * n + col;
    const unsigned long long q = reinterpret_cast<const unsigned long long *>(d.red)[at];
    if (q != 0ull) d.red[at] = fx64_to_double(q, kFxBits - d.qexp[row] - d.qexp[col]);
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
// Candidate cameras: x + (-y) .* scaling for cameras that have observations; step/candidate norms (ba_camera_step_body in
// ba_kernels.hpp; the one-workgroup reduced solve calls it at its end instead of this launch).
__global__ __launch_bounds__(256) void ba_camera_step_kernel(BADev d)
{
    __shared__ double red[48];
    ba_camera_step_body(d, d.y_c, red);
}

// Back-substitution: y_p = M^-1 (E'r - sum_k E_k'F_k y_c), step = -y, candidate point, and the model cost change
// -sum (J s).(r + J s / 2)  (trust_region_minimizer.cc).  One workgroup per point chunk (pchunk_pt0: consecutive points whose
// observations -- contiguous in the point-sorted order -- number at most kPtChunkObs = 256, built once per problem):
//   1. thread = observation: one coalesced read of its Jacobian rows (kept in registers), f = F_k y_c, t_k = E_k' f -> LDS;
//   2. thread = point: adds its t_k in observation order, s_p = -M^-1 (E'r - sum t_k), candidate point, s_p -> LDS;
//   3. thread = observation again: J s = -f + E_k s_p  (-(a + b) == (-a) + (-b) exactly), its share of the scalars.
// No atomics, every sum in a fixed order.  (Round 1 had one thread per point walking its observations: 16 dependent memory round
// trips on an 8-observation track, 43 us on the 25-camera problem; and, from 2^20 observations, two observation-parallel passes
// joined by f64 atomics.)  A single point with more observations than a chunk holds gets a chunk of its own and loops.

// 1/2 rho(|r|^2) of observation k at the candidate (d.cand_c, point X): what ba_cost_kernel<false> evaluates, here inside the
// back-substitution that has just produced X (WITH_COST: unbounded problems, where no slope along the step is needed)
__device__ __forceinline__ void candidate_cost(const BADev &d, int k, const double (&X)[3], double cauchy_a, double &cost, double &bad)
{
    const int c = d.obs_cam[k];
    const float2 uv = d.obs_uv[k];
    double in4[4];
    load_intrinsics(d, d.cand_c, c, in4);
    double cam[6], pt[3];
#pragma unroll
    for (int q = 0; q < 6; ++q) cam[q] = d.cand_c[6 * (size_t)c + q];
    transform_point<false>(cam, X, pt, nullptr, nullptr);
    const double x = pt[0] / pt[2], y = pt[1] / pt[2];
    const double r0 = (double)uv.x - (x * in4[0] + in4[1]), r1 = (double)uv.y - (y * in4[2] + in4[3]);
    const double s2 = r0 * r0 + r1 * r1;
    if (!isfinite(s2)) { bad += 1.0; return; }
    double rho0, rho1;
    loss_eval(cauchy_a, s2, rho0, rho1);
    cost += 0.5 * rho0;
}

// (register budget: four workgroups per CU.  The WITH_COST form had been compiled to 198 registers -- two waves per SIMD for a kernel
// that streams; at 128 BA-25's iteration went 0.136 -> 0.133 ms.  The same audit found nothing else: the sweep at three / four
// workgroups per CU spills, 25.7 / 36 us against 22.5; the f64-MFMA Schur kernel at one / three: 462 / 640 us against 294.)
#ifndef ESFM_BACKSUB_OCC
#define ESFM_BACKSUB_OCC 4
#endif
template <bool WITH_COST>
__global__ __launch_bounds__(kPtChunkObs, ESFM_BACKSUB_OCC) void ba_backsub_chunk_kernel(BADev d, ScalBase sbase, double cauchy_a)
{
    __shared__ double red[8];
    __shared__ double tE[kPtChunkObs][3];
    __shared__ double sps[kPtChunkObs][3];
    __shared__ double cnd[kPtChunkObs][3];      // WITH_COST: the candidate points of the chunk
    const int tid = threadIdx.x;
    const int4 ci4 = d.pchunk_info[blockIdx.x];          // (one load where p0 / p1, then pt_start[p0] / pt_start[p1], were two dependent ones)
    const int p0 = ci4.x, p1 = ci4.y, k0 = ci4.z, k1 = ci4.w;
    const int nobs = k1 - k0;
    const size_t n = d.n_obs;
    double mc = 0.0, ssq = 0.0, csq = 0.0, gd = 0.0, dmax = 0.0, ccost = 0.0, cbad = 0.0;
    double yk[4] = {0.0, 0.0, 0.0, 0.0};
    if (d.has_calib) {
#pragma unroll
        for (int a = 0; a < 4; ++a) yk[a] = d.y_c[6 * d.n_real_cam + a];
    }
    auto point_step_pre = [&](int p, const double g[3], const double (&Mi)[6], const double (&xq)[3], const double (&scq)[3]) {   // the same from operands already in registers
        double sp[3] = {-(Mi[0] * g[0] + Mi[1] * g[1] + Mi[2] * g[2]), -(Mi[1] * g[0] + Mi[3] * g[1] + Mi[4] * g[2]),
                        -(Mi[2] * g[0] + Mi[4] * g[1] + Mi[5] * g[2])};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double xp = xq[a];
            const double dl = sp[a] * scq[a];
            const double cnd = xp + dl;
            const double df = xp - cnd;
            ssq += df * df; csq += cnd * cnd;
            d.cand_p[3 * (size_t)p + a] = cnd;
            if (d.constrained) { d.delta_p[3 * (size_t)p + a] = dl; dmax = fmax(dmax, fabs(dl)); }
        }
        return std::array<double, 3>{sp[0], sp[1], sp[2]};
    };
    auto point_step = [&](int p, const double g[3]) {       // s_p, candidate point, norms
        const double *Mi = d.Minv + 6 * (size_t)p;
        double sp[3] = {-(Mi[0] * g[0] + Mi[1] * g[1] + Mi[2] * g[2]), -(Mi[1] * g[0] + Mi[3] * g[1] + Mi[4] * g[2]),
                        -(Mi[2] * g[0] + Mi[4] * g[1] + Mi[5] * g[2])};
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
        return std::array<double, 3>{sp[0], sp[1], sp[2]};
    };
    if (nobs <= kPtChunkObs) {
        const bool has = tid < nobs;
        const int k = k0 + (has ? tid : 0);
        double Jp[6], f0 = 0.0, f1 = 0.0, r0 = 0.0, r1 = 0.0;
        int pl = 0;
        // the point phase's operands are requested in front of the rows (see ba_point_prep_chunk_kernel)
        const bool is_pt = tid < p1 - p0;
        int pt_b = 0, pt_e = 0;
        double pt_g[3] = {0.0, 0.0, 0.0}, pt_M[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, pt_x[3] = {0.0, 0.0, 0.0}, pt_sc[3] = {0.0, 0.0, 0.0};
        if (is_pt) {
            const size_t p = (size_t)(p0 + tid);
            pt_b = d.pt_start[p] - k0; pt_e = d.pt_start[p + 1] - k0;
#pragma unroll
            for (int a = 0; a < 3; ++a) { pt_g[a] = d.Etr[3 * p + a]; pt_x[a] = d.x_p[3 * p + a]; pt_sc[a] = d.scale_p[3 * p + a]; }
#pragma unroll
            for (int a = 0; a < 6; ++a) pt_M[a] = d.Minv[6 * p + a];
        }
        if (has) {
            const int c = d.obs_cam[k];
            pl = d.obs_pt[k] - p0;
            double Jc[12], yc[6];
#pragma unroll
            for (int a = 0; a < 12; ++a) Jc[a] = d.Jc[a * n + k];
#pragma unroll
            for (int a = 0; a < 6; ++a) { Jp[a] = d.Jp[a * n + k]; yc[a] = d.y_c[6 * c + a]; }
            r0 = d.res[k]; r1 = d.res[n + k];
#pragma unroll
            for (int a = 0; a < 6; ++a) { f0 += Jc[a] * yc[a]; f1 += Jc[6 + a] * yc[a]; }
            if (d.has_calib) {
                f0 += d.Jk[k] * yk[0] + d.Jk[n + k] * yk[1];
                f1 += d.Jk[2 * n + k] * yk[2] + d.Jk[3 * n + k] * yk[3];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) tE[tid][a] = Jp[a] * f0 + Jp[3 + a] * f1;
        }
        __syncthreads();
        if (is_pt) {
            const int p = p0 + tid;
            const int b = pt_b, e = pt_e;
            if (e > b) {
                double g[3] = {pt_g[0], pt_g[1], pt_g[2]};
                for (int o = b; o < e; ++o) { g[0] -= tE[o][0]; g[1] -= tE[o][1]; g[2] -= tE[o][2]; }
                const auto sp = point_step_pre(p, g, pt_M, pt_x, pt_sc);
                sps[tid][0] = sp[0]; sps[tid][1] = sp[1]; sps[tid][2] = sp[2];
                if (WITH_COST) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) cnd[tid][a] = d.cand_p[3 * (size_t)p + a];      // (this thread's own stores)
                }
            } else {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    d.cand_p[3 * (size_t)p + a] = pt_x[a];
                    if (d.constrained) d.delta_p[3 * (size_t)p + a] = 0.0;
                }
            }
        }
        __syncthreads();
        if (has) {
            double m0 = -f0, m1 = -f1;
#pragma unroll
            for (int a = 0; a < 3; ++a) { m0 += Jp[a] * sps[pl][a]; m1 += Jp[3 + a] * sps[pl][a]; }
            mc = -(m0 * (r0 + m0 / 2.0) + m1 * (r1 + m1 / 2.0));
            gd = m0 * r0 + m1 * r1;
            if (WITH_COST) {
                const double X[3] = {cnd[pl][0], cnd[pl][1], cnd[pl][2]};
                candidate_cost(d, k, X, cauchy_a, ccost, cbad);
            }
        }
    } else {
        // one point, more observations than threads: strided passes, block-wide sums in a fixed tree
        const int p = p0;
        double t[3] = {0.0, 0.0, 0.0};
        for (int k = k0 + tid; k < k1; k += kPtChunkObs) {
            const int c = d.obs_cam[k];
            double f0 = 0.0, f1 = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) { const double yc = d.y_c[6 * c + a]; f0 += d.Jc[a * n + k] * yc; f1 += d.Jc[(6 + a) * n + k] * yc; }
            if (d.has_calib) {
                f0 += d.Jk[k] * yk[0] + d.Jk[n + k] * yk[1];
                f1 += d.Jk[2 * n + k] * yk[2] + d.Jk[3 * n + k] * yk[3];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) t[a] += d.Jp[a * n + k] * f0 + d.Jp[(3 + a) * n + k] * f1;
        }
        double g[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] = d.Etr[3 * (size_t)p + a] - block_sum(t[a], red);
        if (tid == 0) {
            const auto sp = point_step(p, g);
            sps[0][0] = sp[0]; sps[0][1] = sp[1]; sps[0][2] = sp[2];
#pragma unroll
            for (int a = 0; a < 3; ++a) cnd[0][a] = d.cand_p[3 * (size_t)p + a];
        }
        __syncthreads();
        for (int k = k0 + tid; k < k1; k += kPtChunkObs) {
            const int c = d.obs_cam[k];
            double m0 = 0.0, m1 = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) { const double sc = -d.y_c[6 * c + a]; m0 += d.Jc[a * n + k] * sc; m1 += d.Jc[(6 + a) * n + k] * sc; }
            if (d.has_calib) {
                m0 -= d.Jk[k] * yk[0] + d.Jk[n + k] * yk[1];
                m1 -= d.Jk[2 * n + k] * yk[2] + d.Jk[3 * n + k] * yk[3];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) { m0 += d.Jp[a * n + k] * sps[0][a]; m1 += d.Jp[(3 + a) * n + k] * sps[0][a]; }
            const double r0 = d.res[k], r1 = d.res[n + k];
            mc -= m0 * (r0 + m0 / 2.0) + m1 * (r1 + m1 / 2.0);
            gd += m0 * r0 + m1 * r1;
            if (WITH_COST) {
                const double X[3] = {cnd[0][0], cnd[0][1], cnd[0][2]};
                candidate_cost(d, k, X, cauchy_a, ccost, cbad);
            }
        }
    }
    if (WITH_COST) {
        const int slots[6] = {SC_MODEL_CHANGE, SC_STEP_SQ_PT, SC_CAND_SQ_PT, SC_GDOTD, SC_CAND_COST, SC_CAND_BAD};
        const double vals[6] = {mc, ssq, csq, d.constrained ? gd : 0.0, ccost, cbad};
        scal_commit<6>(d, sbase, slots, vals, red);
    } else {
        const int slots[4] = {SC_MODEL_CHANGE, SC_STEP_SQ_PT, SC_CAND_SQ_PT, SC_GDOTD};
        const double vals[4] = {mc, ssq, csq, d.constrained ? gd : 0.0};
        scal_commit<4>(d, sbase, slots, vals, red);
    }
    if (d.constrained) {
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
                                                      double cauchy_a, int slot, int bad_slot, ScalBase sbase)
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
    if (SLOPE) {
        const int slots[3] = {slot, bad_slot, SC_LS_GRAD};
        const double vals[3] = {cost, bad, slope};
        scal_commit<3>(d, sbase, slots, vals, red);
    } else {
        const int slots[2] = {slot, bad_slot};
        const double vals[2] = {cost, bad};
        scal_commit<2>(d, sbase, slots, vals, red);
    }
}

// candidate = Plus(x, t delta): cameras (projected onto the box) by workgroup 0, points by all; step / candidate norms.
__global__ __launch_bounds__(256) void ba_take_step_kernel(BADev d, double t, ScalBase sbase)
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
    const double c = block_sum(ssc, red);
    const double e = block_sum(csc, red);
    if (threadIdx.x == 0 && blockIdx.x == 0) { d.scal[SC_STEP_SQ_CAM] = c; d.scal[SC_CAND_SQ_CAM] = e; }
    const int slots[2] = {SC_STEP_SQ_PT, SC_CAND_SQ_PT};
    const double vals[2] = {ssq, csq};
    scal_commit<2>(d, sbase, slots, vals, red);
}

// x_c <- projection onto the box (TrustRegionMinimizer::IterationZero: Plus(x, 0))
__global__ void ba_project_cameras_kernel(BADev d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * d.n_cam && d.cam_nobs[i / 6] > 0.0) d.x_c[i] = fmin(fmax(d.x_c[i], d.lo_c[i]), d.up_c[i]);
}

// |x|^2 over the parameter blocks that take part in the problem (Ceres drops unused blocks).
__global__ __launch_bounds__(256) void ba_param_sqnorm_kernel(BADev d, ScalBase sbase)
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
    const double b = block_sum(sc, red);
    if (threadIdx.x == 0 && blockIdx.x == 0) d.scal[SC_XNORM_SQ_CAM] = b;
    const int slots[1] = {SC_XNORM_SQ_PT};
    const double vals[1] = {sp};
    scal_commit<1>(d, sbase, slots, vals, red);
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

int ba_linearize(hipStream_t st, const BADev &d, int num_cu, double cauchy_a, bool use_scaling, esfm_ctx *timing_ctx, int *deferred_slabs, double cost_bound)
{
    // provisional exponent of the residual column (see the kernel): sqrt(sum r^2) <= sqrt(2 cost) < 2^prov_rexp
    int prov_rexp = INT_MIN;
    if (cost_bound > 0.0 && cost_bound < 1e300) prov_rexp = std::ilogb(std::sqrt(2.0 * cost_bound)) + 1;
    if (deferred_slabs) *deferred_slabs = 0;
    if (d.n_obs <= 0 || d.n_cchunks <= 0) {
        ESFM_HIP_TRY(hipMemsetAsync(d.camacc, 0, sizeof(double) * ba_camacc_doubles(d.n_cam), st));
        ESFM_HIP_TRY(hipMemsetAsync(d.qexp, 0, sizeof(int32_t) * 6 * (size_t)d.n_cam, st));
        return ESFM_OK;
    }
    const size_t priv_bytes = 18 * sizeof(double) + (size_t)d.n_real_cam * kLinLdsPerCam;
    const bool small = d.n_obs < kLinSmallObs;
    const int kLinThreads = small ? kLinThreadsSmall : kLinThreadsLarge;
    // (the 512-thread form fills a CU's registers with ONE workgroup -- 250 registers x 8 waves -- so a second workgroup per CU only runs
    // after the first, paying the 9-us prologue (LDS clear, exponents) and the 8-us epilogue (guard, slab) again: one per CU)
    static const int grid_mul = getenv("ESFM_LIN_GRID_MUL") ? atoi(getenv("ESFM_LIN_GRID_MUL")) : 0;       // developer switch (A/B)
    const int grid = std::min(div_up(d.n_obs, kLinThreads), std::max(1, num_cu) * (grid_mul > 0 ? grid_mul : (small ? 2 : 1)));
    const bool priv = priv_bytes <= 160 * 1024 && d.lin_slabs && (size_t)grid * d.n_cam * 27 <= d.lin_slab_cap;
    ScalBase sbase;
    { const int slots[2] = {SC_COST, SC_LIN_BAD}; if (int rc = scal_reserve<2>(st, d, slots, grid, sbase)) return rc; }
    if (priv) {
        auto kern = small ? (d.has_calib ? &ba_linearize_kernel<true, true, kLinThreadsSmall> : &ba_linearize_kernel<true, false, kLinThreadsSmall>)
                          : (d.has_calib ? &ba_linearize_kernel<true, true, kLinThreadsLarge> : &ba_linearize_kernel<true, false, kLinThreadsLarge>);
        ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)priv_bytes));
        {
            KernelTimer tm(timing_ctx, ESFM_K_BA_LINEARIZE);   // the Jacobian sweep with its in-LDS camera sums (not the slab reduction)
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kLinThreads), priv_bytes, st, d, cauchy_a, use_scaling ? 1 : 0, sbase, d.lin_slabs, prov_rexp);
        }
        LAUNCH_CHECK();
        d.parts->grad_done = d.parts->single_rank;
        // one rank, per-point blocks by the walking kernel next: the slab reduction rides in that launch (ba_point_prep_camacc)
        if (deferred_slabs && d.parts->single_rank && !(d.n_pchunks > 0 && d.n_obs >= (1 << 20)) && d.n_pt > 0) { *deferred_slabs = grid; return ESFM_OK; }
        hipLaunchKernelGGL(ba_camacc_reduce_kernel, dim3(div_up(d.n_cam * 27, kRedEnt)), dim3(256), 0, st, d, d.lin_slabs, grid, d.parts->single_rank ? 1 : 0);
        LAUNCH_CHECK();
        return ESFM_OK;
    }
    {
        KernelTimer tm(timing_ctx, ESFM_K_BA_LINEARIZE);   // the Jacobian sweep alone
        auto kern = small ? (d.has_calib ? &ba_linearize_kernel<false, true, kLinThreadsSmall> : &ba_linearize_kernel<false, false, kLinThreadsSmall>)
                          : (d.has_calib ? &ba_linearize_kernel<false, true, kLinThreadsLarge> : &ba_linearize_kernel<false, false, kLinThreadsLarge>);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kLinThreads), 18 * sizeof(double), st, d, cauchy_a, use_scaling ? 1 : 0, sbase, (double *)nullptr, INT_MIN);
    }
    LAUNCH_CHECK();
    if (d.has_calib) hipLaunchKernelGGL(ba_camacc_chunk_kernel<true>, dim3(d.n_cchunks), dim3(64), 0, st, d);
    else hipLaunchKernelGGL(ba_camacc_chunk_kernel<false>, dim3(d.n_cchunks), dim3(64), 0, st, d);
    hipLaunchKernelGGL(ba_camacc_final_kernel, dim3(div_up(d.n_real_cam * 27 + 42, 256)), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_point_prep(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag, bool fresh, int deferred_slabs)
{
    ScalBase sbase;
    const int slots[1] = {SC_PT_SINGULAR};
    if (deferred_slabs > 0) {      // ba_linearize left the slab reduction to this launch (it checked that this path is taken)
        const int n_red = div_up(d.n_cam * 27, kRedEnt), n_prep = div_up(d.n_pt, 256);
        if (int rc = scal_reserve<1>(st, d, slots, n_prep, sbase)) return rc;
        hipLaunchKernelGGL(ba_point_prep_camacc_kernel, dim3(n_red + n_prep), dim3(256), 0, st, d, radius, min_diag, max_diag, sbase, d.lin_slabs,
                           deferred_slabs, 1, n_red);
        LAUNCH_CHECK();
        return ESFM_OK;
    }
    if (d.n_pt <= 0) return ESFM_OK;
    if (fresh && d.n_pchunks > 0 && d.n_obs >= (1 << 20)) {        // (see ba_point_prep_chunk_kernel for the crossover)
        if (int rc = scal_reserve<1>(st, d, slots, d.n_pchunks, sbase)) return rc;
        hipLaunchKernelGGL(ba_point_prep_chunk_kernel, dim3(d.n_pchunks), dim3(kPtChunkObs), 0, st, d, radius, min_diag, max_diag, sbase);
        LAUNCH_CHECK();
        return ESFM_OK;
    }
    if (int rc = scal_reserve<1>(st, d, slots, div_up(d.n_pt, 64), sbase)) return rc;
    hipLaunchKernelGGL(ba_point_prep_kernel, dim3(div_up(d.n_pt, 64)), dim3(64), 0, st, d, radius, min_diag, max_diag, fresh ? 1 : 0, sbase);
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
    if (d.parts->grad_done) { d.parts->grad_done = false; return ESFM_OK; }    // ba_camacc_reduce_kernel has done it (one rank)
    if (d.n_cam <= 0) return ESFM_OK;
    hipLaunchKernelGGL(ba_camera_gradient_kernel, dim3(div_up(6 * d.n_cam, 256)), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    return ESFM_OK;
}

// e with |x| < 2^e (the exponent the right-hand side's fixed point is scaled by)
static int bound_exponent(double x)
{
    if (!(x > 1e-120)) return -400;
    if (!(x < 1e120)) return 400;
    return std::ilogb(x) + 1;
}

static int schur_to_double(hipStream_t st, const BADev &d, int rhs_exp)
{
    const int n = 6 * d.n_cam;
    hipLaunchKernelGGL(ba_schur_to_double_kernel, dim3(div_up(n, 256), n + 1), dim3(256), 0, st, d, rhs_exp);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_schur(hipStream_t st, const BADev &d, int num_cu, double *slabs, size_t slab_capacity_doubles, double rhs_bound)
{
    // (the slab path writes every entry of red anything reads -- lower blocks and right-hand side -- in its reduce kernel; the other
    // paths and the free-intrinsics kernel accumulate into it)
    const bool slab_path = d.n_obs > 0 && !d.has_calib && slabs &&
                           sizeof(double) * ((size_t)(d.n_cam * (d.n_cam + 1) / 2) * kSchurPitch + 6 * (size_t)d.n_cam) + sizeof(int) * 6 * (size_t)d.n_cam <= 156 * 1024;
    d.parts->red_fixed = false;
    if (!slab_path && !d.parts->red_clean) ESFM_HIP_TRY(hipMemsetAsync(d.red, 0, sizeof(double) * ba_red_doubles(d.n_cam), st));
    if (d.n_obs <= 0) return ESFM_OK;
    d.parts->red_clean = false;
    const int rhs_exp = bound_exponent(rhs_bound);
    const bool finish = !d.has_calib;    // with free intrinsics ba_schur_calib adds its block row first, then converts
    const int nblk = d.n_cam * (d.n_cam + 1) / 2;
    const int slab_doubles = nblk * 36 + 6 * d.n_cam;
    const size_t lds_bytes = sizeof(double) * ((size_t)nblk * kSchurPitch + 6 * (size_t)d.n_cam) + sizeof(int) * 6 * (size_t)d.n_cam;
    constexpr int schur_threads = 1024;     // (512: 39 us, 256: 52 us against 35 us on BA-25 -- every workgroup zeroes and writes out a 96 KB slab)
    const int n_slabs = std::max(1, std::min(num_cu, div_up(d.n_obs, schur_threads)));
    if (lds_bytes <= 156 * 1024 && slabs && (size_t)n_slabs * slab_doubles <= slab_capacity_doubles) {
        ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_schur_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(ba_schur_lds_kernel, dim3(n_slabs), dim3(schur_threads), lds_bytes, st, d, slabs, slab_doubles, rhs_exp);
        LAUNCH_CHECK();
        if (finish) hipLaunchKernelGGL(ba_schur_reduce_kernel<true>, dim3(div_up(slab_doubles, kRedEnt)), dim3(256), 0, st, d, slabs, slab_doubles, n_slabs, rhs_exp);
        else hipLaunchKernelGGL(ba_schur_reduce_kernel<false>, dim3(div_up(slab_doubles, kRedEnt)), dim3(256), 0, st, d, slabs, slab_doubles, n_slabs, rhs_exp);
        LAUNCH_CHECK();
        return ESFM_OK;
    }
    const bool tables = d.n_mchunks[0] + d.n_mchunks[1] + d.n_chunks + d.n_chunks_b + d.n_wide_obs > 0;
    if (tables) {
        // narrow tracks (nearly all of them in a sequence capture): the matrix-core kernel, plain and rotated camera numbering
        if (d.n_mchunks[0] + d.n_mchunks[1] > 0) {
            SchurMfTables tabs;
            for (int tb = 0; tb < 2; ++tb) {
                tabs.slot_obs[tb] = d.mslot_obs[tb]; tabs.slot_pc[tb] = reinterpret_cast<const int2 *>(d.mslot_pc[tb]); tabs.batch_slot[tb] = d.mbatch_slot[tb];
                tabs.chunk_batch0[tb] = d.mchunk_batch0[tb]; tabs.chunk_cam0[tb] = d.mchunk_cam0[tb];
            }
            tabs.n_plain = d.n_mchunks[0]; tabs.rot_seam = d.n_real_cam / 2;
            ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_schur_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfLdsBytes));
            hipLaunchKernelGGL(ba_schur_mfma_kernel, dim3(d.n_mchunks[0] + d.n_mchunks[1]), dim3(256), kMfLdsBytes, st, d, rhs_exp, tabs);
            LAUNCH_CHECK();
        }
        constexpr size_t win_bytes = sizeof(double) * (kWinBlocks * kSchurPitch + 6 * kWinCams);
        if (d.n_chunks > 0 || d.n_chunks_b > 0)
            ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_schur_window_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_bytes));
        if (d.n_chunks > 0) {
            hipLaunchKernelGGL(ba_schur_window_kernel, dim3(d.n_chunks), dim3(1024), win_bytes, st, d, rhs_exp, d.slot_obs, d.chunk_slot, d.chunk_cam0, 0);
            LAUNCH_CHECK();
        }
        if (d.n_chunks_b > 0) {        // the tracks that are narrow once the camera indices are rotated by half the loop
            hipLaunchKernelGGL(ba_schur_window_kernel, dim3(d.n_chunks_b), dim3(1024), win_bytes, st, d, rhs_exp, d.slot_obs_b, d.chunk_slot_b, d.chunk_cam0_b,
                               d.n_real_cam / 2);
            LAUNCH_CHECK();
        }
        // tracks wider than the window in both index spaces go through the plain kernel, spread over the chip (inside the window
        // kernel they would all land in the chunks of the lowest cameras and the whole launch would wait for those workgroups)
        if (d.n_wide_obs > 0) {
            hipLaunchKernelGGL(ba_schur_kernel, dim3(div_up(d.n_wide_obs, 256)), dim3(256), 0, st, d, rhs_exp, d.wide_obs, d.n_wide_obs);
            LAUNCH_CHECK();
        }
    } else {
        hipLaunchKernelGGL(ba_schur_kernel, dim3(div_up(d.n_obs, 256)), dim3(256), 0, st, d, rhs_exp, (const int32_t *)nullptr, 0);
        LAUNCH_CHECK();
    }
    if (finish && ba_solve_is_tiled(d.n_cam) && (d.parts->single_rank || d.sparse)) {
        // the tiled solve's assembly kernel reads red next: it converts on the way and clears what it has read (the structure-aware
        // solve of several ranks: its pack kernel does)
        d.parts->red_fixed = true; d.parts->red_rhs_exp = rhs_exp;
        return ESFM_OK;
    }
    return finish ? schur_to_double(st, d, rhs_exp) : ESFM_OK;
}

bool ba_solve_is_tiled(int n_cam)
{
    const int n = 6 * n_cam;
    return n_cam > 0 && !ba_chol_small_fits(n_cam) && sizeof(double) * ((size_t)(n + 1) * (n + 2) / 2 + n + 2) > 150 * 1024;
}

int ba_solve_reduced(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag)
{
    if (d.n_cam <= 0) return ESFM_OK;
    const int n = 6 * d.n_cam;
    const size_t bytes = sizeof(double) * ((size_t)(n + 1) * (n + 2) / 2 + n + 2);
    if (ba_chol_small_fits(d.n_cam)) return ba_solve_reduced_small(st, d, radius, min_diag, max_diag);
    if (d.sparse) return ba_solve_reduced_sparse(st, d, d.sparse, radius, min_diag, max_diag);
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

// Scalar read-back without a stream synchronisation: one workgroup adds the pending per-workgroup partials to the sum slots (see
// scal_commit), copies the scalar slots into pinned host memory, fences to system scope and then stores the sequence number the
// host is spinning on.
__global__ __launch_bounds__(1024) void ba_publish_scalars_kernel(const double *__restrict__ scal_part, int scal_cap, double *scal, ScalCounts c,
                                                                  double *host, unsigned long long *flag, unsigned long long seq)
{
    scal_reduce_pending(scal_part, scal_cap, scal, c);
    __syncthreads();
    const int i = threadIdx.x;
    // system-scope stores (write-through to the pinned host page) ordered against the flag by a workgroup-scope fence (s_waitcnt) and
    // a barrier: a system-scope release fence here is a write-back of the whole L2 on this part (see ba_chol_large.hip, st_coh)
    if (i < SC_COUNT) {
        __hip_atomic_store(&host[i], scal[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        scal[i] = 0.0;       // every read-back is followed by a reset before the next scalar-producing launch: done here, not by a memset launch
    }
    // (the workgroup-scope fence emits no s_waitcnt vmcnt(0) on gfx950: the explicit one is what keeps the flag behind the slots)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (i == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static ScalCounts take_counts(const BADev &d)
{
    ScalCounts c;
    for (int q = 0; q < SC_SUM_COUNT; ++q) { c.n[q] = d.parts->n[q]; d.parts->n[q] = 0; }
    return c;
}

int ba_publish_scalars(hipStream_t st, const BADev &d, double *host, unsigned long long *flag, unsigned long long seq)
{
    hipLaunchKernelGGL(ba_publish_scalars_kernel, dim3(1), dim3(1024), 0, st, d.scal_part, d.scal_cap, d.scal, take_counts(d), host, flag, seq);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_scal_reduce(hipStream_t st, const BADev &d)
{
    bool any = false;
    for (int q = 0; q < SC_SUM_COUNT; ++q) any = any || d.parts->n[q] > 0;
    if (!any) return ESFM_OK;
    hipLaunchKernelGGL(ba_scal_reduce_kernel, dim3(1), dim3(1024), 0, st, d.scal_part, d.scal_cap, d.scal, take_counts(d));
    LAUNCH_CHECK();
    return ESFM_OK;
}

void ba_scal_discard(const BADev &d, int first_slot, int end_slot)
{
    for (int q = std::max(0, first_slot); q < std::min<int>(SC_SUM_COUNT, end_slot); ++q) d.parts->n[q] = 0;
}

int ba_camera_step(hipStream_t st, const BADev &d)
{
    if (ba_chol_small_fits(d.n_cam)) return ESFM_OK;     // the one-workgroup reduced solve has done it
    hipLaunchKernelGGL(ba_camera_step_kernel, dim3(1), dim3(256), 0, st, d);
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_backsub(hipStream_t st, const BADev &d, bool with_cost, double cauchy_a)
{
    if (d.n_pt <= 0 || d.n_pchunks <= 0) return ESFM_OK;
    ScalBase sbase;
    if (with_cost) {
        // the candidate's cost comes out of the same launch (SC_CAND_COST / SC_CAND_BAD): no ba_cost pass over the observations
        const int slots[6] = {SC_MODEL_CHANGE, SC_STEP_SQ_PT, SC_CAND_SQ_PT, SC_GDOTD, SC_CAND_COST, SC_CAND_BAD};
        if (int rc = scal_reserve<6>(st, d, slots, d.n_pchunks, sbase)) return rc;
        hipLaunchKernelGGL(ba_backsub_chunk_kernel<true>, dim3(d.n_pchunks), dim3(kPtChunkObs), 0, st, d, sbase, cauchy_a);
    } else {
        const int slots[4] = {SC_MODEL_CHANGE, SC_STEP_SQ_PT, SC_CAND_SQ_PT, SC_GDOTD};
        if (int rc = scal_reserve<4>(st, d, slots, d.n_pchunks, sbase)) return rc;
        hipLaunchKernelGGL(ba_backsub_chunk_kernel<false>, dim3(d.n_pchunks), dim3(kPtChunkObs), 0, st, d, sbase, cauchy_a);
    }
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_cost(hipStream_t st, const BADev &d, int num_cu, const double *cams, const double *pts, double cauchy_a, int slot, int bad_slot,
            bool with_slope)
{
    if (d.n_obs <= 0) return ESFM_OK;
    const int grid = std::min(div_up(d.n_obs, 256), std::max(1, num_cu) * 8);
    ScalBase sbase;
    if (with_slope) {
        const int slots[3] = {slot, bad_slot, SC_LS_GRAD};
        if (int rc = scal_reserve<3>(st, d, slots, grid, sbase)) return rc;
        hipLaunchKernelGGL(ba_cost_kernel<true>, dim3(grid), dim3(256), 0, st, d, cams, pts, cauchy_a, slot, bad_slot, sbase);
    } else {
        const int slots[2] = {slot, bad_slot};
        if (int rc = scal_reserve<2>(st, d, slots, grid, sbase)) return rc;
        hipLaunchKernelGGL(ba_cost_kernel<false>, dim3(grid), dim3(256), 0, st, d, cams, pts, cauchy_a, slot, bad_slot, sbase);
    }
    LAUNCH_CHECK();
    return ESFM_OK;
}

int ba_take_step(hipStream_t st, const BADev &d, double t)
{
    const int grid = std::max(1, std::min(div_up(3LL * d.n_pt, 256), 1024));
    ScalBase sbase;
    { const int slots[2] = {SC_STEP_SQ_PT, SC_CAND_SQ_PT}; if (int rc = scal_reserve<2>(st, d, slots, grid, sbase)) return rc; }
    hipLaunchKernelGGL(ba_take_step_kernel, dim3(grid), dim3(256), 0, st, d, t, sbase);
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

int ba_schur_calib(hipStream_t st, const BADev &d, double rhs_bound)
{
    if (!d.has_calib || d.n_pt <= 0 || d.n_obs <= 0) return ESFM_OK;
    const size_t row_bytes = sizeof(double) * (8 + (size_t)4 * 6 * d.n_real_cam);
    const bool use_lds = row_bytes <= 64 * 1024;
    const size_t lds = use_lds ? row_bytes : sizeof(double) * 8;
    const int rhs_exp = bound_exponent(rhs_bound);
    hipLaunchKernelGGL(ba_schur_calib_kernel, dim3(div_up(d.n_pt, 256)), dim3(256), lds, st, d, use_lds ? 1 : 0, rhs_exp);
    LAUNCH_CHECK();
    return schur_to_double(st, d, rhs_exp);
}

int ba_param_sqnorm(hipStream_t st, const BADev &d)
{
    const int grid = std::max(1, std::min(div_up(3LL * d.n_pt, 256), 1024));
    ScalBase sbase;
    { const int slots[1] = {SC_XNORM_SQ_PT}; if (int rc = scal_reserve<1>(st, d, slots, grid, sbase)) return rc; }
    hipLaunchKernelGGL(ba_param_sqnorm_kernel, dim3(grid), dim3(256), 0, st, d, sbase);
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
