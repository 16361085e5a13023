// Dense solve of the reduced camera system when it does not fit one workgroup's LDS (n = 6 n_cam > ~185;
// BASELINE config 5: 512 cameras, n = 3072).  Blocked right-looking Cholesky on 64 x 64 f64 tiles across the
// whole chip, the right-hand side carried as an extra block row (forward substitution for free), then a
// blocked backward substitution.  This is DENSE_SCHUR's factorisation step (reference cpp_code/src/ba.cpp:201,
// Ceres' Eigen LLT [upstream]) for large camera counts.
//
//   chol_assemble_kernel   W = F'F + D_c^2 + S_schur (lower), rhs row = F'r + rhs_corr, identity padding
//   chol_potrf0_kernel     factor of the first diagonal tile (one wave)
//   chol_trsm_kernel(k)    X_ik = A_ik L_kk^-T for the tiles below the diagonal one (one wave per tile, rhs block row included)
//   chol_update_kernel(k)  trailing update C_ij -= X_ik X_jk' for k < j <= i on v_mfma_f64_16x16x4_f64; the workgroup that
//                          finishes tile (k+1, k+1) factors it in the same launch (one wave, no barrier in the factorisation)
//   chol_back_kernel(k)    y_k = L_kk^-T z_k (redundantly per workgroup), z_b -= L_kb' y_k for b < k
#include "ba_kernels.hpp"

namespace esfm {

constexpr int CB = 64;          // tile edge
constexpr int CLD = CB + 1;     // LDS leading dimension of the back-substitution's tile (f64, odd: conflict-free column access)

__global__ __launch_bounds__(256) void chol_assemble_kernel(BADev d, double *__restrict__ W, int ld, int nb, double radius,
                                                            double min_diag, double max_diag)
{
    const int n = 6 * d.n_cam;
    const long long rows = (long long)(nb + 1) * CB;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * ld) return;
    const int i = (int)(e / ld), j = (int)(e % ld);
    double v = 0.0;
    if (i < n) {
        if (j <= i) {
            v = d.red[(size_t)i * n + j];
            if (i / 6 == j / 6) {
                const int c = i / 6;
                v += d.camacc[36 * (size_t)c + 6 * (i % 6) + (j % 6)];
                if (i == j) v += fmin(fmax(d.camacc[36 * (size_t)c + 7 * (i % 6)], min_diag), max_diag) / radius;
            }
        }
    } else if (i < nb * CB) {
        v = (i == j) ? 1.0 : 0.0;  // padding rows: identity
    } else if (i == nb * CB) {
        v = (j < n) ? d.camacc[36 * (size_t)d.n_cam + j] + d.red[(size_t)n * n + j] : 0.0;  // rhs row
    }
    W[(size_t)i * ld + j] = v;
}

// ---------------------------------------------------------------------------------------------
// 64 x 64 tile kernels built from 16 x 16 sub-blocks on the f64 matrix cores (v_mfma_f64_16x16x4_f64: lane l supplies
// A[l & 15][l >> 4] and B[l >> 4][l & 15]; its four results are D[(l >> 4) + 4 g][l & 15]).  Everything is a short loop:
// round 1 ran the tile factorisation with 256 threads between barriers out of LDS (45 us per block column); two one-wave
// rewrites measured this round were no better -- tile in LDS: 40 us (36 dependent ds_read_b64 per 32 FMAs); tile in
// registers, fully unrolled: 78 us (25 KB of straight-line code executed once per launch: instruction fetch).
//   P Q^T accumulate:  acc += sign * P (16 x 16, row-major in LDS) * Q^T (Q 16 x 16 row-major in LDS)  -- 4 MFMAs
//   potrf16_inv:       one wave factors a 16 x 16 diagonal sub-block in registers (lane = row, v_readlane broadcasts) and
//                      inverts the factor (lane = column of the inverse), ~500 instructions
//   tile_potrf64:      4 sub-block steps: potrf16_inv, panel X = A Linv^T (MFMA), trailing update (MFMA); 3 barriers each
//   strip_trsm64:      one wave solves its 16-row strip of X = A L^-T: per sub-block  acc = A_b - sum X_b' L_bb'^T,
//                      X_b = acc Linv_bb^T, all MFMA, re-laid out through the wave's own LDS rows (no workgroup barrier)
constexpr int SB = 16;                  // sub-block edge
constexpr int ULD = CB + 2;             // LDS leading dimension of MFMA operand tiles: 32 lanes, 32 distinct 8-byte bank pairs
constexpr int VLD = SB + 2;             // the same for a 16 x 16 block
constexpr int LSLOT = CB * CB + CB + 4 * SB * SB;   // per diagonal tile in Ldiag: L, 1 / diag(L), inverses of its four sub-blocks
typedef double doublex4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double lane_value_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// 1 / sqrt(x) for the pivots: the hardware seed (v_rsq_f64, ~2^-26) and two Newton steps in f64 -- a short dependent chain; the
// library rsqrt's longer one sits on the critical path of every one of the n pivots (dependent f64 operations cost ~16 cycles each)
__device__ __forceinline__ double rsqrt_pivot(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-h, y * y, 1.5);
    y = y * fma(-h, y * y, 1.5);
    return y;
}

__device__ __forceinline__ doublex4 pqt16(doublex4 acc, const double *P, int ldp, const double *Q, int ldq, double sign, int lane)
{
    const double *pp = P + (lane & 15) * ldp + (lane >> 4), *qp = Q + (lane & 15) * ldq + (lane >> 4);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * pp[4 * kk], qp[4 * kk], acc, 0, 0, 0);
    return acc;
}
// D-layout <-> row-major 16 x 16 block in LDS
__device__ __forceinline__ doublex4 load_d16(const double *B, int ld, int lane)
{
    doublex4 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = B[((lane >> 4) + 4 * g) * ld + (lane & 15)];
    return v;
}
__device__ __forceinline__ void store_d16(double *B, int ld, doublex4 v, int lane)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) B[((lane >> 4) + 4 * g) * ld + (lane & 15)] = v[g];
}

// One wave: Cholesky factor of the 16 x 16 block D (LDS, row-major ldd) written back in place (upper part zeroed), the
// inverse of the factor to Vi (row-major VLD) and the reciprocal diagonal to rd[0..16).  Lane r (and r + 16, r + 32, r + 48,
// redundantly) holds row r in registers; the pivot and the column below it reach the other lanes through v_readlane -- per
// pivot: readlane, 1 / sqrt, scale, readlane, FMA, ~200 dependent cycles.  (A version that kept the block in LDS and published
// column and row through it measured 675 cycles per pivot, 10.0k per block against 5.0k for this one:
// scratch/ubench/potrf_bench.hip.)  The inverse is formed right-looking too (lane = column of the inverse): no serial FMA chain.
template <bool WRITE_L = true>
__device__ __forceinline__ void potrf16_inv(double *D, int ldd, double *Vi, double *rd, int *fail, int lane)
{
    const int r = lane & 15;
    double x[SB];
#pragma unroll
    for (int c = 0; c < SB; ++c) x[c] = D[r * ldd + c];
    double my_rd = 1.0;
#pragma unroll
    for (int c = 0; c < SB; ++c) {
        const double piv = lane_value_f64(x[c], c);
        if (!(piv > 0.0) || !isfinite(piv)) { if (lane == 0) *fail = 1; }
        const double rinv = rsqrt_pivot(piv > 0.0 ? piv : 1.0);
        x[c] = (r == c) ? piv * rinv : x[c] * rinv;
        if (r == c) my_rd = rinv;
#pragma unroll
        for (int c2 = c + 1; c2 < SB; ++c2) x[c2] -= x[c] * lane_value_f64(x[c], c2);   // rows above the diagonal: garbage, never read
    }
    // inverse: lane r solves L y = e_r.  t starts as e_r; step k fixes y_k = t_k / L_kk and eliminates it from the rows below.
    double t[SB];
#pragma unroll
    for (int i = 0; i < SB; ++i) t[i] = (r == i) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        t[k] *= lane_value_f64(my_rd, k);
#pragma unroll
        for (int i = k + 1; i < SB; ++i) t[i] -= lane_value_f64(x[k], i) * t[k];      // L[i][k] = row i's x[k]
    }
    if (lane < SB) {
        if (WRITE_L) {
#pragma unroll
            for (int c = 0; c < SB; ++c) D[r * ldd + c] = (c <= r) ? x[c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) Vi[i * VLD + r] = (i >= r) ? t[i] : 0.0;       // (WRITE_L false: Vi may be D itself)
        rd[r] = my_rd;
    }
}

// 256 threads: in-place Cholesky of the 64 x 64 tile T (LDS, row-major ULD; upper part must be zero).  Vi: 4 blocks of
// 16 x VLD (inverses of the diagonal sub-blocks), rd: 64 reciprocal diagonals.
__device__ __forceinline__ void tile_potrf64(double *T, double *Vi, double *rd, int *fail)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
        if (wave == 0) potrf16_inv(T + (SB * b) * ULD + SB * b, ULD, Vi + b * SB * VLD, rd + SB * b, fail, lane);
        __syncthreads();
        // panel: strips i = b+1 .. 3:  X_i = A_i Linv_bb^T   (A_i is read whole before it is overwritten: one wave per strip)
        if (wave > b) {
            const int i = wave;
            doublex4 acc = pqt16(doublex4{0.0, 0.0, 0.0, 0.0}, T + (SB * i) * ULD + SB * b, ULD, Vi + b * SB * VLD, VLD, 1.0, lane);
            __builtin_amdgcn_wave_barrier();
            store_d16(T + (SB * i) * ULD + SB * b, ULD, acc, lane);
        }
        __syncthreads();
        // trailing update: blocks (i, j), b < j <= i <= 3, one per wave and round
        int idx = 0;
        for (int i = b + 1; i < 4; ++i)
            for (int j = b + 1; j <= i; ++j, ++idx)
                if ((idx & 3) == wave) {
                    doublex4 acc = load_d16(T + (SB * i) * ULD + SB * j, ULD, lane);
                    acc = pqt16(acc, T + (SB * i) * ULD + SB * b, ULD, T + (SB * j) * ULD + SB * b, ULD, -1.0, lane);
                    store_d16(T + (SB * i) * ULD + SB * j, ULD, acc, lane);
                }
        __syncthreads();
    }
}

// One wave: X = A L^-T for its 16-row strip S (LDS, 16 x 64, row-major ULD, in place).  L: the factored 64 x 64 tile (LDS,
// ULD), Vi: the inverses of its diagonal sub-blocks.  scratch: 16 x VLD doubles private to the wave.
__device__ __forceinline__ void strip_trsm64(double *S, const double *L, const double *Vi, double *scratch, int lane)
{
    for (int b = 0; b < 4; ++b) {
        doublex4 acc = load_d16(S + SB * b, ULD, lane);
        for (int bp = 0; bp < b; ++bp) acc = pqt16(acc, S + SB * bp, ULD, L + (SB * b) * ULD + SB * bp, ULD, -1.0, lane);
        store_d16(scratch, VLD, acc, lane);
        __builtin_amdgcn_wave_barrier();
        acc = pqt16(doublex4{0.0, 0.0, 0.0, 0.0}, scratch, VLD, Vi + b * SB * VLD, VLD, 1.0, lane);
        __builtin_amdgcn_wave_barrier();
        store_d16(S + SB * b, ULD, acc, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

// publish a factored tile: L (row-major 64 x 64, upper part zero), reciprocal diagonal, sub-block inverses
__device__ __forceinline__ void publish_diag(double *__restrict__ Ld, const double *T, const double *Vi, const double *rd)
{
    const int tid = threadIdx.x;
    for (int e = tid; e < CB * CB; e += 256) Ld[e] = T[(e / CB) * ULD + (e % CB)];
    if (tid < CB) Ld[CB * CB + tid] = rd[tid];
    for (int e = tid; e < 4 * SB * SB; e += 256) Ld[CB * CB + CB + e] = Vi[(e / SB) * VLD + (e % SB)];
}

// Factor of the FIRST diagonal tile (the others are factored by the trailing update that finishes them).
// Ldiag: LSLOT doubles per diagonal tile, read by the TRSM and back-substitution launches.
__global__ __launch_bounds__(256) void chol_potrf0_kernel(double *__restrict__ W, double *__restrict__ Ldiag, int ld, double *__restrict__ scal)
{
    __shared__ double T[CB * ULD];
    __shared__ double Vi[4 * SB * VLD];
    __shared__ double rd[CB];
    __shared__ int fail;
    const int tid = threadIdx.x;
    if (tid == 0) fail = 0;
    for (int e = tid; e < CB * CB; e += 256) { const int r = e / CB, c = e % CB; T[r * ULD + c] = (c <= r) ? W[(size_t)r * ld + c] : 0.0; }
    __syncthreads();
    tile_potrf64(T, Vi, rd, &fail);
    publish_diag(Ldiag, T, Vi, rd);
    if (tid == 0 && fail) scal[SC_CHOL_FAIL] = 1.0;
}

// X_ik = A_ik L_kk^-T for the block rows i = k + 1 .. nb (nb = right-hand-side block row): one workgroup per tile, one wave
// per 16-row strip.
__global__ __launch_bounds__(256) void chol_trsm_kernel(double *__restrict__ W, const double *__restrict__ Ldiag, int ld, int k)
{
    __shared__ double L[CB * ULD];
    __shared__ double T[CB * ULD];
    __shared__ double Vi[4 * SB * VLD];
    __shared__ double scratch[4][SB * VLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bi = k + 1 + blockIdx.x;
    const double *__restrict__ Lk = Ldiag + (size_t)k * LSLOT;
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c = e % CB;
        L[r * ULD + c] = Lk[e];
        T[r * ULD + c] = W[(size_t)(bi * CB + r) * ld + k * CB + c];
    }
    for (int e = tid; e < 4 * SB * SB; e += 256) Vi[(e / SB) * VLD + (e % SB)] = Lk[CB * CB + CB + e];
    __syncthreads();
    strip_trsm64(T + (SB * wave) * ULD, L, Vi, scratch[wave], lane);
    __syncthreads();
    for (int e = tid; e < CB * CB; e += 256) W[(size_t)(bi * CB + e / CB) * ld + k * CB + (e % CB)] = T[(e / CB) * ULD + (e % CB)];
}

// Trailing update C_ij -= X_ik X_jk' for the tiles k < j <= i <= nb (j <= nb - 1) on the f64 matrix cores: wave w of the
// workgroup owns rows [16 w, 16 w + 16) of the tile, four 16 x 16 outputs, K = 64 in 16 steps.  The workgroup that finishes the
// NEXT diagonal tile (k + 1, k + 1) factors it on the spot (tile_potrf64) and publishes it, so the factorisation never costs a
// launch of its own.
__global__ __launch_bounds__(256) void chol_update_kernel(double *__restrict__ W, double *__restrict__ Ldiag, int ld, int nb, int k,
                                                          double *__restrict__ scal)
{
    __shared__ double Ai[CB * ULD];
    __shared__ double Aj[CB * ULD];
    __shared__ double Vi[4 * SB * VLD];
    __shared__ double rd[CB];
    __shared__ int fail;
    // linear tile id -> (i, j): tiles of block row i (k+1 .. nb) are j = k+1 .. min(i, nb-1)
    const int m = nb - k - 1;  // square trailing block rows
    int t = blockIdx.x, i, j;
    const int tri = m * (m + 1) / 2;
    if (t < tri) {
        int ri = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((ri + 1) * (ri + 2) / 2 <= t) ++ri;
        while (ri * (ri + 1) / 2 > t) --ri;
        i = k + 1 + ri; j = k + 1 + (t - ri * (ri + 1) / 2);
    } else {
        i = nb; j = k + 1 + (t - tri);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c = e % CB;
        Ai[r * ULD + c] = W[(size_t)(i * CB + r) * ld + k * CB + c];
        Aj[r * ULD + c] = W[(size_t)(j * CB + r) * ld + k * CB + c];
    }
    if (tid == 0) fail = 0;
    __syncthreads();
    doublex4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = doublex4{0.0, 0.0, 0.0, 0.0};
    const double *ap = Ai + (16 * wave + (lane & 15)) * ULD + (lane >> 4);
    const double *bp = Aj + (lane & 15) * ULD + (lane >> 4);
#pragma unroll 4
    for (int kk = 0; kk < CB / 4; ++kk) {
        const double a = ap[4 * kk];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[16 * cb * ULD + 4 * kk], acc[cb], 0, 0, 0);
    }
    const bool next_diag = (i == k + 1 && j == k + 1);
    if (!next_diag) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
                if (i != j || c <= r) W[(size_t)(i * CB + r) * ld + j * CB + c] -= acc[cb][g];
            }
        return;
    }
    // the next diagonal tile: finish it in LDS (Ai is free once every wave is past its MFMAs), factor, publish
    __syncthreads();
    double *T = Ai;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
            T[r * ULD + c] = (c <= r) ? W[(size_t)(i * CB + r) * ld + j * CB + c] - acc[cb][g] : 0.0;
        }
    __syncthreads();
    tile_potrf64(T, Vi, rd, &fail);
    publish_diag(Ldiag + (size_t)(k + 1) * LSLOT, T, Vi, rd);
    for (int e = tid; e < CB * CB; e += 256) { const int r = e / CB, c = e % CB; if (c <= r) W[(size_t)(i * CB + r) * ld + j * CB + c] = T[r * ULD + c]; }
    if (tid == 0 && fail) scal[SC_CHOL_FAIL] = 1.0;
}

// Backward substitution step k: z = rhs row (row nb*CB of W).  Every workgroup solves L_kk' y_k = z_k (one wave),
// workgroup b < k then applies z_b -= L_kb' y_k; workgroup k stores y_k.
__global__ __launch_bounds__(256) void chol_back_kernel(double *__restrict__ W, const double *__restrict__ Ldiag, int ld, int nb, int k)
{
    __shared__ double Lkk[CB * CLD];
    __shared__ double y[CB];
    __shared__ double rd[CB];
    const int tid = threadIdx.x;
    double *z = W + (size_t)nb * CB * ld;
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c = e % CB;
        Lkk[r * CLD + c] = Ldiag[(size_t)k * LSLOT + e];
    }
    if (tid < CB) rd[tid] = Ldiag[(size_t)k * LSLOT + CB * CB + tid];
    __syncthreads();
    if (tid < CB) {  // one wave; lane = row index, its y value lives in a register
        double yl = z[k * CB + tid];
        for (int c = CB - 1; c >= 0; --c) {
            const double yc = __shfl(yl, c) * rd[c];
            if (tid == c) yl = yc;
            if (tid < c) yl -= Lkk[c * CLD + tid] * yc;
        }
        y[tid] = yl;
    }
    __syncthreads();
    const int b = blockIdx.x;
    if (b == k) { if (tid < CB) z[k * CB + tid] = y[tid]; return; }
    // z_b[t] -= sum_r L[k*CB + r][b*CB + t] * y[r]; 4 row-groups per column, reduced through LDS
    __shared__ double part[4][CB];
    const int t = tid & 63, g = tid >> 6;
    double s = 0.0;
    for (int r = g; r < CB; r += 4) s += W[(size_t)(k * CB + r) * ld + b * CB + t] * y[r];
    part[g][t] = s;
    __syncthreads();
    if (tid < CB) z[b * CB + tid] -= part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
}

// ---------------------------------------------------------------------------------------------
// Reduced systems that fit ONE workgroup's LDS (n = 6 n_cam <= 176, i.e. up to 29 cameras: BASELINE's BA-25 has n = 150): the
// whole solve -- and the camera step that follows it -- in one launch.  The lower block triangle lives in LDS (block (i, j) at
// i (i + 1) / 2 + j, 16 x VLD doubles each), the right-hand side is carried along as a row vector z (forward substitution for
// free).  The n pivots are a serial chain, so everything that is not the chain is kept off it.  Per block column b:
//   1. wave 0 factors the diagonal block in registers and inverts the factor in the same pass (potrf16_fused_inv: lane = row,
//      v_readlane broadcasts shared by both recurrences);
//   2. the panel X_i = A_i Linv' and 3. the trailing update A_ij -= X_i X_j' run on the f64 matrix cores, one 16 x 16 block per
//      wave and round.
// The backward substitution then runs through the block inverses as matrix-vector products, and the workgroup writes y and the
// candidate cameras (ba_camera_step).  (Measured alternative: factor only + one thread per panel row solving by substitution +
// the inverses side by side at the end: 82 us against 90 us for the unfused factor-then-invert; substitution is 16 dependent
// LDS-fed steps per block column, 1.45 us.)
// Round 1's kernel (ba_chol_solve_kernel: 8-column panels, dot products from packed rows, 19 panels x 3 barriers): 127 us.
constexpr int kSmallThreads = 1024;
constexpr int kSmallMaxNb = 11;
__device__ __forceinline__ int blk_off(int i, int j) { return (i * (i + 1) / 2 + j) * (SB * VLD); }

// 1 / sqrt(x) from the hardware seed (v_rsq_f64, relative error <= 2^-26) and ONE Newton step: relative error <= 1.5 * 2^-52.
// The factor only has to be backward stable (the iteration log is compared with the oracle at 1e-9), and the second step is 30
// cycles on the critical path of each of the n pivots.
__device__ __forceinline__ double rsqrt_pivot1(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    return y * fma(-0.5 * x, y * y, 1.5);
}

// One wave: the 16 x 16 block D (LDS, row-major VLD) is replaced by the INVERSE of its Cholesky factor (above the diagonal: +-0).
// Lane r (and r + 16, ... redundantly) holds row r of the block in x and solves L y = e_r in t.  Factorisation and inversion are
// both right-looking and share every broadcast: once column c is final, lane i's L[i][c] (DPP row_newbcast) updates column i of
// the block (x[i] -= L[r][c] L[i][c]) AND row i of the inverse (t[i] -= L[i][c] t[c]).  The whole thing is one generated asm
// block (gen_potrf16_asm.py -> potrf16_gfx950.inc, ~740 instructions): the pivots are a serial chain (broadcast, v_rsq_f64, one
// Newton step, scale) and the updates have to be issued in the shadow of its latencies, which hipcc does not do -- measured per
// 16 x 16 block at n = 150: factor then invert, v_readlane broadcasts: 9.7k cycles; fused, v_readlane: 8.4k (the compiler parks
// 30 scalars per pivot in VGPR lanes, v_writelane + s_nop); fused, DPP from C++: spills to scratch; this one: see DESIGN.md.
#include "potrf16_gfx950.inc"
__device__ __forceinline__ void potrf16_fused_inv(double *D, int *fail, int lane)
{
    const int r = lane & 15;
    const uint32_t row_addr = (uint32_t)(uintptr_t)(D + r * VLD), col_addr = (uint32_t)(uintptr_t)(D + r);
    int bad;
    asm volatile(ESFM_POTRF16_ASM : "=&v"(bad) : "v"(row_addr), "v"(col_addr), "v"(lane) : ESFM_POTRF16_CLOBBERS);
    if (__any(bad) && lane == 0) *fail = 1;
}

__global__ __launch_bounds__(kSmallThreads) void ba_chol_small_kernel(BADev d, double radius, double min_diag, double max_diag)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = 6 * d.n_cam, nb = (n + SB - 1) / SB, np = nb * SB;
    double *A = sm;                                     // nb (nb + 1) / 2 blocks; after the factorisation a diagonal block holds the INVERSE of its factor
    double *z = A + (size_t)(nb * (nb + 1) / 2) * (SB * VLD);    // np: right-hand side, then the solution
    double *rd = z + np;                                // np
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) fail = 0;
#ifdef ESFM_CHOL_PROFILE
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm0 = wall_clock64();
#define CHOL_MARK(q) do { const long long tm1 = wall_clock64(); prof[q] += tm1 - tm0; tm0 = tm1; } while (0)
#else
#define CHOL_MARK(q) do { } while (0)
#endif
    const double *S = d.red, *rc = d.red + (size_t)n * n, *FtF = d.camacc, *Ftr = d.camacc + 36 * (size_t)d.n_cam;
    // assemble W = F'F + D_c^2 + S_schur (lower block triangle, full diagonal blocks), identity padding, z = F'r + rhs_corr.
    // Pass 1: S -> LDS, one block per wave and round, lane = (row, 4 consecutive columns), every round's loads in flight at
    // once (the rounds used to wait for each other's memory round trip: 7.7 us).  Pass 2: the 6 x 6 camera blocks of F'F + D^2.
    {
        constexpr int kRounds = (kSmallMaxNb * (kSmallMaxNb + 1) / 2 + kSmallThreads / 64 - 1) / (kSmallThreads / 64);
        const int nblk = nb * (nb + 1) / 2;
        const int r = lane >> 2, c0 = (lane & 3) * 4;
        double v[kRounds][4];
        int bi = 0, bj = 0;
        for (int q = 0; q < wave; ++q) { if (++bj > bi) { ++bi; bj = 0; } }
#pragma unroll
        for (int q = 0; q < kRounds; ++q) {
            const bool live = wave + q * (kSmallThreads / 64) < nblk;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                int i = SB * bi + r, k = SB * bj + c0 + cc;
                if (k > i) { const int t = i; i = k; k = t; }   // only inside diagonal blocks: mirror (the factorisation reads the lower part)
                v[q][cc] = (live && i < n) ? S[(size_t)i * n + k] : ((live && i == k) ? 1.0 : 0.0);
            }
            for (int w = 0; w < kSmallThreads / 64; ++w) { if (++bj > bi) { ++bi; bj = 0; } }
        }
#pragma unroll
        for (int q = 0; q < kRounds; ++q) {
            const int blk = wave + q * (kSmallThreads / 64);
            if (blk < nblk) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) A[blk * (SB * VLD) + r * VLD + c0 + cc] = v[q][cc];
            }
        }
    }
    double ftf = 0.0;
    if (tid < 36 * d.n_cam) ftf = FtF[tid];                 // (n_cam <= 29: one entry per thread)
    const double zr = tid < n ? Ftr[tid] + rc[tid] : 0.0;
    __syncthreads();
    if (tid < 36 * d.n_cam) {
        const int cam = tid / 36, a = (tid % 36) / 6, b2 = tid % 6;
        if (b2 <= a) {
            const int i = 6 * cam + a, k = 6 * cam + b2;
            double add = ftf;
            if (a == b2) add += fmin(fmax(ftf, min_diag), max_diag) / radius;
            double *B = A + blk_off(i / SB, k / SB);
            B[(i % SB) * VLD + (k % SB)] += add;
            if (i / SB == k / SB && i != k) B[(k % SB) * VLD + (i % SB)] += add;
        }
    }
    if (tid < np) z[tid] = zr;
    __syncthreads();
    CHOL_MARK(0);

    // Block column b: panel X_i = A_i Linv_b' (MFMA), then the trailing update A_ij -= X_i X_j' (MFMA) -- during which wave 0
    // updates block (b+1, b+1) FIRST and factors / inverts it (look-ahead: the serial pivot chain of the next block column runs
    // in the shadow of this one's update).  The right-hand side rides along off that critical path: z_b <- z_b Linv_b' by the
    // last wave during the panel, z_j -= z_b X_j' by the last threads during the update.
    if (wave == 0) potrf16_fused_inv(A + blk_off(0, 0), &fail, lane);
    __syncthreads();
    CHOL_MARK(1);
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        if (wave == kSmallThreads / 64 - 1) {
            double acc = 0.0;
            if (lane < SB) {
#pragma unroll
                for (int k = 0; k < SB; ++k) acc = fma(z[SB * b + k], A[blk_off(b, b) + lane * VLD + k], acc);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < SB) z[SB * b + lane] = acc;
        }
        for (int i = b + 1 + wave; i < nb; i += kSmallThreads / 64) {
            doublex4 acc = pqt16(doublex4{0.0, 0.0, 0.0, 0.0}, A + blk_off(i, b), VLD, A + blk_off(b, b), VLD, 1.0, lane);
            __builtin_amdgcn_wave_barrier();
            store_d16(A + blk_off(i, b), VLD, acc, lane);
        }
        __syncthreads();
        CHOL_MARK(2);
        const int m = nb - 1 - b;
        if (wave == 0) {
            if (m > 0) {
                doublex4 acc = load_d16(A + blk_off(b + 1, b + 1), VLD, lane);
                acc = pqt16(acc, A + blk_off(b + 1, b), VLD, A + blk_off(b + 1, b), VLD, -1.0, lane);
                store_d16(A + blk_off(b + 1, b + 1), VLD, acc, lane);
                __builtin_amdgcn_wave_barrier();
                potrf16_fused_inv(A + blk_off(b + 1, b + 1), &fail, lane);
            }
        } else {
            // the other trailing blocks (b + 1 < i, j <= i) over waves 1..15, and z_j -= z_b X_j' for j > b
            for (int t = wave; t < m * (m + 1) / 2; t += kSmallThreads / 64 - 1) {
                int ri = 0, rj = t;
                while (rj > ri) { rj -= ri + 1; ++ri; }
                const int i = b + 1 + ri, j = b + 1 + rj;
                doublex4 acc = load_d16(A + blk_off(i, j), VLD, lane);
                acc = pqt16(acc, A + blk_off(i, b), VLD, A + blk_off(j, b), VLD, -1.0, lane);
                store_d16(A + blk_off(i, j), VLD, acc, lane);
            }
            for (int e = kSmallThreads - 1 - tid; e < m * SB; e += kSmallThreads - 64) {     // (the last waves have the fewest blocks)
                const int j = b + 1 + e / SB, c = e % SB;
                const double *Xj = A + blk_off(j, b) + c * VLD;
                double s0 = 0.0;
#pragma unroll
                for (int k = 0; k < SB; ++k) s0 = fma(z[SB * b + k], Xj[k], s0);
                z[SB * j + c] -= s0;
            }
        }
        __syncthreads();
        CHOL_MARK(3);
    }
    CHOL_MARK(4);
    // backward substitution L' y = z through the block inverses: y_b = Linv_bb' (z_b - sum_{i > b} L_ib' y_i)
    for (int b = nb - 1; b >= 0; --b) {
        if (tid < SB) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < SB; ++k) acc += A[blk_off(b, b) + k * VLD + tid] * z[SB * b + k];
            rd[SB * b + tid] = acc;          // y_b (rd is free after the inversions)
        }
        __syncthreads();
        for (int e = tid; e < b * SB; e += kSmallThreads) {
            const int j = e / SB, c = e % SB;
            const double *Lbj = A + blk_off(b, j);
            double s0 = 0.0;
#pragma unroll
            for (int k = 0; k < SB; ++k) s0 += Lbj[k * VLD + c] * rd[SB * b + k];
            z[SB * j + c] -= s0;
        }
        __syncthreads();
    }
    CHOL_MARK(5);
    for (int i = tid; i < n; i += kSmallThreads) { const double y = fail ? 0.0 : rd[i]; rd[i] = y; d.y_c[i] = y; }
    if (tid == 0 && fail) d.scal[SC_CHOL_FAIL] = 1.0;
    __syncthreads();
    ba_camera_step_body(d, rd, z);       // candidate cameras from y (z: reduction scratch from here on)
#ifdef ESFM_CHOL_PROFILE
    CHOL_MARK(6);
    if (tid == 0) printf("chol_small [10 ns]: assemble %lld potrf %lld panel %lld update %lld inverse %lld back %lld tail %lld\n", prof[0], prof[1], prof[2], prof[3], prof[4], prof[5], prof[6]);
#endif
#undef CHOL_MARK
}

bool ba_chol_small_fits(int n_cam) { return (6 * n_cam + SB - 1) / SB <= kSmallMaxNb; }

int ba_solve_reduced_small(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag)
{
    const int n = 6 * d.n_cam, nb = (n + SB - 1) / SB;
    const size_t bytes = sizeof(double) * ((size_t)(nb * (nb + 1) / 2) * (SB * VLD) + 2 * (size_t)nb * SB);
    ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_chol_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(ba_chol_small_kernel, dim3(1), dim3(kSmallThreads), bytes, st, d, radius, min_diag, max_diag);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

__global__ void chol_extract_kernel(BADev d, const double *__restrict__ W, int ld, int nb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = 6 * d.n_cam;
    if (i >= n) return;
    const bool fail = d.scal[SC_CHOL_FAIL] != 0.0;
    d.y_c[i] = fail ? 0.0 : W[(size_t)nb * CB * ld + i];
}

size_t ba_chol_large_doubles(int n_cam)
{
    const int n = 6 * n_cam, nb = (n + CB - 1) / CB;
    return (size_t)(nb + 1) * CB * (size_t)(nb * CB) + (size_t)nb * LSLOT;
}

int ba_solve_reduced_large(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag)
{
    const int n = 6 * d.n_cam, nb = (n + CB - 1) / CB, ld = nb * CB;
    double *W = d.chol;
    double *Ldiag = W + (size_t)(nb + 1) * CB * ld;
    const long long tot = (long long)(nb + 1) * CB * ld;
    hipLaunchKernelGGL(chol_assemble_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d, W, ld, nb, radius, min_diag, max_diag);
    ESFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(chol_potrf0_kernel, dim3(1), dim3(256), 0, st, W, Ldiag, ld, d.scal);
    for (int k = 0; k < nb; ++k) {
        // block column k: X = A L_kk^-T for the tiles below the diagonal one (right-hand-side row included), then the trailing
        // update, whose (k + 1, k + 1) workgroup also factors the next diagonal tile
        hipLaunchKernelGGL(chol_trsm_kernel, dim3(nb - k), dim3(256), 0, st, W, Ldiag, ld, k);
        const int m = nb - k - 1;
        const int tiles = m * (m + 1) / 2 + m;
        if (tiles > 0) hipLaunchKernelGGL(chol_update_kernel, dim3(tiles), dim3(256), 0, st, W, Ldiag, ld, nb, k, d.scal);
    }
    ESFM_HIP_TRY(hipGetLastError());
    for (int k = nb - 1; k >= 0; --k) hipLaunchKernelGGL(chol_back_kernel, dim3(k + 1), dim3(256), 0, st, W, Ldiag, ld, nb, k);
    hipLaunchKernelGGL(chol_extract_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d, W, ld, nb);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm
