// Dense solve of the reduced camera system: DENSE_SCHUR's factorisation step (reference cpp_code/src/ba.cpp:201, Ceres' Eigen
// LLT [upstream]).
//
// n = 6 n_cam > 176 (BASELINE config 5: 512 cameras, n = 3072): blocked right-looking Cholesky on 64 x 64 f64 tiles across the
// whole chip, the right-hand side carried as an extra block row (forward substitution for free), then the backward substitution.
//   chol_assemble_kernel   W = F'F + D_c^2 + S_schur (lower), rhs row = F'r + rhs_corr, identity padding
//   chol2_step_kernel(k)   ONE launch per block column: trailing update C_ij -= X_ik X_jk' on v_mfma_f64_16x16x4_f64; its first
//                          workgroup factors the next diagonal tile (tile_potrf64) and inverts the factor (inv64); the workgroups
//                          of the next block column wait for that inverse and turn their tiles into factor tiles X_i,k+1
//   chol2_back_kernel      the whole backward substitution in one launch, solution blocks handed on through flags in memory
// n <= 176 (BASELINE's BA-25: n = 150): ba_chol_small_kernel, the whole solve and the camera step in one workgroup.
// Both sit on potrf16_fused_*: one wave factors a 16 x 16 block and inverts the factor in one pass of generated, scheduled asm.
#include "ba_kernels.hpp"

namespace esfm {

constexpr int CB = 64;          // tile edge

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
// A[l & 15][l >> 4] and B[l >> 4][l & 15]; its four results are D[(l >> 4) + 4 g][l & 15]).
//   pqt16 / pq16:      acc += sign * P Q' / P Q  for 16 x 16 row-major blocks in LDS  -- 4 MFMAs
//   tile_potrf64:      4 sub-block steps: potrf16_fused_full, panel X = A Linv' (MFMA), trailing update (MFMA)
//   inv64:             the inverse of the factored tile from its sub-block inverses
// History of the diagonal-block factorisation, per 16 x 16 block: 256 threads out of LDS between barriers (round 1): 11 us; one
// wave, block in registers, v_readlane broadcasts, factor then invert: 4 us; one pass of scheduled asm with DPP broadcasts: 1.5 us.
constexpr int SB = 16;                  // sub-block edge
constexpr int ULD = CB + 2;             // LDS leading dimension of MFMA operand tiles: 32 lanes, 32 distinct 8-byte bank pairs
constexpr int VLD = SB + 2;             // the same for a 16 x 16 block
constexpr int LSLOT = CB * CB;          // per diagonal tile in Ldiag: the inverse of its factor (64 x 64, row-major)
constexpr int LINV_OFF = 0;
typedef double doublex4 __attribute__((ext_vector_type(4)));

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

// The same for the 64 x 64 tile factorisation: D (row pitch ldd doubles) keeps the factor (upper part zeroed), the inverse goes
// to Vi (row-major VLD) and the reciprocal diagonal to rd[0..16).
__device__ __forceinline__ void potrf16_fused_full(double *D, int ldd, double *Vi, double *rd, int *fail, int lane)
{
    const int r = lane & 15;
    const uint32_t row_addr = (uint32_t)(uintptr_t)(D + r * ldd), col_addr = (uint32_t)(uintptr_t)(Vi + r), rd_addr = (uint32_t)(uintptr_t)rd;
    int bad;
    asm volatile(ESFM_POTRF16_FULL_ASM : "=&v"(bad) : "v"(row_addr), "v"(col_addr), "v"(lane), "v"(rd_addr) : ESFM_POTRF16_CLOBBERS);
    if (__any(bad) && lane == 0) *fail = 1;
}

// 256 threads: in-place Cholesky of the 64 x 64 tile T (LDS, row-major ULD; upper part must be zero).  Vi: 4 blocks of
// 16 x VLD (inverses of the diagonal sub-blocks), rd: 64 reciprocal diagonals.  Sub-block column b: panel X_i = A_i Linv_bb'
// (waves b+1 .. 3), then the trailing update -- during which wave 0 updates sub-block (b+1, b+1) first and factors it at once
// (look-ahead: the next pivot chain runs while the other waves finish the update); two barriers per sub-block column.
__device__ __forceinline__ void tile_potrf64(double *T, double *Vi, double *rd, int *fail)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) potrf16_fused_full(T, ULD, Vi, rd, fail, lane);
    __syncthreads();
#pragma unroll 1
    for (int b = 0; b < 3; ++b) {
        // panel: strips i = b+1 .. 3:  X_i = A_i Linv_bb^T   (A_i is read whole before it is overwritten: one wave per strip)
        if (wave > b) {
            const int i = wave;
            doublex4 acc = pqt16(doublex4{0.0, 0.0, 0.0, 0.0}, T + (SB * i) * ULD + SB * b, ULD, Vi + b * SB * VLD, VLD, 1.0, lane);
            __builtin_amdgcn_wave_barrier();
            store_d16(T + (SB * i) * ULD + SB * b, ULD, acc, lane);
        }
        __syncthreads();
        // trailing update: blocks (i, j), b < j <= i <= 3; wave 0 takes (b+1, b+1) and goes on to factor it
        if (wave == 0) {
            const int i = b + 1;
            doublex4 acc = load_d16(T + (SB * i) * ULD + SB * i, ULD, lane);
            acc = pqt16(acc, T + (SB * i) * ULD + SB * b, ULD, T + (SB * i) * ULD + SB * b, ULD, -1.0, lane);
            store_d16(T + (SB * i) * ULD + SB * i, ULD, acc, lane);
            __builtin_amdgcn_wave_barrier();
            potrf16_fused_full(T + (SB * i) * ULD + SB * i, ULD, Vi + i * SB * VLD, rd + SB * i, fail, lane);
        } else {
            int idx = 0;
            for (int i = b + 1; i < 4; ++i)
                for (int j = b + 1; j <= i; ++j) {
                    if (i == b + 1 && j == b + 1) continue;
                    if (idx++ % 3 == wave - 1) {
                        doublex4 acc = load_d16(T + (SB * i) * ULD + SB * j, ULD, lane);
                        acc = pqt16(acc, T + (SB * i) * ULD + SB * b, ULD, T + (SB * j) * ULD + SB * b, ULD, -1.0, lane);
                        store_d16(T + (SB * i) * ULD + SB * j, ULD, acc, lane);
                    }
                }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Second generation of the large solve (this round): ONE launch per block column and ONE for the whole backward substitution.
//   * the workgroup that factors a diagonal tile also inverts the factor (inv64: the four 16 x 16 sub-block inverses are there
//     already; the six blocks below them are products), so "X_ik = A_ik L_kk^-T" is a dense product with L_kk^-1 and needs no
//     triangular sweep -- every trailing-update workgroup forms the X_ik, X_jk it needs itself (3 tile products instead of 1:
//     the matrix cores are idle anyway, the launch and the sweep were what cost 11 us per block column);
//   * the factor tiles X_ik go to a second matrix W2 (the update workgroups of the same launch still read A_ik from W);
//   * the backward substitution is one launch of nb workgroups that hand the solution blocks on through flags in memory:
//     workgroup b folds  z_b -= L_ib' y_i  for i = nb-1 .. b+1 as the y_i appear, then publishes  y_b = L_bb^-T z_b.
//     Workgroup b has blockIdx nb-1-b: it only ever waits for workgroups dispatched before it, so the chain cannot deadlock
//     whatever part of the grid is resident.  (48 launches of 16 us before.)
__device__ __forceinline__ doublex4 pq16(doublex4 acc, const double *P, int ldp, const double *Q, int ldq, double sign, int lane)
{
    // acc += sign * P (16 x 16 row-major) * Q (16 x 16 row-major)
    const double *pp = P + (lane & 15) * ldp + (lane >> 4), *qp = Q + (lane >> 4) * ldq + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * pp[4 * kk], qp[4 * kk * ldq], acc, 0, 0, 0);
    return acc;
}

// 256 threads: O (LDS, 64 x 64, row-major ULD) = inverse of the lower-triangular factor T (LDS, ULD) whose diagonal sub-block
// inverses are in Vi.  Wave j computes block column j:  O_jj = Vi_j,  O_ij = -Vi_i (sum_{m=j}^{i-1} T_im O_mj)  for i > j.
__device__ __forceinline__ void inv64(double *O, const double *T, const double *Vi, double (*scratch)[SB * VLD])
{
    const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6;
    for (int e = lane; e < SB * CB; e += 64) {          // this wave's block column: zero above the diagonal block, Vi_j on it
        const int r = e / SB, c = e % SB;
        O[r * ULD + SB * j + c] = (r >= SB * j && r < SB * (j + 1)) ? Vi[j * SB * VLD + (r - SB * j) * VLD + c] : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = j + 1; i < 4; ++i) {
        doublex4 acc = doublex4{0.0, 0.0, 0.0, 0.0};
        for (int m = j; m < i; ++m) acc = pq16(acc, T + (SB * i) * ULD + SB * m, ULD, O + (SB * m) * ULD + SB * j, ULD, 1.0, lane);
        store_d16(scratch[j], VLD, acc, lane);
        __builtin_amdgcn_wave_barrier();
        acc = pq16(doublex4{0.0, 0.0, 0.0, 0.0}, Vi + i * SB * VLD, VLD, scratch[j], VLD, -1.0, lane);
        __builtin_amdgcn_wave_barrier();
        store_d16(O + (SB * i) * ULD + SB * j, ULD, acc, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

// (only the inverse of the factor is read again -- by the next block column's workgroups and by the backward substitution)
__device__ __forceinline__ void publish_diag2(double *__restrict__ Ld, const double *O)
{
    for (int e = threadIdx.x; e < CB * CB; e += 256) Ld[LINV_OFF + e] = O[(e / CB) * ULD + (e % CB)];
}

// this wave's 16-row strip of  P (64 x 64, LDS ULD) * Q' (Q 64 x 64, LDS ULD): four 16 x 16 outputs, K = 64
template <bool NEGATE = false>
__device__ __forceinline__ void strip_pqt64(doublex4 (&acc)[4], const double *P, const double *Q, int wave, int lane)
{
    const double *ap = P + (16 * wave + (lane & 15)) * ULD + (lane >> 4);
    const double *bp = Q + (lane & 15) * ULD + (lane >> 4);
#pragma unroll 4
    for (int kk = 0; kk < CB / 4; ++kk) {
        const double a = NEGATE ? -ap[4 * kk] : ap[4 * kk];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[16 * cb * ULD + 4 * kk], acc[cb], 0, 0, 0);
    }
}

// Block column k in one launch (k = -1: the first one, nothing to subtract yet).  Workgroup (i, j), k < j <= i <= nb (block row
// nb = the right-hand side; j <= nb - 1), reads the factor tiles X_ik, X_jk from W2 and forms  C_ij - X_ik X_jk'.  Then
//   (k+1, k+1), blockIdx 0:  finishes the next diagonal tile in LDS, factors it, inverts the factor, publishes both, raises
//                            ready[k+1];
//   (i, k+1), i > k+1:       its tile is the final A_i,k+1: it waits for ready[k+1] (blockIdx 0 was dispatched before it, so the
//                            wait cannot deadlock), multiplies by L_k+1,k+1^-T and stores the factor tile X_i,k+1 to W2 -- the
//                            "triangular solve" of the next block column rides in the tail of this launch;
//   everything else:         writes C_ij back to W.
__global__ __launch_bounds__(256) void chol2_step_kernel(double *__restrict__ W, double *__restrict__ W2, double *__restrict__ Ldiag, int ld, int nb,
                                                         int k, int *__restrict__ ready, double *__restrict__ scal)
{
    __shared__ double Xi[CB * ULD];
    __shared__ double Xj[CB * ULD];
    __shared__ double Vi[4 * SB * VLD];
    __shared__ double scratch[4][SB * VLD];
    __shared__ double rd[CB];
    __shared__ int fail;
    const int m = nb - k - 1;  // square trailing block rows
    int t = blockIdx.x, i, j;
    // tile order: block column k + 1 first (its workgroups end with a wait and a second product: they must not be dispatched in the
    // last round), i = k+1 .. nb; then the rest of the trailing triangle row by row, then the rest of the right-hand side's row
    if (k < 0) { i = t; j = 0; }                        // the first launch only has block column 0 to do
    else if (t <= m) { i = k + 1 + t; j = k + 1; }
    else {
        t -= m + 1;
        const int tri = (m - 1) * m / 2;                // tiles (i, j), k + 2 <= j <= i <= nb - 1
        if (t < tri) {
            int ri = 0, rj = t;
            while (rj > ri) { rj -= ri + 1; ++ri; }
            i = k + 2 + ri; j = k + 2 + rj;
        } else {
            i = nb; j = k + 2 + (t - tri);
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // C = A_ij - X_i X_j' in the D layout (this wave: rows 16 wave + (lane >> 4) + 4 g, columns 16 cb + (lane & 15)): the
    // accumulators start from A_ij -- its loads are in flight together with the X tiles' -- and the product is subtracted
    doublex4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
            acc[cb][g] = (i != j || c <= r) ? W[(size_t)(i * CB + r) * ld + j * CB + c] : 0.0;
        }
    if (tid == 0) fail = 0;
    if (k >= 0) {
        for (int e = tid; e < CB * CB; e += 256) {
            const int r = e / CB, c = e % CB;
            Xi[r * ULD + c] = W2[(size_t)(i * CB + r) * ld + k * CB + c];
            if (j != i) Xj[r * ULD + c] = W2[(size_t)(j * CB + r) * ld + k * CB + c];
        }
        __syncthreads();
        strip_pqt64<true>(acc, Xi, (j != i) ? Xj : Xi, wave, lane);
        if (i == j) {                // (the part above the diagonal of a diagonal tile is not stored)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
                    if (c > r) acc[cb][g] = 0.0;
                }
        }
    }
    if (j != k + 1) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
                if (i != j || c <= r) W[(size_t)(i * CB + r) * ld + j * CB + c] = acc[cb][g];
            }
        return;
    }
#ifdef ESFM_CHOL_PROFILE
    long long tq[8]; int nq = 0;
#define STEP_MARK() do { if (k == 40) tq[nq++] = wall_clock64(); } while (0)
#else
#define STEP_MARK() do { } while (0)
#endif
    STEP_MARK();
    __syncthreads();                 // every wave is past its reads of Xi / Xj
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) store_d16(Xi + (16 * wave) * ULD + 16 * cb, ULD, acc[cb], lane);
    if (i == k + 1) {
        // the next diagonal tile: factor, invert, publish
        __syncthreads();
        STEP_MARK();
        tile_potrf64(Xi, Vi, rd, &fail);
        STEP_MARK();
        inv64(Xj, Xi, Vi, scratch);
        __syncthreads();
        STEP_MARK();
        publish_diag2(Ldiag + (size_t)(k + 1) * LSLOT, Xj);
        if (tid == 0 && fail) scal[SC_CHOL_FAIL] = 1.0;
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&ready[k + 1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        STEP_MARK();
#ifdef ESFM_CHOL_PROFILE
        if (k == 40 && tid == 0) printf("step40 wg0 [10ns]: start..update %lld potrf %lld inv %lld publish %lld (abs end %lld)\n", tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[4]);
#endif
        return;
    }
    // a tile of the next block column: X_i,k+1 = C L_k+1,k+1^-T once the inverse is there
    if (tid == 0) {
        long spins = 0;
        // relaxed polls (an acquire per poll invalidates caches the other workgroups are working from), one acquire at the end
        while (__hip_atomic_load(&ready[k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1L << 22)) { scal[SC_CHOL_FAIL] = 1.0; break; }     // never seen; keeps a broken launch from hanging the device
        }
    }
    __syncthreads();
    STEP_MARK();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    {
        const double *Lk = Ldiag + (size_t)(k + 1) * LSLOT + LINV_OFF;
        for (int e = tid; e < CB * CB; e += 256) Xj[(e / CB) * ULD + (e % CB)] = Lk[e];
    }
    __syncthreads();
    doublex4 x[4] = {doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}};
    strip_pqt64<false>(x, Xi, Xj, wave, lane);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
            W2[(size_t)(i * CB + r) * ld + (k + 1) * CB + c] = x[cb][g];
        }
#ifdef ESFM_CHOL_PROFILE
    STEP_MARK();
    if (k == 40 && tid == 0 && blockIdx.x == 1) printf("step40 wg1 [10ns]: update-done..flag seen %lld, tail %lld (abs flag seen %lld, end %lld)\n", tq[1] - tq[0], tq[2] - tq[1], tq[1], tq[2]);
#endif
#undef STEP_MARK
}

// The whole backward substitution L' y = z.  z_b = row 0 of the factor's tile (nb, b) in W2; ybuf: nb * CB doubles; flags: nb ints,
// zeroed before the launch.
__global__ __launch_bounds__(256) void chol2_back_kernel(BADev d, const double *__restrict__ W2, const double *__restrict__ Ldiag, int ld, int nb,
                                                         double *__restrict__ ybuf, int *__restrict__ flags)
{
    __shared__ double z[CB];
    __shared__ double y[CB];
    __shared__ double part[4][CB];
    __shared__ int gave_up;
    const int tid = threadIdx.x, t = tid & 63, g = tid >> 6;
    const int b = nb - 1 - (int)blockIdx.x;
    if (tid < CB) z[tid] = W2[(size_t)nb * CB * ld + b * CB + tid];
    if (tid == 0) gave_up = 0;
    __syncthreads();
    for (int i = nb - 1; i > b; --i) {
        // the tile's loads are in flight while the workgroup waits for y_i
        double l[CB / 4];
#pragma unroll
        for (int q = 0; q < CB / 4; ++q) l[q] = W2[(size_t)(i * CB + g + 4 * q) * ld + b * CB + t];
        if (tid == 0) {
            long spins = 0;
            while (__hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1L << 24)) { gave_up = 1; break; }        // never seen; keeps a broken launch from hanging the device
            }
        }
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (tid < CB) y[tid] = ybuf[i * CB + tid];
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < CB / 4; ++q) s = fma(l[q], y[g + 4 * q], s);
        part[g][t] = s;
        __syncthreads();
        if (tid < CB) z[tid] -= ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
        __syncthreads();
    }
    // y_b = Linv_bb' z_b
    {
        const double *Lb = Ldiag + (size_t)b * LSLOT + LINV_OFF;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < CB / 4; ++q) { const int r = g + 4 * q; s = fma(Lb[r * CB + t], z[r], s); }
        part[g][t] = s;
        __syncthreads();
        if (tid < CB) {
            const double yb = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
            ybuf[b * CB + tid] = yb;
            const bool fail = d.scal[SC_CHOL_FAIL] != 0.0 || gave_up;
            if (b * CB + tid < 6 * d.n_cam) d.y_c[b * CB + tid] = fail ? 0.0 : yb;
            if (gave_up && tid == 0) d.scal[SC_CHOL_FAIL] = 1.0;
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&flags[b], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
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

size_t ba_chol_large_doubles(int n_cam)
{
    const int n = 6 * n_cam, nb = (n + CB - 1) / CB;
    // W, W2 (the factor), the diagonal slots, y, flags
    return 2 * (size_t)(nb + 1) * CB * (size_t)(nb * CB) + (size_t)nb * LSLOT + (size_t)nb * CB + (size_t)nb + 1;
}

int ba_solve_reduced_large(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag)
{
    const int n = 6 * d.n_cam, nb = (n + CB - 1) / CB, ld = nb * CB;
    const size_t wsz = (size_t)(nb + 1) * CB * ld;
    double *W = d.chol, *W2 = W + wsz;
    double *Ldiag = W2 + wsz;
    double *ybuf = Ldiag + (size_t)nb * LSLOT;
    int *flags = reinterpret_cast<int *>(ybuf + (size_t)nb * CB);
    const long long tot = (long long)wsz;
    hipLaunchKernelGGL(chol_assemble_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d, W, ld, nb, radius, min_diag, max_diag);
    ESFM_HIP_TRY(hipGetLastError());
    ESFM_HIP_TRY(hipMemsetAsync(flags, 0, sizeof(int) * 2 * (size_t)nb, st));     // y-ready and factor-ready
    for (int k = -1; k < nb - 1; ++k) {
        // block column k's trailing update; its first workgroup factors and inverts the next diagonal tile, the workgroups of the
        // next block column turn their tiles into factor tiles (see chol2_step_kernel).  k = -1 starts the chain.
        const int m = nb - k - 1;
        hipLaunchKernelGGL(chol2_step_kernel, dim3(k < 0 ? nb + 1 : m * (m + 1) / 2 + m), dim3(256), 0, st, W, W2, Ldiag, ld, nb, k, flags + nb, d.scal);
    }
    ESFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(chol2_back_kernel, dim3(nb), dim3(256), 0, st, d, W2, Ldiag, ld, nb, ybuf, flags);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm
