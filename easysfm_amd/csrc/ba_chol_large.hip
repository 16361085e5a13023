// Dense solve of the reduced camera system: DENSE_SCHUR's factorisation step (reference cpp_code/src/ba.cpp:201, Ceres' Eigen
// LLT [upstream]).
//
// n = 6 n_cam > 176 (BASELINE config 5: 512 cameras, n = 3072): blocked right-looking Cholesky on 64 x 64 f64 tiles across the
// whole chip, the right-hand side carried as an extra block row (forward substitution for free), then the backward substitution.
//   chol_assemble_kernel   W = F'F + D_c^2 + S_schur (lower), rhs row = F'r + rhs_corr, identity padding
//   chol2_step_kernel(k)   ONE launch per block column: trailing update C_ij -= X_ik X_jk' on v_mfma_f64_16x16x4_f64; its first
//                          workgroup factors the next diagonal tile and inverts the factor (tile_potrf64_inv); the workgroups
//                          of the next block column wait for that inverse and turn their tiles into factor tiles X_i,k+1
//   chol2_back_kernel      the whole backward substitution in one launch, solution blocks handed on through memory (a block is its own flag)
// n <= 176 (BASELINE's BA-25: n = 150): ba_chol_small_kernel, the whole solve and the camera step in one workgroup.
// Both sit on potrf16_fused_*: one wave factors a 16 x 16 block and inverts the factor in one pass of generated, scheduled asm.
#include "ba_kernels.hpp"
#include "ba_chol_tile.hpp"
namespace esfm {

// FIXED: d.red still holds the Schur kernels' fixed-point integers (entry (i, j) scaled by 2^(60 - qexp[i] - qexp[j]), the right-hand
// side by 2^(60 - qexp[j] - rhs_exp)): converted here as ba_schur_to_double_kernel would have, and what has been read is cleared, so
// that the next LM iteration's Schur kernels start from zeros without a memset of the whole buffer.
template <bool FIXED>
__global__ __launch_bounds__(256) void chol_assemble_kernel(BADev d, double *__restrict__ W, int ld, int nb, double radius,
                                                            double min_diag, double max_diag, double *__restrict__ ybuf, int *__restrict__ flags, int rhs_exp)
{
    const int n = 6 * d.n_cam;
    const long long rows = (long long)(nb + 1) * CB;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < (long long)nb * CB) reinterpret_cast<unsigned long long *>(ybuf)[e] = kYPending;
    if (e < 5 * nb + 1) flags[e] = 0;                     // chol3_kernel's flags (a memset launch of its own until round 3)
    if (e >= rows * ld) return;
    const int i = (int)(e / ld), j = (int)(e % ld);
    double v = 0.0;
    unsigned long long *redq = reinterpret_cast<unsigned long long *>(d.red);
    if (i < n) {
        if (FIXED && j < n && j < 6 * (i / 6 + 1)) {     // everything the Schur kernels can have written in this row (full diagonal blocks)
            const unsigned long long q = redq[(size_t)i * n + j];
            if (j <= i && q != 0ull) v = fx64_to_double(q, kFxBits - d.qexp[i] - d.qexp[j]);
            if (q != 0ull) redq[(size_t)i * n + j] = 0ull;
        }
        if (j <= i) {
            if (!FIXED) v = d.red[(size_t)i * n + j];
            if (i / 6 == j / 6) {
                const int c = i / 6;
                v += d.camacc[36 * (size_t)c + 6 * (i % 6) + (j % 6)];
                if (i == j) v += fmin(fmax(d.camacc[36 * (size_t)c + 7 * (i % 6)], min_diag), max_diag) / radius;
            }
        }
    } else if (i < nb * CB) {
        v = (i == j) ? 1.0 : 0.0;  // padding rows: identity
    } else if (i == nb * CB) {
        if (FIXED) {
            if (j < n) {
                const unsigned long long q = redq[(size_t)n * n + j];
                v = d.camacc[36 * (size_t)d.n_cam + j] + fx64_to_double(q, kFxBits - d.qexp[j] - rhs_exp);
                redq[(size_t)n * n + j] = 0ull;
            }
        } else {
            v = (j < n) ? d.camacc[36 * (size_t)d.n_cam + j] + d.red[(size_t)n * n + j] : 0.0;  // rhs row
        }
    }
    W[(size_t)i * ld + j] = v;
}

// The factorisation as ONE launch, a dataflow over tiles.  Workgroup = tile (i, j), 0 <= j <= i <= nb (block row nb = the
// right-hand side), dispatched column by column (diagonal tile first):
//   * its tile C_ij stays in the MFMA accumulators while it subtracts  X_i,kk X_j,kk'  for the columns kk left of it, each awaited
//     through a flag (xready[i][kk]);
//   * then it becomes a factor tile: it waits for L_jj^-1 (ready[j]), multiplies by it and stores X_ij to W2 (raises xready[i][j]);
//   * the FIRST tile below the diagonal, (j+1, j), goes on: it holds X_j+1,j -- the last thing the next diagonal tile is waiting
//     for -- so it subtracts X X' from that tile itself (fetched beforehand: tile (j+1, j+1) stops one step early and publishes
//     its partial sum, dpart[j+1]), factors it and inverts the factor (tile_potrf64_inv) and raises ready[j+1].  The pivot chain
//     thus runs  ... -> L_jj^-1 -> X_j+1,j -> L_j+1,j+1^-1 -> ...  inside one workgroup per column, with ONE flag between columns
//     (31 us -> 22 us per column against handing X_j+1,j to the diagonal tile's own workgroup: a flag, a 32 KB coherent read).
// Every wait is for a workgroup with a smaller blockIdx, so the chain cannot deadlock whatever part of the grid is resident.
// History: one trsm + one update launch per block column (round 1, 145 launches) 4.46 ms; one launch per block column 1.97 ms --
// of each 31 us step ~15 us were two agent-scope releases (L2 write-backs, see st_coh); this kernel 1.5 ms before the merge above.

__global__ __launch_bounds__(256) void chol3_kernel(double *__restrict__ W, double *__restrict__ W2, double *__restrict__ Ldiag, int ld, int nb,
                                                    int *__restrict__ ready, int *__restrict__ xcount, int *__restrict__ dpart, int *__restrict__ rpart,
                                                    double *__restrict__ scal)
{
    __shared__ __attribute__((aligned(16))) double Xi[CB * ULD];
    __shared__ __attribute__((aligned(16))) double Xj[CB * ULD];
    __shared__ double Vi[4 * SB * VLD];
    __shared__ int fail;
    int t = blockIdx.x, i = -1, j = -1;
    for (int c = 0; c < nb; ++c) {
        const int cnt = nb - c + 1;
        if (t < cnt) { i = c + t; j = c; break; }
        t -= cnt;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) fail = 0;
    const bool chain_wg = (i == j + 1 && i < nb);
    if (chain_wg) CHOL_T(j, 0);
    __shared__ int seen;
    auto wait_flag = [&](const int *f, bool urgent, int at_least = 1) {  // thread 0 polls (relaxed) until *f >= at_least, then a workgroup-scope acquire
        if (tid == 0) {
            long spins = 0;
            while ((seen = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < at_least) {
                if (urgent) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(100);     // (hundreds of resident pollers: the far ones back off)
                if (++spins > (1L << 27)) { scal[SC_CHOL_FAIL] = 1.0; break; }     // never seen; keeps a broken launch from hanging the device
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto factor_and_publish = [&](double *T, double *O, int c) {      // T: the finished diagonal tile of column c (LDS); O: scratch tile
        tile_potrf64_inv(T, O, Vi, &fail, Ldiag + (size_t)c * LSLOT + LINV_OFF, &rpart[c], c > 0 ? c - 1 : 63);
        CHOL_T(c - 1, 7);
        if (tid == 0 && fail) scal[SC_CHOL_FAIL] = 1.0;
        publish_flag(&ready[c]);
    };
    // C = A_ij in the D layout (this wave: rows 16 wave + (lane >> 4) + 4 g, columns 16 cb + (lane & 15))
    doublex4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
            acc[cb][g] = (i != j || c <= r) ? W[(size_t)(i * CB + r) * ld + j * CB + c] : 0.0;
        }
    const bool diag = i == j;
    const int kk_end = diag ? j - 1 : j;               // a diagonal tile leaves its last step to the workgroup that produces X_j,j-1
    // xcount[r] = number of factor tiles of block row r that are in W2 (they appear in column order): one poll per row tells how
    // many steps can run without waiting -- a tile dispatched late has dozens of columns to catch up with.  The factor tiles are
    // read with plain loads: nobody reads one before its flag, so no cache holds an older copy, and the many readers share L2.
    // The loop is software-pipelined through registers: the loads of step kk + 1 are in flight during the MFMAs of step kk
    // (11 us -> ~6 us per step; with one workgroup per CU nothing else hides the latency).
    int have = 0;
    double rx[CB * CB / 256], ry[CB * CB / 256];
    auto ensure = [&](int kk, bool block) -> bool {   // are X_i,kk and X_j,kk there?  (block: wait for them)
        if (kk < have) return true;
        const bool urgent = j - kk <= 2;               // the next columns' tiles are the critical chain
        wait_flag(&xcount[i], urgent, block ? kk + 1 : 0);
        have = seen;
        if (!diag) { __syncthreads(); wait_flag(&xcount[j], urgent, block ? kk + 1 : 0); have = min(have, seen); }
        __syncthreads();
        return kk < have;
    };
    auto fetch = [&](int kk) {
#pragma unroll
        for (int q = 0; q < CB * CB / 256; ++q) {
            const int e = tid + 256 * q, r = e / CB, c = e % CB;
            rx[q] = W2[(size_t)(i * CB + r) * ld + kk * CB + c];
            if (!diag) ry[q] = W2[(size_t)(j * CB + r) * ld + kk * CB + c];
        }
    };
    bool fetched = false;
    if (kk_end > 0) { ensure(0, true); fetch(0); fetched = true; }
    for (int kk = 0; kk < kk_end; ++kk) {
        if (!fetched) { ensure(kk, true); fetch(kk); }
#pragma unroll
        for (int q = 0; q < CB * CB / 256; ++q) {
            const int e = tid + 256 * q, r = e / CB, c = e % CB;
            Xi[r * ULD + c] = rx[q];
            if (!diag) Xj[r * ULD + c] = ry[q];
        }
        __syncthreads();
        fetched = kk + 1 < kk_end && ensure(kk + 1, false);
        if (fetched) fetch(kk + 1);
        strip_pqt64<true>(acc, Xi, diag ? Xi : Xj, wave, lane);
        __syncthreads();               // every wave is past its reads of Xi / Xj
    }
    if (diag) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
                if (c > r) acc[cb][g] = 0.0;            // (the part above the diagonal is not stored)
            }
        if (j == 0) {                                   // nothing to its left: factor it here
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) store_d16(Xi + (16 * wave) * ULD + 16 * cb, ULD, acc[cb], lane);
            __syncthreads();
            factor_and_publish(Xi, Xj, 0);
            return;
        }
        // the partial sum goes back to W for workgroup (j, j-1) to finish
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
                st_coh(&W[(size_t)(i * CB + r) * ld + j * CB + c], acc[cb][g]);
            }
        publish_flag(&dpart[j]);
        return;
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) store_d16(Xi + (16 * wave) * ULD + 16 * cb, ULD, acc[cb], lane);
    // the first tile below the diagonal also finishes the next diagonal tile: fetch its partial sum while waiting for L_jj^-1
    const bool next_diag = chain_wg;
    if (next_diag) CHOL_T(j, 1);
    // The ten lower 16 x 16 blocks of the next diagonal tile over the four waves as 3 + 3 + 3 + 1 (a row strip per wave would be
    // 1 + 2 + 3 + 4: the slowest wave sets the pace, 48 matrix instructions instead of 64): wave w's q-th block is (dbi, dbj)[w][q].
    const int dbi[3] = {wave == 3 ? 3 : wave, wave == 2 ? 2 : (wave == 1 ? 1 : 3), wave == 2 ? 2 : 3};
    const int dbj[3] = {wave == 3 ? 3 : 0, wave == 3 ? 3 : (wave == 0 ? 0 : 1), wave == 3 ? 3 : (wave == 0 ? 1 : 2)};
    const int dnb = wave == 3 ? 1 : 3;      // (wave 3 runs (3, 3) three times and keeps one: no branch around a matrix instruction)
    doublex4 dacc[3] = {doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}};
    if (next_diag) {
        wait_flag(&dpart[i], false);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * dbi[q] + (lane >> 4) + 4 * g, c = 16 * dbj[q] + (lane & 15);
                if (q < dnb) dacc[q][g] = ld_coh(&W[(size_t)(i * CB + r) * ld + i * CB + c]);
            }
    }
    // X_ij = C L_jj^-T once the inverse is there
    if (next_diag) CHOL_T(j, 2);
    {
        // all 32 KB in flight at once (as a loop the compiler made it sixteen round trips, load -> wait -> LDS store: 4 us of the
        // 25 us a block column takes, on the critical chain) -- and the first 16 KB (rows 0..31: thread t's q-th load is in row
        // 8 q + t / 32) as soon as the factoring workgroup has let go of them, a pivot chain before the tile is finished
        const double2 *Lk = reinterpret_cast<const double2 *>(Ldiag + (size_t)j * LSLOT + LINV_OFF);
        double2 lv[CB * CB / 512];
        wait_flag(&rpart[j], true, 3);
#pragma unroll
        for (int q = 0; q < 4; ++q) lv[q] = Lk[tid + 256 * q];
        wait_flag(&ready[j], true);
        if (next_diag) CHOL_T(j, 3);
#pragma unroll
        for (int q = 4; q < CB * CB / 512; ++q) lv[q] = Lk[tid + 256 * q];
#pragma unroll
        for (int q = 0; q < CB * CB / 512; ++q) {
            const int e = 2 * (tid + 256 * q);
            *reinterpret_cast<double2 *>(&Xj[(e / CB) * ULD + (e % CB)]) = lv[q];
        }
    }
    __syncthreads();
    if (next_diag) CHOL_T(j, 4);
    doublex4 x[4] = {doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}};
    strip_pqt64<false, true>(x, Xi, Xj, wave, lane);
    if (!next_diag) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
                st_coh(&W2[(size_t)(i * CB + r) * ld + j * CB + c], x[cb][g]);
            }
        publish_flag(&xcount[i], j + 1);
        return;
    }
    // (j+1, j): X goes to LDS as the operand of the next diagonal tile's last step, and to W2 for everybody else (its flag is
    // raised after the factorisation has been started -- nobody on the critical chain waits for it)
    __syncthreads();                                    // every wave is past its reads of Xi (C) and Xj (L^-1)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) store_d16(Xi + (16 * wave) * ULD + 16 * cb, ULD, x[cb], lane);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
            st_coh(&W2[(size_t)(i * CB + r) * ld + j * CB + c], x[cb][g]);
        }
    __syncthreads();
    {
        const double *xp = Xi + (lane & 15) * ULD + (lane >> 4);
#pragma unroll 4
        for (int kk = 0; kk < CB / 4; ++kk) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
                dacc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xp[16 * dbi[q] * ULD + 4 * kk], xp[16 * dbj[q] * ULD + 4 * kk], dacc[q], 0, 0, 0);
        }
    }
    // (the tile above the diagonal is zero: Xj held L_jj^-1, whose upper blocks are zeros, and the waves rewrite the lower ones)
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if ((lane & 15) > (lane >> 4) + 4 * g) {
            dacc[0][g] = wave == 0 || wave == 3 ? 0.0 : dacc[0][g];          // blocks (0,0) and (3,3)
            dacc[1][g] = wave == 1 ? 0.0 : dacc[1][g];                        // (1,1)
            dacc[2][g] = wave == 2 ? 0.0 : dacc[2][g];                        // (2,2)
        }
    __syncthreads();                                    // every wave is past its reads of Xi and of Xj (L^-1)
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q < dnb) store_d16(Xj + (16 * dbi[q]) * ULD + 16 * dbj[q], ULD, dacc[q], lane);
    __syncthreads();
    CHOL_T(j, 5);
    publish_flag(&xcount[i], j + 1);                    // (the X stores above have long been acknowledged)
    CHOL_T(j, 6);
    factor_and_publish(Xj, Xi, i);
    CHOL_T(j, 9);
}

// The whole backward substitution L' y = z.  z_b = row 0 of the factor's tile (nb, b) in W2; ybuf: nb * CB doubles, every one the
// "pending" pattern before the launch (chol_assemble_kernel).  A solution block is its own flag: the 64 threads that need y_i poll
// its 64 words until none is pending -- one memory round trip per link of the chain instead of two (flag, then data: 2.35 us per
// block, 113 us for BA-512's 48; now see DESIGN.md), and the publisher just stores.  8-byte stores are single transactions, so a
// word is either pending or final.
__global__ __launch_bounds__(256) void chol2_back_kernel(BADev d, const double *__restrict__ W2, const double *__restrict__ Ldiag, int ld, int nb,
                                                         double *__restrict__ ybuf)
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
        if (tid < CB) {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(ybuf) + i * CB + tid;
            unsigned long long v;
            long spins = 0;
            while ((v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kYPending) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1L << 24)) { gave_up = 1; v = 0ull; break; }    // never seen; keeps a broken launch from hanging the device
            }
            y[tid] = __longlong_as_double((long long)v);
        }
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
            double yb = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
            const bool fail = d.scal[SC_CHOL_FAIL] != 0.0 || gave_up;
            if (__double_as_longlong(yb) == (long long)kYPending) yb = __longlong_as_double(0x7ff8000000000000ll);   // (never: keep the chain alive anyway)
            st_coh(&ybuf[b * CB + tid], yb);
            if (b * CB + tid < 6 * d.n_cam) d.y_c[b * CB + tid] = fail ? 0.0 : yb;
            if (gave_up && tid == 0) d.scal[SC_CHOL_FAIL] = 1.0;
        }
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
    // W, W2 (the factor), the diagonal slots, y, flags (y-ready, inverse-ready, partial-diagonal-ready, inverse-rows-0..31-ready: nb each; factor tiles per block row: nb + 1 ints)
    return 2 * (size_t)(nb + 1) * CB * (size_t)(nb * CB) + (size_t)nb * LSLOT + (size_t)nb * CB + (size_t)(5 * nb + 2) / 2 + 1;
}

int ba_solve_reduced_large(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag)
{
    const int n = 6 * d.n_cam, nb = (n + CB - 1) / CB, ld = nb * CB;
    const size_t wsz = (size_t)(nb + 1) * CB * ld;
    double *W = d.chol, *W2 = W + wsz;
    double *Ldiag = W2 + wsz;
    double *ybuf = Ldiag + (size_t)nb * LSLOT;
    int *flags = reinterpret_cast<int *>(ybuf + (size_t)nb * CB);        // [nb] (unused: a solution block is its own flag) | [nb] inverse | [nb] partial diagonal | [nb + 1] factor tiles per row | [nb] rows 0..31 of the inverse
    const long long tot = (long long)wsz;
    if (d.parts->red_fixed) {
        hipLaunchKernelGGL(chol_assemble_kernel<true>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d, W, ld, nb, radius, min_diag, max_diag, ybuf, flags,
                           d.parts->red_rhs_exp);
        d.parts->red_fixed = false; d.parts->red_clean = true;
    } else {
        hipLaunchKernelGGL(chol_assemble_kernel<false>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d, W, ld, nb, radius, min_diag, max_diag, ybuf, flags, 0);
    }
    ESFM_HIP_TRY(hipGetLastError());
    const long long tiles = (long long)nb * (nb + 1) / 2 + nb;           // (i, j), 0 <= j <= i <= nb, j <= nb - 1
    hipLaunchKernelGGL(chol3_kernel, dim3((unsigned)tiles), dim3(256), 0, st, W, W2, Ldiag, ld, nb, flags + nb, flags + 3 * nb, flags + 2 * nb, flags + 4 * nb + 1, d.scal);
    ESFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(chol2_back_kernel, dim3(nb), dim3(256), 0, st, d, W2, Ldiag, ld, nb, ybuf);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm
