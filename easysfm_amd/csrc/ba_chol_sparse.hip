// Structure-aware solve of the reduced camera system (round 5).  DENSE_SCHUR (reference cpp_code/src/ba.cpp:201) factors all of
// S = F'F + D^2 - sum_p (F'E) M^-1 (E'F); but block (a, b) of S is non-zero only if cameras a and b observe a common point, and the
// reference's own problem construction -- one residual block per observation, ba.cpp:140-151 -- fixes that structure before the
// first iteration.  ba_sparse_plan.cpp orders the cameras by nested dissection of the co-visibility graph and lists the 64 x 64
// tiles of the symbolic fill; here the dense dataflow factorisation of ba_chol_large.hip is restated over that tile list:
//   chol_sparse_assemble_kernel   W = P (F'F + D_c^2 + S_schur) P' tile by tile (identity padding between supernodes), right-hand side
//                                 row; reads the Schur kernels' fixed-point integers and leaves zeros behind (one rank), or the
//                                 all-reduced packed blocks (several ranks)
//   chol_sparse_kernel            ONE launch, workgroup = tile of the fill: subtract X_I,K X_J,K' for the columns K both block rows
//                                 have, then X_IJ = C L_JJ^-T; the last off-diagonal tile of block row I goes on to finish, factor and
//                                 invert diagonal tile (I, I).  Independent subtrees of the elimination tree run side by side: on
//                                 BASELINE config 5 (512 cameras of a closed loop) the dependency chain is 8 tile columns, not 48
//   chol_sparse_back_kernel       backward substitution down the elimination tree, solution blocks as their own flags
//   ba_sparse_pack_kernel         (several ranks) only the co-visible camera blocks travel: fixed point -> f64, packed
// Skipped tiles are exact zeros of the dense factorisation in the same elimination order, so the result differs from the dense
// path's only by the ORDER of the eliminations (a symmetric permutation), i.e. in rounding.  Every wait is for a workgroup
// dispatched earlier; the schedule is fixed, the result bit-reproducible.
#define ESFM_CHOL_NO_TRACE
#include "ba_chol_tile.hpp"
#include "ba_chol_sparse.hpp"

#include <algorithm>
#include <cstdlib>

namespace esfm {

#ifdef ESFM_SPARSE_TRACE
// timing-only build (scratch/build_variant_sparse.sh NAME -DESFM_SPARSE_TRACE): every workgroup of chol_sparse_kernel leaves
// s_memrealtime stamps (10 ns ticks) at its stages; scratch/sparse_trace.py reads them through esfm_debug_sparse_trace
__device__ unsigned long long g_sparse_trace[4096 * 8];
#define SP_T(q) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_sparse_trace[blockIdx.x * 8 + (q)] = wall_clock64(); } while (0)
extern "C" int esfm_debug_sparse_trace(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sparse_trace), sizeof(g_sparse_trace)); }
#else
#define SP_T(q) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// element (hi, lo) of the reduced system, hi >= lo original unknowns, from the fixed-point buffer; what is read is cleared
__device__ __forceinline__ double take_fixed(unsigned long long *redq, size_t idx, int sh)
{
    const unsigned long long q = redq[idx];
    if (q == 0ull) return 0.0;
    redq[idx] = 0ull;
    return fx64_to_double(q, sh);
}

// position of block (a, b), b <= a, in the packed exchange buffer: the lower neighbours of a (and a itself) ascending
__device__ __forceinline__ int packed_block(const int32_t *__restrict__ cov_start, const int32_t *__restrict__ cov_adj, int a, int b)
{
    int lo = cov_start[a], hi = cov_start[a + 1] - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cov_adj[mid] < b) lo = mid + 1; else hi = mid;
    }
    return cov_adj[lo] == b ? lo : -1;
}

// PACKED: the blocks come from the all-reduced exchange buffer (doubles); else from d.red in fixed point (cleared on the way)
template <bool PACKED>
__global__ __launch_bounds__(256) void chol_sparse_assemble_kernel(BADev d, SparseDev sp, double radius, double min_diag, double max_diag, int rhs_exp,
                                                                   const double *__restrict__ packed)
{
    // four workgroups per tile (16 rows each): the kernel is a few dependent loads per element, and 249 workgroups (BA-512) left most
    // of the chip without one (28.6 us; now see DESIGN.md)
    const int slot = blockIdx.x >> 2, quarter = blockIdx.x & 3, tid = threadIdx.x;
    const int I = sp.tile_ij[2 * slot], J = sp.tile_ij[2 * slot + 1];
    const int n = 6 * d.n_cam, nb = sp.nb;
    if (tid == 0 && quarter == 0) sp.xready[slot] = 0;
    if (I == J && quarter == 0) {
        if (tid < CB) reinterpret_cast<unsigned long long *>(sp.ybuf)[J * CB + tid] = kYPending;
        if (tid == 64) sp.ready[J] = 0;
        if (tid == 65) sp.rpart[J] = 0;
    }
    unsigned long long *redq = reinterpret_cast<unsigned long long *>(d.red);
    double *W = sp.W + (size_t)slot * (CB * CB);
    const int c = tid & 63;
    const int ic = sp.col_src[J * CB + c];
#pragma unroll
    for (int q = 0; q < CB * CB / 1024; ++q) {
        const int r = 16 * quarter + (tid >> 6) + 4 * q;
        double v = 0.0;
        if (I == nb) {                                         // right-hand side: row 0 of the tile
            if (r == 0 && ic >= 0) {
                const double corr = PACKED ? packed[(size_t)36 * sp.n_blocks + ic] : take_fixed(redq, (size_t)n * n + ic, kFxBits - d.qexp[ic] - rhs_exp);
                v = d.camacc[36 * (size_t)d.n_cam + ic] + corr;
            }
        } else if (I != J || c <= r) {
            const int ia = sp.col_src[I * CB + r];
            if (ia < 0 || ic < 0) {
                v = (I == J && r == c) ? 1.0 : 0.0;             // identity padding
            } else {
                const int hi = max(ia, ic), lo = min(ia, ic), ch = hi / 6, cl = lo / 6;
                if (PACKED) {
                    const int k = packed_block(sp.cov_start, sp.cov_adj, ch, cl);
                    if (k >= 0) v = packed[(size_t)36 * k + 6 * (hi - 6 * ch) + (lo - 6 * cl)];
                } else {
                    v = take_fixed(redq, (size_t)hi * n + lo, kFxBits - d.qexp[hi] - d.qexp[lo]);
                    if (ch == cl && hi != lo) (void)take_fixed(redq, (size_t)lo * n + hi, 0);       // (the Schur kernels write whole diagonal blocks)
                }
                if (ch == cl) {
                    v += d.camacc[36 * (size_t)ch + 6 * (hi - 6 * ch) + (lo - 6 * cl)];
                    if (hi == lo) v += fmin(fmax(d.camacc[36 * (size_t)ch + 7 * (hi - 6 * ch)], min_diag), max_diag) / radius;
                }
            }
        }
        W[r * CB + c] = v;
    }
}

// Several ranks: this rank's contribution to the co-visible blocks (and the right-hand side), converted to f64 for the all-reduce;
// d.red is left all zeros.  One workgroup per 7 blocks (36 entries each); the last workgroups move the right-hand side.
__global__ __launch_bounds__(256) void ba_sparse_pack_kernel(BADev d, SparseDev sp, int rhs_exp, double *__restrict__ packed)
{
    const int n = 6 * d.n_cam;
    unsigned long long *redq = reinterpret_cast<unsigned long long *>(d.red);
    const int nblk_wg = (sp.n_blocks + 6) / 7;
    if ((int)blockIdx.x < nblk_wg) {
        const int k = 7 * blockIdx.x + threadIdx.x / 36, e = threadIdx.x % 36;
        if (threadIdx.x >= 252 || k >= sp.n_blocks) return;
        const int a = sp.cov_row[k], b = sp.cov_adj[k], ra = e / 6, ca = e % 6;
        const int i = 6 * a + ra, j = 6 * b + ca;
        double v = take_fixed(redq, (size_t)i * n + j, kFxBits - d.qexp[i] - d.qexp[j]);
        if (a == b && ca > ra) v = 0.0;                        // (only the lower part of a diagonal block is the matrix)
        packed[(size_t)36 * k + e] = v;
    } else {
        const int j = ((int)blockIdx.x - nblk_wg) * 256 + threadIdx.x;
        if (j < n) packed[(size_t)36 * sp.n_blocks + j] = take_fixed(redq, (size_t)n * n + j, kFxBits - d.qexp[j] - rhs_exp);
    }
}

// ---------------------------------------------------------------------------------------------
// The factorisation: one launch, workgroup = tile (I, J) of the fill in dispatch order (sp.wgs; column by column).
//   kind 1  diagonal tile of a column with nothing to its left: factor it, invert the factor, raise ready[J]
//   kind 0  C = A_IJ - sum_K X_I,K X_J,K' over its update list (each operand awaited through its tile's flag), then wait for
//           L_JJ^-1, X_IJ = C L_JJ^-T, store, raise xready[slot]
//   kind 3  the same for the right-hand side's row (block row nb)
//   kind 2  the LAST off-diagonal tile of block row I ("chain" workgroup): as kind 0, and then it finishes diagonal tile (I, I)
//           itself -- its operand list holds EVERY X_I,K, K < J, and each one is subtracted (X X') from the diagonal tile while it
//           sits in LDS for the own tile's update; its own X_IJ straight from LDS last -- factors it and inverts the factor: along a
//           chain of the elimination tree the next column's inverse is one workgroup away from this column's, as in chol3_kernel.
//           (First build: the diagonal tile's updates in a loop of their own AFTER the own tile's, in column order -- a separator's
//           workgroup then did a dozen long-available tiles behind the one it had really been waiting for: 217 us on BA-512, of
//           which 110 us were such queues, scratch/sparse_trace.py.)
__global__ __launch_bounds__(256) void chol_sparse_kernel(SparseDev sp, double *__restrict__ scal)
{
    __shared__ __attribute__((aligned(16))) double Xi[CB * ULD];
    __shared__ __attribute__((aligned(16))) double Xj[CB * ULD];
    __shared__ double Vi[4 * SB * VLD];
    __shared__ int fail;
    __shared__ int seen;
    const SparseWg w = sp.wgs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) fail = 0;
    SP_T(0);
    // thread 0 polls (relaxed) until every flag of the list reads >= at_least (block == false: looks once); then a workgroup-scope acquire
    auto wait2 = [&](const int *fa, const int *fb, int at_least, bool block) -> bool {
        if (tid == 0) {
            long spins = 0;
            int ok;
            while (true) {
                ok = __hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= at_least &&
                     (fb == nullptr || __hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= at_least);
                if (ok || !block) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1L << 27)) { scal[SC_CHOL_FAIL] = 1.0; ok = 1; break; }     // never seen; keeps a broken launch from hanging the device
            }
            seen = ok;
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const bool r = seen != 0;
        __syncthreads();
        return r;
    };
    auto factor_and_publish = [&](double *T, double *O, int c) {      // T: the finished diagonal tile of column c (LDS); O: scratch tile
        tile_potrf64_inv(T, O, Vi, &fail, sp.Ldiag + (size_t)c * LSLOT + LINV_OFF, &sp.rpart[c]);
        if (tid == 0 && fail) scal[SC_CHOL_FAIL] = 1.0;
        publish_flag(&sp.ready[c]);
    };
    const bool diag = w.kind == 1;
    // C = A_IJ in the D layout (this wave: rows 16 wave + (lane >> 4) + 4 g, columns 16 cb + (lane & 15))
    doublex4 acc[4];
    {
        const double *A = sp.W + (size_t)w.slot * (CB * CB);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
                acc[cb][g] = A[r * CB + c];                 // (a diagonal tile's upper part was assembled as zeros)
            }
    }
    // The ten lower 16 x 16 blocks of the next diagonal tile over the four waves as 3 + 3 + 3 + 1 (see chol3_kernel)
    const bool chain = w.kind == 2;
    const int dbi[3] = {wave == 3 ? 3 : wave, wave == 2 ? 2 : (wave == 1 ? 1 : 3), wave == 2 ? 2 : 3};
    const int dbj[3] = {wave == 3 ? 3 : 0, wave == 3 ? 3 : (wave == 0 ? 0 : 1), wave == 3 ? 3 : (wave == 0 ? 1 : 2)};
    const int dnb = wave == 3 ? 1 : 3;
    doublex4 dacc[3] = {doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}};
    auto diag_update = [&](const double *X) {            // dacc -= (X X') on this wave's blocks; X: 64 x 64 in LDS (ULD)
        const double *xp = X + (lane & 15) * ULD + (lane >> 4);
#pragma unroll 4
        for (int kk = 0; kk < CB / 4; ++kk) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
                dacc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xp[16 * dbi[q] * ULD + 4 * kk], xp[16 * dbj[q] * ULD + 4 * kk], dacc[q], 0, 0, 0);
        }
    };
    if (chain) {
        const double *A = sp.W + (size_t)w.dslot * (CB * CB);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = 16 * dbi[q] + (lane >> 4) + 4 * g, c = 16 * dbj[q] + (lane & 15);
                if (q < dnb) dacc[q][g] = A[r * CB + c];
            }
    }
    // Update loop over the operand list (in the plan's order of expected availability), software-pipelined through registers: the
    // loads of step u + 1 are in flight during the MFMAs of step u.  Step u brings X_I,K and -- second slot >= 0 -- X_J,K: the own
    // tile loses X_I,K X_J,K'; a chain workgroup's diagonal tile (I, I) loses X_I,K X_I,K' from the same LDS copy of X_I,K.
    {
        double2 rx[CB * CB / 512], ry[CB * CB / 512];      // 16-byte loads: thread t's q-th piece is elements 2 (t + 256 q), 2 (t + 256 q) + 1 of the tile
        auto avail = [&](int u, bool block) {
            const int sb = sp.upd[2 * u + 1];
            return wait2(&sp.xready[sp.upd[2 * u]], sb >= 0 ? &sp.xready[sb] : nullptr, 1, block);
        };
        auto fetch = [&](int u) {
            const int sb = sp.upd[2 * u + 1];
            const double2 *xa = reinterpret_cast<const double2 *>(sp.W2 + (size_t)sp.upd[2 * u] * (CB * CB));
            const double2 *xb = reinterpret_cast<const double2 *>(sp.W2 + (size_t)max(sb, 0) * (CB * CB));
#pragma unroll
            for (int q = 0; q < CB * CB / 512; ++q) { rx[q] = xa[tid + 256 * q]; if (sb >= 0) ry[q] = xb[tid + 256 * q]; }
        };
        bool fetched = false;
        for (int u = w.upd0; u < w.upd1; ++u) {
            if (!fetched) { avail(u, true); fetch(u); }
            const bool own = sp.upd[2 * u + 1] >= 0;
#pragma unroll
            for (int q = 0; q < CB * CB / 512; ++q) {
                const int e = 2 * (tid + 256 * q), r = e / CB, c = e % CB;
                *reinterpret_cast<double2 *>(&Xi[r * ULD + c]) = rx[q];
                if (own) *reinterpret_cast<double2 *>(&Xj[r * ULD + c]) = ry[q];
            }
            __syncthreads();
            fetched = u + 1 < w.upd1 && avail(u + 1, false);
            if (fetched) fetch(u + 1);
            if (own) strip_pqt64<true>(acc, Xi, Xj, wave, lane);
            if (chain) diag_update(Xi);
            __syncthreads();               // every wave is past its reads of Xi / Xj
        }
    }
    SP_T(1);                                           // updates done
    if (diag) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) store_d16(Xi + (16 * wave) * ULD + 16 * cb, ULD, acc[cb], lane);
        __syncthreads();
        factor_and_publish(Xi, Xj, w.J);
        SP_T(7);
        return;
    }
    SP_T(2);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) store_d16(Xi + (16 * wave) * ULD + 16 * cb, ULD, acc[cb], lane);
    // X_IJ = C L_JJ^-T once the inverse is there: all 32 KB in flight at once, the first 16 KB (rows 0..31) as soon as the factoring
    // workgroup has let go of them, a pivot chain before the tile is finished
    {
        const double2 *Lk = reinterpret_cast<const double2 *>(sp.Ldiag + (size_t)w.J * LSLOT + LINV_OFF);
        double2 lv[CB * CB / 512];
        // (a tile whose updates ended long after the inverse was published -- every separator row -- finds `ready` set and skips the
        // first poll: one memory round trip less on its way)
        const bool there = wait2(&sp.ready[w.J], nullptr, 1, false);
        if (!there) wait2(&sp.rpart[w.J], nullptr, 3, true);
#pragma unroll
        for (int q = 0; q < 4; ++q) lv[q] = Lk[tid + 256 * q];
        if (!there) wait2(&sp.ready[w.J], nullptr, 1, true);
        SP_T(3);                                       // inverse of the column's diagonal tile seen
#pragma unroll
        for (int q = 4; q < CB * CB / 512; ++q) lv[q] = Lk[tid + 256 * q];
#pragma unroll
        for (int q = 0; q < CB * CB / 512; ++q) {
            const int e = 2 * (tid + 256 * q);
            *reinterpret_cast<double2 *>(&Xj[(e / CB) * ULD + (e % CB)]) = lv[q];
        }
    }
    __syncthreads();
    doublex4 x[4] = {doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}, doublex4{0, 0, 0, 0}};
    strip_pqt64<false, true>(x, Xi, Xj, wave, lane);
    double *Xout = sp.W2 + (size_t)w.slot * (CB * CB);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * wave + (lane >> 4) + 4 * g, c = 16 * cb + (lane & 15);
            st_coh(&Xout[r * CB + c], x[cb][g]);
        }
    if (!chain) { publish_flag(&sp.xready[w.slot]); SP_T(5); return; }
    // chain: X also goes to LDS as the operand of diagonal tile (I, I)'s last update; its flag is raised once the factorisation
    // below has its operands (nobody on the critical chain waits for it)
    __syncthreads();                                    // every wave is past its reads of Xi (C) and Xj (L^-1)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) store_d16(Xi + (16 * wave) * ULD + 16 * cb, ULD, x[cb], lane);
    __syncthreads();
    diag_update(Xi);
    // (above the diagonal of the diagonal blocks: zeros)
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if ((lane & 15) > (lane >> 4) + 4 * g) {
            dacc[0][g] = wave == 0 || wave == 3 ? 0.0 : dacc[0][g];          // blocks (0,0) and (3,3)
            dacc[1][g] = wave == 1 ? 0.0 : dacc[1][g];                        // (1,1)
            dacc[2][g] = wave == 2 ? 0.0 : dacc[2][g];                        // (2,2)
        }
    __syncthreads();                                    // every wave is past its reads of Xi
    // Xj held L_JJ^-1, whose upper blocks are zeros and whose lower ones the waves rewrite now: the tile above the diagonal stays zero
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q < dnb) store_d16(Xj + (16 * dbi[q]) * ULD + 16 * dbj[q], ULD, dacc[q], lane);
    __syncthreads();
    SP_T(4);                                           // X in LDS, diagonal tile complete
    publish_flag(&sp.xready[w.slot]);                   // (the X stores above have long been acknowledged)
    SP_T(5);
    factor_and_publish(Xj, Xi, w.I);
    SP_T(7);
}

// Backward substitution L' y = z down the elimination tree.  Workgroup for column b (dispatched in descending b): z_b = row 0 of
// the right-hand side's factor tile of the column, minus L_ib' y_i for the rows i of its column structure, each y_i awaited in
// memory (a solution block is its own flag, see chol2_back_kernel); y_b = L_bb^-T z_b goes to ybuf and, un-permuted, to d.y_c.
__global__ __launch_bounds__(256) void chol_sparse_back_kernel(BADev d, SparseDev sp)
{
    __shared__ double z[CB];
    __shared__ double y[CB];
    __shared__ double part[4][CB];
    __shared__ int gave_up;
    const int tid = threadIdx.x, t = tid & 63, g = tid >> 6;
    const int b = sp.nb - 1 - (int)blockIdx.x;
    if (tid < CB) z[tid] = sp.W2[(size_t)sp.rhs_slot[b] * (CB * CB) + tid];
    if (tid == 0) gave_up = 0;
    __syncthreads();
    for (int u = sp.back0[b]; u < sp.back0[b + 1]; ++u) {
        const int i = sp.back[2 * u];
        const double *Lib = sp.W2 + (size_t)sp.back[2 * u + 1] * (CB * CB);
        double l[CB / 4];
#pragma unroll
        for (int q = 0; q < CB / 4; ++q) l[q] = Lib[(g + 4 * q) * CB + t];
        if (tid < CB) {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(sp.ybuf) + i * CB + tid;
            unsigned long long v;
            long spins = 0;
            while ((v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kYPending) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1L << 24)) { gave_up = 1; v = 0ull; break; }
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
    const double *Lb = sp.Ldiag + (size_t)b * LSLOT + LINV_OFF;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < CB / 4; ++q) { const int r = g + 4 * q; s = fma(Lb[r * CB + t], z[r], s); }
    part[g][t] = s;
    __syncthreads();
    if (tid < CB) {
        double yb = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
        const bool fail = d.scal[SC_CHOL_FAIL] != 0.0 || gave_up;
        if (__double_as_longlong(yb) == (long long)kYPending) yb = __longlong_as_double(0x7ff8000000000000ll);
        st_coh(&sp.ybuf[b * CB + tid], yb);
        const int src = sp.col_src[b * CB + tid];
        if (src >= 0) d.y_c[src] = fail ? 0.0 : yb;
        if (gave_up && tid == 0) d.scal[SC_CHOL_FAIL] = 1.0;
    }
}

// ---------------------------------------------------------------------------------------------
// host side

struct SparseSolve {
    SparsePlan plan;
    SparseDev dev{};
    std::vector<void *> allocs;
    int n_pack_blocks = 0;
    bool packed_source = false;          // the next solve reads the all-reduced packed blocks instead of d.red
    const double *packed = nullptr;
};

static int sp_alloc(SparseSolve *S, void **out, size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 16));
    if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return e == hipErrorOutOfMemory ? ESFM_ERR_OOM : ESFM_ERR_HIP; }
    S->allocs.push_back(p);
    *out = p;
    return ESFM_OK;
}

void ba_sparse_destroy(SparseSolve *S)
{
    if (!S) return;
    for (void *p : S->allocs) (void)hipFree(p);
    delete S;
}

int ba_sparse_create(hipStream_t st, const SparsePlan &plan, const CamGraph &g, SparseSolve **out)
{
    *out = nullptr;
    SparseSolve *S = new SparseSolve();
    S->plan = plan;
    const SparsePlan &P = S->plan;
    SparseDev &dv = S->dev;
    dv.nb = P.nb; dv.n_tiles = (int)P.tiles.size(); dv.n_wgs = (int)P.wgs.size();
    std::vector<int32_t> tile_ij(2 * P.tiles.size()), rhs_slot((size_t)P.nb);
    for (size_t k = 0; k < P.tiles.size(); ++k) {
        tile_ij[2 * k] = P.tiles[k].I; tile_ij[2 * k + 1] = P.tiles[k].J;
        if (P.tiles[k].I == P.nb) rhs_slot[(size_t)P.tiles[k].J] = (int32_t)k;
    }
    std::vector<SparseWg> wgs(P.wgs.size());
    for (size_t k = 0; k < P.wgs.size(); ++k) {
        const SparsePlan::Wg &a = P.wgs[k];
        wgs[k] = SparseWg{a.I, a.J, a.slot, a.upd0 / 2, a.upd1 / 2, a.dslot, a.kind};
    }
    // the exchange's block list: for camera a its neighbours b <= a (a itself last), ascending
    std::vector<int32_t> cov_start((size_t)g.n + 1, 0), cov_adj, cov_row;
    for (int a = 0; a < g.n; ++a) {
        cov_start[(size_t)a] = (int32_t)cov_adj.size();
        for (int32_t k = g.start[(size_t)a]; k < g.start[(size_t)a + 1] && g.adj[(size_t)k] < a; ++k) { cov_adj.push_back(g.adj[(size_t)k]); cov_row.push_back(a); }
        cov_adj.push_back(a); cov_row.push_back(a);
    }
    cov_start[(size_t)g.n] = (int32_t)cov_adj.size();
    dv.n_blocks = (int)cov_adj.size();
    int rc = ESFM_OK;
    auto up = [&](auto **dst, const auto &vec) {
        using T = typename std::remove_reference<decltype(vec)>::type::value_type;
        if (rc != ESFM_OK) return;
        void *p = nullptr;
        rc = sp_alloc(S, &p, sizeof(T) * vec.size());
        if (rc != ESFM_OK) return;
        *dst = reinterpret_cast<std::remove_pointer_t<std::remove_reference_t<decltype(*dst)>> *>(p);
        if (!vec.empty() && hipMemcpyAsync(p, vec.data(), sizeof(T) * vec.size(), hipMemcpyHostToDevice, st) != hipSuccess) { set_error("plan upload failed"); rc = ESFM_ERR_HIP; }
    };
    up(&dv.col_src, P.col_src); up(&dv.tile_ij, tile_ij); up(&dv.wgs, wgs); up(&dv.upd, P.upd);
    up(&dv.back0, P.back0); up(&dv.back, P.back); up(&dv.rhs_slot, rhs_slot);
    up(&dv.cov_start, cov_start); up(&dv.cov_adj, cov_adj); up(&dv.cov_row, cov_row);
    auto A = [&](auto **dst, size_t count) {
        if (rc != ESFM_OK) return;
        void *p = nullptr;
        rc = sp_alloc(S, &p, sizeof(**dst) * count);
        if (rc == ESFM_OK) *dst = reinterpret_cast<std::remove_pointer_t<std::remove_reference_t<decltype(*dst)>> *>(p);
    };
    const size_t tile = (size_t)CB * CB;
    A(&dv.W, tile * P.tiles.size()); A(&dv.W2, tile * P.tiles.size()); A(&dv.Ldiag, tile * (size_t)P.nb); A(&dv.ybuf, (size_t)CB * P.nb);
    A(&dv.xready, P.tiles.size()); A(&dv.ready, (size_t)P.nb); A(&dv.rpart, (size_t)P.nb);
    if (rc == ESFM_OK && hipStreamSynchronize(st) != hipSuccess) { set_error("plan upload failed"); rc = ESFM_ERR_HIP; }   // (the host vectors go out of scope)
    if (rc != ESFM_OK) { ba_sparse_destroy(S); return rc; }
    *out = S;
    return ESFM_OK;
}

const SparsePlan &ba_sparse_plan_of(const SparseSolve *S) { return S->plan; }
size_t ba_sparse_packed_doubles(const SparseSolve *S, int n_cam) { return (size_t)36 * (size_t)S->dev.n_blocks + 6 * (size_t)n_cam; }

int ba_sparse_pack(hipStream_t st, const BADev &d, SparseSolve *S, double *packed)
{
    const int n = 6 * d.n_cam, nblk_wg = (S->dev.n_blocks + 6) / 7;
    hipLaunchKernelGGL(ba_sparse_pack_kernel, dim3(nblk_wg + (n + 255) / 256), dim3(256), 0, st, d, S->dev, d.parts->red_rhs_exp, packed);
    ESFM_HIP_TRY(hipGetLastError());
    d.parts->red_fixed = false; d.parts->red_clean = true;
    S->packed_source = true; S->packed = packed;
    return ESFM_OK;
}

int ba_solve_reduced_sparse(hipStream_t st, const BADev &d, SparseSolve *S, double radius, double min_diag, double max_diag)
{
    const SparseDev &dv = S->dev;
    if (S->packed_source) {
        hipLaunchKernelGGL(chol_sparse_assemble_kernel<true>, dim3(4 * dv.n_tiles), dim3(256), 0, st, d, dv, radius, min_diag, max_diag, 0, S->packed);
        S->packed_source = false;
    } else {
        // (a buffer of zeros is fixed point on any grid: a problem without observations -- ba_schur returns before it writes anything --
        // still has a plan when its cameras number more than a leaf, and must solve like the dense path does)
        if (!d.parts->red_fixed && !d.parts->red_clean) { set_error("structure-aware solve: the Schur buffer is not in fixed point"); return ESFM_ERR_INVALID_ARG; }
        hipLaunchKernelGGL(chol_sparse_assemble_kernel<false>, dim3(4 * dv.n_tiles), dim3(256), 0, st, d, dv, radius, min_diag, max_diag, d.parts->red_rhs_exp,
                           (const double *)nullptr);
        d.parts->red_fixed = false; d.parts->red_clean = true;
    }
    ESFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(chol_sparse_kernel, dim3(dv.n_wgs), dim3(256), 0, st, dv, d.scal);
    ESFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(chol_sparse_back_kernel, dim3(dv.nb), dim3(256), 0, st, d, dv);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm
