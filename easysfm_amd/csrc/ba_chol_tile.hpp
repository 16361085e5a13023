// 64 x 64 f64 tile primitives of the reduced camera solve, shared by the dense dataflow factorisation (ba_chol_large.hip) and the
// structure-aware one (ba_chol_sparse.hip): 16 x 16 sub-block products on v_mfma_f64_16x16x4_f64, the one-wave factor-and-invert of a
// 16 x 16 block (generated asm, potrf16_gfx950.inc), the 64 x 64 tile factorisation with its inverse, and the write-through stores /
// flags that hand data from one workgroup to another inside a launch.
#pragma once
#include "ba_kernels.hpp"

namespace esfm {

constexpr int CB = 64;          // tile edge

// "not there yet" in the solution buffer of the backward substitution: all ones, a NaN no arithmetic produces (chol2_back_kernel)
constexpr unsigned long long kYPending = ~0ull;

// ---------------------------------------------------------------------------------------------
// 64 x 64 tile kernels built from 16 x 16 sub-blocks on the f64 matrix cores (v_mfma_f64_16x16x4_f64: lane l supplies
// A[l & 15][l >> 4] and B[l >> 4][l & 15]; its four results are D[(l >> 4) + 4 g][l & 15]).
//   pqt16 / pq16:      acc += sign * P Q' / P Q  for 16 x 16 row-major blocks in LDS  -- 4 MFMAs
//   tile_potrf64_inv:  4 sub-block steps: potrf16_fused_to, panel X = A Linv' (MFMA), trailing update (MFMA); the inverse of the
//                      factored tile from its sub-block inverses alongside
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
    asm volatile(ESFM_POTRF16_ASM : "=&v"(bad) : "v"(row_addr), "v"(col_addr) : ESFM_POTRF16_CLOBBERS);
    if (__any(bad) && lane == 0) *fail = 1;
}

// The same for a 16 x 16 diagonal sub-block of the 64 x 64 tile factorisation: D has row pitch ldd doubles and is NOT written (nothing
// reads a factored diagonal sub-block again -- the panel and the tile's inverse go through Vi); the inverse goes to Vi (row-major VLD).
__device__ __forceinline__ void potrf16_fused_to(const double *D, int ldd, double *Vi, int *fail, int lane)
{
    const int r = lane & 15;
    const uint32_t row_addr = (uint32_t)(uintptr_t)(D + r * ldd), col_addr = (uint32_t)(uintptr_t)(Vi + r);
    int bad;
    asm volatile(ESFM_POTRF16_ASM : "=&v"(bad) : "v"(row_addr), "v"(col_addr) : ESFM_POTRF16_CLOBBERS);
    if (__any(bad) && lane == 0) *fail = 1;
}

// ---------------------------------------------------------------------------------------------
// Second generation of the large solve (this round): ONE launch per block column and ONE for the whole backward substitution.
//   * the workgroup that factors a diagonal tile also inverts the factor (tile_potrf64_inv: the four 16 x 16 sub-block inverses are there
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

// Data handed from one workgroup to another INSIDE a launch (factor tiles, tile inverses, solution blocks) is written and read
// with agent-scope relaxed atomics -- plain stores / loads with the coherence bits set, write-through and L2-bypassing -- and
// ordered against the flag by a WORKGROUP-scope fence (s_waitcnt) plus a barrier.  An agent-scope release (__threadfence) is a
// write-back of the XCD's whole L2, with megabytes of other workgroups' dirty tiles in it: two of those per block column were
// ~15 us of the 31 us step.
__device__ __forceinline__ void st_coh(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_coh(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void publish_flag(int *f, int value = 1)
{
    // The workgroup-scope release fence alone emits NO s_waitcnt vmcnt(0) on gfx950 (a workgroup shares its L1 outside tgsplit
    // mode, so the compiler has nothing to wait for): the flag store could overtake the sc1 data stores on another channel.  The
    // explicit wait makes every wave's write-through stores L2-acknowledged before the barrier lets the flag out.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's stores have been acknowledged by L2
    __syncthreads();                                            // ... and everybody else's
    if (threadIdx.x == 0) __hip_atomic_store(f, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#if defined(ESFM_CHOL_TRACE) && !defined(ESFM_CHOL_NO_TRACE)
// timing-only build (scratch/build_variant_chol.sh NAME -DESFM_CHOL_TRACE): the chain workgroup (j+1, j) leaves s_memrealtime stamps
// (10 ns ticks) at its stages; scratch/chol_trace.py reads them through esfm_debug_chol_trace
__device__ unsigned long long g_chol_trace[64 * 12];
#define CHOL_T(col, q) do { if (threadIdx.x == 0) g_chol_trace[(col) * 12 + (q)] = wall_clock64(); } while (0)
// CHOL_ACC(var): var += ticks since the previous CHOL_ACC (wave 0 only)
#define CHOL_ACC(var) do { const long long tm1 = wall_clock64(), cy1 = clock64(); var += tm1 - tm0; var##_cyc += cy1 - cy0; tm0 = tm1; cy0 = cy1; } while (0)
extern "C" int esfm_debug_chol_trace(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_trace), sizeof(g_chol_trace)); }
#else
#define CHOL_T(col, q) do { } while (0)
#define CHOL_ACC(var) do { } while (0)
#endif

// 256 threads: in-place Cholesky of the 64 x 64 tile T (LDS, row-major ULD; upper part must be zero) AND the inverse of the factor
// in O (LDS, ULD).  Vi: 4 blocks of 16 x VLD (inverses of the diagonal sub-blocks).
// Factorisation, sub-block column b: panel X_i = A_i Linv_bb' (waves b+1 .. 3), then the trailing update -- during which wave 0
// updates sub-block (b+1, b+1) first and factors it at once (look-ahead: the next pivot chain runs while the other waves finish the
// update); two barriers per sub-block column.
// Inverse:  O_jj = Vi_j,  O_ij = -Vi_i S_ij,  S_ij = sum_{m=j}^{i-1} L_im O_mj  for i > j.  Block column jc is wave jc + 1's, and its
// rows are formed IN THE SHADOW of wave 0's pivot chains instead of after them: while wave 0 factors sub-block b + 1 (potrf16: ~1.6
// us, nothing for the others to do once their few trailing blocks are updated), wave jc + 1 finishes row b of its column and sums
// S_b+1,jc -- every operand is final by then: Vi_b since the barrier before the window, L_b+1,m since this step's panel.  After the
// last pivot chain one product per wave is left (O_3,jc = -Vi_3 S_3,jc).  S_ij is parked in O_ij's own place (no scratch: 9 KB of
// LDS less is what lets two workgroups share a CU).  (Until round 3 the inverse was a pass of its own after the factorisation:
// 2.8 us of the block column's critical chain, now 0.4.)
// The inverse leaves for memory (Ld: 64 x 64 row-major, read by the next block column's workgroups and by the backward substitution)
// from here, every wave storing what it computed itself: rows 0..31 while wave 0 is still in the LAST pivot chain -- part_flag counts
// the three waves that have done so, and a consumer that sees 3 starts fetching those 16 KB a microsecond before the tile is finished
// -- and rows 32..63 at the end; the caller raises the tile's ready flag behind them (publish_flag).
__device__ __forceinline__ void tile_potrf64_inv(double *T, double *O, double *Vi, int *fail, double *__restrict__ Ld, int *part_flag, int trace_col = 0)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jc = wave - 1;
#if defined(ESFM_CHOL_TRACE) && !defined(ESFM_CHOL_NO_TRACE)
    long long tm0 = wall_clock64(), cy0 = clock64(), t_potrf = 0, t_rest = 0, t_potrf_cyc = 0, t_rest_cyc = 0;
#endif
    auto copy_vi = [&](int b) {                          // O_bb = Vi_b
        for (int e = lane; e < SB * SB; e += 64) O[(SB * b + e / SB) * ULD + SB * b + e % SB] = Vi[b * SB * VLD + (e / SB) * VLD + e % SB];
    };
    auto finish_row = [&](int b) {                       // O_b,jc = -Vi_b S_b,jc  (jc < b; S is parked in O_b,jc's place)
        double *Obj = O + (SB * b) * ULD + SB * jc;
        const doublex4 acc = pq16(doublex4{0.0, 0.0, 0.0, 0.0}, Vi + b * SB * VLD, VLD, Obj, ULD, -1.0, lane);
        __builtin_amdgcn_wave_barrier();
        store_d16(Obj, ULD, acc, lane);
        __builtin_amdgcn_wave_barrier();
    };
    if (wave == 0) {
        potrf16_fused_to(T, ULD, Vi, fail, lane);
        CHOL_ACC(t_potrf);
    } else {
        // zeros above the diagonal blocks: block column jc (rows < 16 jc), and wave 1 also takes block column 3
        for (int e = lane; e < SB * jc * SB; e += 64) O[(e / SB) * ULD + SB * jc + e % SB] = 0.0;
        if (wave == 1) for (int e = lane; e < SB * 3 * SB; e += 64) O[(e / SB) * ULD + SB * 3 + e % SB] = 0.0;
    }
    __syncthreads();
#pragma unroll 1
    for (int b = 0; b < 3; ++b) {
        if (wave > b) {                                  // panel: strip i = wave (A_i is read whole before it is overwritten)
            const int i = wave;
            doublex4 acc = pqt16(doublex4{0.0, 0.0, 0.0, 0.0}, T + (SB * i) * ULD + SB * b, ULD, Vi + b * SB * VLD, VLD, 1.0, lane);
            __builtin_amdgcn_wave_barrier();
            store_d16(T + (SB * i) * ULD + SB * b, ULD, acc, lane);
        }
        __syncthreads();
        if (wave == 0) {
            const int i = b + 1;
            doublex4 acc = load_d16(T + (SB * i) * ULD + SB * i, ULD, lane);
            acc = pqt16(acc, T + (SB * i) * ULD + SB * b, ULD, T + (SB * i) * ULD + SB * b, ULD, -1.0, lane);
            store_d16(T + (SB * i) * ULD + SB * i, ULD, acc, lane);
            __builtin_amdgcn_wave_barrier();
            CHOL_ACC(t_rest);
            potrf16_fused_to(T + (SB * i) * ULD + SB * i, ULD, Vi + i * SB * VLD, fail, lane);
            CHOL_ACC(t_potrf);
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
            // the inverse: row b of block column jc, then the sum for row b + 1
            if (jc <= b) {
                if (b == 2) {                            // rows 0..31 of this wave's block column(s) are final since the last window: on their way (8 per lane)
                    for (int e = lane; e < 2 * SB * SB; e += 64) st_coh(&Ld[(e / SB) * CB + SB * jc + e % SB], O[(e / SB) * ULD + SB * jc + e % SB]);
                    if (wave == 1) for (int e = lane; e < 2 * SB * SB; e += 64) st_coh(&Ld[(e / SB) * CB + SB * 3 + e % SB], 0.0);
                }
                if (jc == b) copy_vi(b); else finish_row(b);
                __builtin_amdgcn_wave_barrier();
                doublex4 acc = doublex4{0.0, 0.0, 0.0, 0.0};
                for (int m = jc; m <= b; ++m) acc = pq16(acc, T + (SB * (b + 1)) * ULD + SB * m, ULD, O + (SB * m) * ULD + SB * jc, ULD, 1.0, lane);
                store_d16(O + (SB * (b + 1)) * ULD + SB * jc, ULD, acc, lane);
                if (b == 2) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores are acknowledged (issued a microsecond ago)
                    if (lane == 0) __hip_atomic_fetch_add(part_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0) copy_vi(3); else finish_row(3);
    {
        const int cb = wave == 0 ? 3 : jc;              // rows 32..63 of this wave's block column (wave 0: the last one, zeros over Vi_3)
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < 2 * SB * SB; e += 64) st_coh(&Ld[(2 * SB + e / SB) * CB + SB * cb + e % SB], O[(2 * SB + e / SB) * ULD + SB * cb + e % SB]);
    }
#if defined(ESFM_CHOL_TRACE) && !defined(ESFM_CHOL_NO_TRACE)
    if (tid == 0) { CHOL_ACC(t_rest); g_chol_trace[trace_col * 12 + 10] = (unsigned long long)t_potrf; g_chol_trace[trace_col * 12 + 11] = (unsigned long long)t_rest;
                    g_chol_trace[trace_col * 12 + 8] = (unsigned long long)t_potrf_cyc; (void)t_rest_cyc; }    // (slot 8, "inv64", is free since the inverse moved)
#endif
}

// this wave's 16-row strip of  P (64 x 64, LDS ULD) * Q' (Q 64 x 64, LDS ULD): four 16 x 16 outputs, K = 64
// QTRI: Q is lower triangular (the inverse of a factor): its 16 x 16 blocks (cb, kb) with kb > cb are zero and are skipped -- 40
// matrix instructions per wave instead of 64 (v_mfma_f64_16x16x4_f64 issues every 64 cycles: the product is 2 us of a CU otherwise)
template <bool NEGATE = false, bool QTRI = false>
__device__ __forceinline__ void strip_pqt64(doublex4 (&acc)[4], const double *P, const double *Q, int wave, int lane)
{
    const double *ap = P + (16 * wave + (lane & 15)) * ULD + (lane >> 4);
    const double *bp = Q + (lane & 15) * ULD + (lane >> 4);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int kk = 4 * kb + k4;
            const double a = NEGATE ? -ap[4 * kk] : ap[4 * kk];
#pragma unroll
            for (int cb = QTRI ? kb : 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[16 * cb * ULD + 4 * kk], acc[cb], 0, 0, 0);
        }
    }
}

}  // namespace esfm
