// Pairwise descriptor matching on gfx950 (MI355X): the replacement for the
// knnMatch(...,2) + Lowe ratio loop of FeatureMatching::matchFeaturesSURF / ORB
// (reference cpp_code/src/feature_matching.cpp:115-142 and :71-97), batched over
// the pair loop of cpp_code/test/sfm.cpp:140-161.
//
// L2 (SURF, float):
//   l2_split_bf16_kernel     64-float rows as bf16 hi + lo halves (train image, query image = -2 x) and |row|^2
//   l2_knn_bf16_kernel       64-float rows: distance GEMM as three bf16 MFMAs per product with a fused per-lane top-3,
//                            the exact re-rank and the rounding-error certificate of l2_knn_mfma_kernel (eps 2^-15)
//   l2_row_norms_kernel      |t|^2 per descriptor row (other widths)
//   l2_knn_mfma_kernel       128-float rows: distance GEMM on v_mfma_f32_32x32x2_f32 with a fused
//                            per-lane top-3 epilogue, then an exact re-rank of the
//                            6 candidates per query in the oracle's summation
//                            order and a rounding-error certificate
//   l2_exact_scan_kernel     exact brute-force scan for the (rare) queries the
//                            certificate rejects, and for widths without an MFMA build
// Hamming (ORB, 256-bit):
//   hamming_knn_mfma_kernel  256-bit descriptors: 0/1 byte expansion on the i8 matrix cores, fused top-2
//   hamming_knn_kernel       other widths: XOR + popcount, (distance,index) packed into one u32 key
// Both:
//   ratio_compact_kernel     ratio test in double + ordered compaction per pair
//
// DESIGN.md "Matching" explains the data layout and the certificate.
#include "match_kernels.hpp"
#ifndef ESFM_L2_SEGMENT_INC            // (scratch/build_variant.sh swaps in experimental schedules)
#define ESFM_L2_SEGMENT_INC "l2_segment_gfx950.inc"
#endif
#include ESFM_L2_SEGMENT_INC           // ESFM_L2_SEGMENT_ASM: the matcher's hand-scheduled main loop (gen_l2_segment_asm.py)
#ifndef ESFM_L2X1_SEGMENT_INC
#define ESFM_L2X1_SEGMENT_INC "l2x1_segment_gfx950.inc"
#endif
#include ESFM_L2X1_SEGMENT_INC         // ESFM_L2X1_SEGMENT_ASM, ESFM_L2X1_KEEP: the one-product pass's main loop (gen_l2x1_segment_asm.py)
#include "hmx1_segment_gfx950.inc"     // ESFM_HMX1_SEGMENT_ASM: the same loop around v_mfma_f32_32x32x64_f8f6f4 on FP4 operands (256-bit Hamming)

#include <float.h>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

namespace esfm {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// LDS-DMA staging (buffer_load_dwordx4 ... lds): lane l of the wave writes its 16 bytes to lds_dst + 16 l, from byte
// voff (per lane) + soff (wave-uniform) of the buffer.  Issued from inline asm on purpose: through the builtin hipcc orders
// every later LDS read behind vmcnt(0) (it cannot tell a double buffer's halves apart) and the transfer would serialise with the
// compute it is meant to hide under.  The asm is invisible to the waitcnt pass, so the CALLER waits: lds_dma_wait() in front of
// the barrier that publishes the tile.  M0 (the LDS base of the transfer) is saved and restored around the instruction.
__device__ __forceinline__ u32x4 raw_buffer_rsrc(const void *base, uint32_t bytes)
{
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xFFFFu);   // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane(bytes);                            // the per-lane offset is range-checked against it
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void lds_dma_b128(uint32_t lds_dst /* wave-uniform */, int voff, u32x4 rsrc, int soff /* wave-uniform */)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// helpers

__device__ __forceinline__ int xcd_remap(int bid, int nb)
{
    // Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8).  Give each XCD a
    // contiguous range of logical blocks so that the blocks sharing one pair's train set hit the
    // same 4 MiB L2.  Bijective for any nb.
    const int q = nb >> 3, r = nb & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ int find_pair_by_block(const PairDesc *pairs, int n_pairs, int lb)
{
    int lo = 0, hi = n_pairs - 1;  // last p with blk_off[p] <= lb
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (pairs[mid].blk_off <= lb) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ int find_pair_by_query(const PairDesc *pairs, int n_pairs, long long gq)
{
    int lo = 0, hi = n_pairs - 1;  // last p with out_off[p] <= gq
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (pairs[mid].out_off <= gq) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Squared L2 distance in the oracle's canonical order (oracle/match_ref.c esfm_ref_l2sqr):
// 8 partial sums over blocks of 8, separate multiply and add (no FMA), (acc[c]+acc[c+4]) summed
// left to right, then the scalar tail.  Bit-exact with the CPU restatement.
template <bool VEC>
__device__ __forceinline__ float l2sqr_canonical(const float *__restrict__ a, const float *__restrict__ b, int n)
{
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j <= n - 8; j += 8) {
        float av[8], bv[8];
        if (VEC) {
            const float4 a0 = *reinterpret_cast<const float4 *>(a + j), a1 = *reinterpret_cast<const float4 *>(a + j + 4);
            const float4 b0 = *reinterpret_cast<const float4 *>(b + j), b1 = *reinterpret_cast<const float4 *>(b + j + 4);
            av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) { av[c] = a[j + c]; bv[c] = b[j + c]; }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float t = __fsub_rn(av[c], bv[c]);
            acc[c] = __fadd_rn(acc[c], __fmul_rn(t, t));
        }
    }
    const float s0 = __fadd_rn(acc[0], acc[4]);
    const float s1 = __fadd_rn(acc[1], acc[5]);
    const float s2 = __fadd_rn(acc[2], acc[6]);
    const float s3 = __fadd_rn(acc[3], acc[7]);
    float d = __fadd_rn(s0, s1);
    d = __fadd_rn(d, s2);
    d = __fadd_rn(d, s3);
    for (; j < n; ++j) {
        const float t = __fsub_rn(a[j], b[j]);
        d = __fadd_rn(d, __fmul_rn(t, t));
    }
    return d;
}

// l2sqr_canonical for 64-float rows held in registers: the same 8 chains, the same final order.
// (Measured in the distance pass's tail: the packed form below made the whole kernel 1.5 % SLOWER -- 1.499 -> 1.522 ms; the
// tail's arithmetic runs beside the other workgroup's MFMAs and the chip is power-limited there -- so the tail keeps this one.)
__device__ __forceinline__ float l2sqr64_canonical_regs(const float4 (&a)[16], const float4 (&b)[16])
{
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float av[8] = {a[2 * j].x, a[2 * j].y, a[2 * j].z, a[2 * j].w, a[2 * j + 1].x, a[2 * j + 1].y, a[2 * j + 1].z, a[2 * j + 1].w};
        const float bv[8] = {b[2 * j].x, b[2 * j].y, b[2 * j].z, b[2 * j].w, b[2 * j + 1].x, b[2 * j + 1].y, b[2 * j + 1].z, b[2 * j + 1].w};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float t = __fsub_rn(av[c], bv[c]);
            acc[c] = __fadd_rn(acc[c], __fmul_rn(t, t));
        }
    }
    const float s0 = __fadd_rn(acc[0], acc[4]);
    const float s1 = __fadd_rn(acc[1], acc[5]);
    const float s2 = __fadd_rn(acc[2], acc[6]);
    const float s3 = __fadd_rn(acc[3], acc[7]);
    float d = __fadd_rn(s0, s1);
    d = __fadd_rn(d, s2);
    return __fadd_rn(d, s3);
}
// The same with a row spread over SIXTEEN LANES (lane l of a 16-lane DPP row holds floats 4 l .. 4 l + 3 of both operands): float
// 4 l + x belongs to chain c = 4 (l & 1) + x at step j = l >> 1, so a chain runs over the lanes of equal parity in lane order --
// seven `row_shr:2` additions  A_k[l] = A_(k-1)[l - 2] + d[l]  (A_0 = d; the chain's 0 + d_0 is d_0) leave chains 0 .. 3 in lane 14
// and 4 .. 7 in lane 15; lane 15 then forms s_x = acc[x] + acc[x + 4] and ((s0 + s1) + s2) + s3.  The result is valid in lane 15
// of every row (four rows per wave).  Same operations on the same operands in the same order as l2sqr64_canonical_regs.
__device__ __forceinline__ float dpp_row_shr_f(float v, int n_is_2)
{
    return n_is_2 ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true))
                  : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
}
__device__ __forceinline__ float l2sqr64_canonical_row16(const u32x4 a, const u32x4 b)
{
    float d[4], acc[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const float t = __fsub_rn(__uint_as_float(a[x]), __uint_as_float(b[x]));
        d[x] = __fmul_rn(t, t);
        acc[x] = d[x];
    }
#pragma unroll
    for (int k = 1; k < 8; ++k)
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[x] = __fadd_rn(dpp_row_shr_f(acc[x], 1), d[x]);
    float sx[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) sx[x] = __fadd_rn(dpp_row_shr_f(acc[x], 0), acc[x]);      // lane 15: acc[x] of lane 14 + acc[x + 4] of its own
    float r = __fadd_rn(sx[0], sx[1]);
    r = __fadd_rn(r, sx[2]);
    return __fadd_rn(r, sx[3]);
}
// The same on two rows that sit in LDS as 16 x 16 B with their slots XOR-swizzled (slot c of a row at piece c ^ sw): the pieces are
// read as they are used, so neither row has to be held in 64 registers.  a_row / b_row: LDS byte address of the row, a_sw16 / b_sw16:
// 16 sw.  The 32 piece addresses are formed HERE, every time, from operands the compiler cannot see through (one v_xad_u32 each): as
// ordinary loop invariants they are hoisted out of the caller's loops, live across everything, get spilled, and every LDS read then
// waits for the scratch reload of its own address (seen in the re-rank: 219 spilled registers).  Same 8 chains, same final order.
typedef float floatx4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const floatx4_t *lds_cf4p;
__device__ __forceinline__ float l2sqr64_canonical_lds(uint32_t a_row, uint32_t a_sw16, uint32_t b_row, uint32_t b_sw16)
{
    asm volatile("" : "+v"(a_sw16), "+v"(b_sw16));
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const floatx4_t a0 = *(lds_cf4p)(uintptr_t)((a_sw16 ^ (uint32_t)(32 * j)) + a_row), a1 = *(lds_cf4p)(uintptr_t)((a_sw16 ^ (uint32_t)(32 * j + 16)) + a_row);
        const floatx4_t b0 = *(lds_cf4p)(uintptr_t)((b_sw16 ^ (uint32_t)(32 * j)) + b_row), b1 = *(lds_cf4p)(uintptr_t)((b_sw16 ^ (uint32_t)(32 * j + 16)) + b_row);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float t = __fsub_rn(av[c], bv[c]);
            acc[c] = __fadd_rn(acc[c], __fmul_rn(t, t));
        }
    }
    const float s0 = __fadd_rn(acc[0], acc[4]);
    const float s1 = __fadd_rn(acc[1], acc[5]);
    const float s2 = __fadd_rn(acc[2], acc[6]);
    const float s3 = __fadd_rn(acc[3], acc[7]);
    float d = __fadd_rn(s0, s1);
    d = __fadd_rn(d, s2);
    return __fadd_rn(d, s3);
}
// The same with two neighbouring chains per packed instruction (v_pk_add_f32 / v_pk_mul_f32: every half is an IEEE single
// operation, the result is bit-identical): the re-scan kernels, which have the chip to themselves, run on these.
typedef float float2v __attribute__((ext_vector_type(2)));

// Correctly rounded f32 square root.  NOT __fsqrt_rn: in this toolchain that maps to
// __ocml_native_sqrt_f32 (about 1 ulp), while sqrtf is IEEE-exact under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt and matches the CPU's sqrtss bit for bit.
__device__ __forceinline__ float sqrt_rn_f32(float x) { return sqrtf(x); }

// (distance, index) ordered pair; "better" = the order a stable ascending scan with strict-<
// insertion produces (OpenCV batchDistance): smaller distance, ties to the lower train index.
struct Cand { float d; int i; float d2; };

__device__ __forceinline__ bool cand_better(float d, int i, const Cand &b)
{
    // an empty slot holds FLT_MAX: like the oracle's strict `d < d1`, a distance of FLT_MAX, +inf or NaN is never a neighbour
    return (i >= 0) && (d < b.d || (d == b.d && i < b.i));
}

__device__ __forceinline__ void best2_insert(Cand &b0, Cand &b1, float d, int i, float d2)
{
    if (cand_better(d, i, b1)) {
        if (cand_better(d, i, b0)) { b1 = b0; b0.d = d; b0.i = i; b0.d2 = d2; }
        else { b1.d = d; b1.i = i; b1.d2 = d2; }
    }
}

// ---------------------------------------------------------------------------------------------
// |row|^2 for every descriptor row (float chain; only used by the approximate pass + certificate)
__global__ void l2_row_norms_kernel(const float *__restrict__ desc, int dim, long long n_rows, float *__restrict__ norms)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const float *p = desc + r * dim;
    float s = 0.f;
    for (int k = 0; k < dim; ++k) s = fmaf(p[k], p[k], s);
    norms[r] = s;
}

// ---------------------------------------------------------------------------------------------
// MFMA distance pass.
//
// One workgroup (4 waves) owns QB = 128 query rows of one pair and streams the whole train set
// through LDS in tiles of TT = 64 rows.  Each wave owns 32 queries for the entire kernel: their
// descriptors, scaled by -2, stay in HALF = DIM/2 VGPRs per lane as the MFMA B operand
// (lane l: query l&31, features [HALF*(l>>5), HALF*(l>>5)+HALF)).  A train sub-tile of 32 rows is
// the A operand, read from LDS with ds_read_b128 (XOR-swizzled 16-B slots: conflict-free).  The
// accumulator starts at |t|^2, so after DIM/2 MFMAs D[t][q] = |t|^2 - 2 q.t  (= d^2 - |q|^2) with
// no epilogue arithmetic.  C/D layout: lane l, reg r -> train row (r&3)+8*(r>>2)+4*(l>>5), query
// l&31, i.e. the 16 values in a lane belong to ONE query, so the running top-3 is lane-local.
template <int DIM, int TT>
__global__ __launch_bounds__(256) void l2_knn_mfma_kernel(const float *__restrict__ desc, const float *__restrict__ norms,
                                                          const PairDesc *__restrict__ pairs, int n_pairs,
                                                          int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist,
                                                          int32_t *__restrict__ flagged, int32_t *__restrict__ counters,
                                                          int flag_cap)
{
    constexpr int QB = 128, HALF = DIM / 2, NCH = HALF / 4, SLOTS = DIM / 4;
    constexpr int STAGE = TT * SLOTS / 256;  // float4 per thread per tile
    static_assert(DIM % 8 == 0 && STAGE >= 1, "DIM");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *lds_tile = reinterpret_cast<float4 *>(smem);                       // [2][TT*SLOTS]
    float *lds_norm = reinterpret_cast<float *>(smem + 2 * TT * SLOTS * 16);   // [2][TT]
    float *lds_red = lds_norm + 2 * TT;                                        // [4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int pi = find_pair_by_block(pairs, n_pairs, lb);
    const PairDesc pd = pairs[pi];
    const int nq = pd.nq, nt = pd.nt;
    const float *__restrict__ Q = desc + (size_t)pd.q_row0 * DIM;
    const float *__restrict__ T = desc + (size_t)pd.t_row0 * DIM;
    const float *__restrict__ tn = norms + pd.t_row0;
    const int qrow = (lb - pd.blk_off) * QB + wave * 32 + j;
    const bool qvalid = qrow < nq;

    // B operand: this lane's half of its query row, times -2 (exact scaling).
    float breg[HALF];
    {
        const float4 *qp = reinterpret_cast<const float4 *>(Q + (size_t)(qvalid ? qrow : 0) * DIM + h * HALF);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float4 v = qvalid ? qp[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            breg[4 * c + 0] = -2.f * v.x; breg[4 * c + 1] = -2.f * v.y; breg[4 * c + 2] = -2.f * v.z; breg[4 * c + 3] = -2.f * v.w;
        }
    }

    // Running top-3, two levels.
    //
    // On gfx950 the f32-input MFMA runs at the f32 VECTOR rate and VALU work does NOT hide under it
    // (measured: every VALU instruction next to v_mfma_f32_32x32x2_f32 adds ~3 cycles per SIMD), so
    // the fold is budgeted in instructions per element.  Level 1 (per element, 4 VALU ops, no
    // compares): the low 8 mantissa bits of s are replaced by an 8-bit position code
    // (key = (s & ~0xFF) | code, one v_and_or_b32) and the three smallest keys of the current
    // 512-row segment are kept with v_med3_f32 / v_med3_f32 / v_min -- as floats, the keys order like
    // s truncated to 15 mantissa bits.  Level 2 (once per segment = 256 elements per lane): the three
    // segment keys are decoded to (key, train row) and merged into the lane's master top-3 with the
    // compare/select chain.  The truncation error (< 2^-14 |key|) is charged to the certificate.
    constexpr float kBig = 3.0e38f;       // finite "empty slot" sentinel; padded train rows carry |t|^2 = kBig too
    constexpr int kSegSub = 16;           // sub-tiles (32 rows) per segment -> 8-bit codes
    float k0 = kBig, k1 = kBig, k2 = kBig;             // segment keys
    float v0 = kBig, v1 = kBig, v2 = kBig;             // master keys
    int c0 = -1, c1 = -1, c2 = -1;                     // master train rows
    float tmax = 0.f;  // max |t|^2 seen by this thread (threads < TT only)
    int poison = 0;    // a train row with a non-finite norm (inf / NaN entries, or an overflowing |t|^2): its scores can be NaN, and a NaN key
                       // corrupts the v_med3 network -- no query of this workgroup is certified, the exact re-scan decides
    unsigned kmask = 0xFFFFFF00u;
    asm volatile("" : "+v"(kmask));   // keep the mask in a VGPR: v_and_or_b32 can then take the code as its one SGPR operand
    auto fold = [&](float s, int code /* wave-uniform */) {
        float key;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(s), "v"(kmask), "s"(code));
        k2 = __builtin_amdgcn_fmed3f(k1, k2, key);
        k1 = __builtin_amdgcn_fmed3f(k0, k1, key);
        k0 = __builtin_amdgcn_fmed3f(k0, key, -kBig);   // min without the NaN-quieting v_max pair
    };
    auto master_insert = [&](float key, int seg_sub0) {
        // decode: code = 16 * (sub-tile within segment) + accumulator register
        const int code = (int)(__float_as_uint(key) & 0xFFu);
        const int r = code & 15;
        const int t = (seg_sub0 + (code >> 4)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool live = key < 1.0e38f;
        const bool l2 = live && key < v2, l1 = live && key < v1, l0 = live && key < v0;
        const int t2 = l2 ? t : c2;
        const int t1 = l1 ? t : c1;
        c2 = l1 ? c1 : t2;
        c1 = l0 ? c0 : t1;
        c0 = l0 ? t : c0;
        const float n2 = l2 ? key : v2;
        const float n1 = l1 ? key : v1;
        v2 = l1 ? v1 : n2;
        v1 = l0 ? v0 : n1;
        v0 = l0 ? key : v0;
    };
    auto flush = [&](int seg_sub0) {
        master_insert(k0, seg_sub0); master_insert(k1, seg_sub0); master_insert(k2, seg_sub0);
        k0 = k1 = k2 = kBig;
    };

    const int ntiles = (nt + TT - 1) / TT;
    float4 stage[STAGE];
    float stage_n = kBig;
    // Staging loads go through a buffer descriptor over the train set: rows past nt read as zeros in
    // hardware, the per-thread byte offset is one loop-invariant VGPR and the tile offset is scalar, so
    // a tile costs no address VALU (VALU does not overlap the f32 MFMA, every instruction counts).
    const __amdgpu_buffer_rsrc_t trsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(T), 0, nt * DIM * 4, 0x00020000);
    const int voff = (tid / SLOTS) * (DIM * 4) + (tid % SLOTS) * 16;   // row-in-pass * row bytes + slot * 16
    auto gload = [&](int tile) {
#pragma unroll
        for (int i = 0; i < STAGE; ++i) {
            const int soff = (tile * TT + i * (256 / SLOTS)) * (DIM * 4);   // wave-uniform
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(trsrc, voff, soff, 0);
            stage[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
        const int t = tile * TT + (tid & (TT - 1));
        const float nv = tn[min(t, nt - 1)];
        stage_n = (t < nt) ? nv : kBig;
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < STAGE; ++i) {
            const int s = tid + 256 * i, row = s / SLOTS, slot = s % SLOTS;
            lds_tile[buf * TT * SLOTS + row * SLOTS + (slot ^ (row & 15))] = stage[i];
        }
        if (tid < TT) { lds_norm[buf * TT + tid] = stage_n; if (stage_n < 1.0e38f) tmax = fmaxf(tmax, stage_n); else if (!(stage_n == kBig)) poison = 1; }
    };

    if (ntiles > 0) { gload(0); lstore(0); }
    __syncthreads();

    static_assert(TT % 64 == 0 && TT <= 256, "a tile is a whole number of 64-row sub-tile pairs");
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        gload(min(tile + 1, ntiles - 1));  // next tile in flight under the MFMAs below (last trip: harmless re-load)
#pragma unroll
        for (int sp = 0; sp < TT / 64; ++sp) {
            const int base = sp * 64;
            floatx16 acc0, acc1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 n0 = *reinterpret_cast<const float4 *>(&lds_norm[buf * TT + base + 8 * g + 4 * h]);
                const float4 n1 = *reinterpret_cast<const float4 *>(&lds_norm[buf * TT + base + 32 + 8 * g + 4 * h]);
                acc0[4 * g + 0] = n0.x; acc0[4 * g + 1] = n0.y; acc0[4 * g + 2] = n0.z; acc0[4 * g + 3] = n0.w;
                acc1[4 * g + 0] = n1.x; acc1[4 * g + 1] = n1.y; acc1[4 * g + 2] = n1.z; acc1[4 * g + 3] = n1.w;
            }
            float4 a0[NCH], a1[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {   // (base + 32 + j) & 15 == j & 15
                a0[c] = lds_tile[buf * TT * SLOTS + (base + j) * SLOTS + ((h * NCH + c) ^ (j & 15))];
                a1[c] = lds_tile[buf * TT * SLOTS + (base + 32 + j) * SLOTS + ((h * NCH + c) ^ (j & 15))];
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c].x, breg[4 * c + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c].x, breg[4 * c + 0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c].y, breg[4 * c + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c].y, breg[4 * c + 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c].z, breg[4 * c + 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c].z, breg[4 * c + 2], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c].w, breg[4 * c + 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c].w, breg[4 * c + 3], acc1, 0, 0, 0);
            }
            // The fold reads the accumulators from inline asm, for which hipcc pads no hazards: an MFMA's
            // result needs ~18 wait states (16-pass op) before a non-MFMA reader.  Routing both
            // accumulators through this statement orders every fold after the last MFMA plus the pad.
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc0), "+v"(acc1));
            const int sub = tile * (TT / 32) + 2 * sp;  // global sub-tile index of acc0
            const int cb = __builtin_amdgcn_readfirstlane((sub % kSegSub) * 16);   // code base inside the segment (SGPR)
#pragma unroll
            for (int r = 0; r < 16; ++r) fold(acc0[r], cb + r);
#pragma unroll
            for (int r = 0; r < 16; ++r) fold(acc1[r], cb + 16 + r);
            if ((sub + 2) % kSegSub == 0) flush(sub + 2 - kSegSub);
        }
        if (tile + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }
    {
        const int nsub = ntiles * (TT / 32);
        if (nsub % kSegSub != 0) flush((nsub / kSegSub) * kSegSub);
    }

    // max |t|^2 over the train set (for the certificate's error bound)
    {
        float m = tmax;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) lds_red[wave] = m;
        poison = __syncthreads_or(poison);
        tmax = fmaxf(fmaxf(lds_red[0], lds_red[1]), fmaxf(lds_red[2], lds_red[3]));
    }

    // ---- exact re-rank of this lane's 3 candidates in the oracle's order ----
    Cand b0 = {FLT_MAX, -1, 0.f}, b1 = {FLT_MAX, -1, 0.f};
    float ed[3], ed2[3];
    int ei[3];
    {
        const int cc[3] = {c0, c1, c2};
        const float *qp = Q + (size_t)(qvalid ? qrow : 0) * DIM;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            ei[m] = -1; ed[m] = FLT_MAX; ed2[m] = 0.f;
            if (cc[m] >= 0 && qvalid) {
                const int t = cc[m];
                const float d2 = l2sqr_canonical<true>(qp, T + (size_t)t * DIM, DIM);
                ei[m] = t; ed2[m] = d2; ed[m] = sqrt_rn_f32(d2);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) best2_insert(b0, b1, ed[m], ei[m], ed2[m]);
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const float pd_ = __shfl_xor(ed[m], 32), pd2 = __shfl_xor(ed2[m], 32);
        const int pi_ = __shfl_xor(ei[m], 32);
        best2_insert(b0, b1, pd_, pi_, pd2);
    }
    const float tau = fminf(v2, __shfl_xor(v2, 32));  // every train outside the 6 candidates has s >= tau

    if (qvalid && h == 0) {
        const size_t o = 2 * ((size_t)pd.out_off + qrow);
        knn_idx[o] = b0.i; knn_idx[o + 1] = b1.i;
        knn_dist[o] = b0.d; knn_dist[o + 1] = b1.d;
        // Certificate: |(|q|^2 + s(t)) - D(t)| <= 2^-16 (|q|^2 + max|t|^2) for every train t, and
        // every train outside the candidates has key >= tau, hence s >= tau - 2^-14 |tau| (truncation);
        // the candidate set provably contains the two best iff |q|^2 + tau - eps exceeds the second
        // best exact d^2 by more than sqrt's rounding can hide.
        bool certified = (tau >= 1.0e38f) && !poison;   // an empty slot in either lane: every train row is a candidate (a NaN tau is NOT certified)
        if (!certified && b1.i >= 0 && !poison) {
            const double qn = (double)norms[pd.q_row0 + qrow];
            const double eps = (qn + (double)tmax) * (1.0 / 65536.0) + fabs((double)tau) * (1.0 / 16384.0);
            certified = (qn + (double)tau - eps) > (double)b1.d2 * (1.0 + 1.0 / 2097152.0);
        }
        if (!certified) {
            const int slot = atomicAdd(&counters[0], 1);
            if (slot < flag_cap) { flagged[2 * slot] = pi; flagged[2 * slot + 1] = qrow; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 distance pass (64-float descriptors): the same kernel shape as l2_knn_mfma_kernel, with the f32 MFMA (157 TFLOP/s,
// no VALU co-execution) replaced by three bf16 MFMAs (2.5 PFLOP/s, VALU runs beside them).  Every float a is split into
// hi = bf16(a) and lo = bf16(a - hi) (round to nearest even; a = hi + lo + e, |e| <= 2^-18 |a|), and
//   q.t ~ sum hi_q hi_t + hi_q lo_t + lo_q hi_t          (the dropped terms are <= 3.01 * 2^-18 sum |q_i t_i|)
// is accumulated by v_mfma_f32_32x32x16_bf16 on top of |t|^2, with -2 folded into the query operand.  bf16 products are exact
// in f32; the accumulation error and the split error go into the certificate's eps (2^-15 instead of 2^-16 of |q|^2 + max|t|^2,
// DESIGN.md), so the exact re-rank and the rescan of uncertified queries keep the result bit-identical to the oracle's.
// The split image (l2_split_bf16_kernel) has the f32 rows' size: per 16 features 32 B of hi then 32 B of lo, so a lane's A
// fragment of K-step ks is the 16-B slot 4 ks + h (hi) or 4 ks + 2 + h (lo) of its train row -- the staging code, the XOR
// swizzle and the conflict-free ds_read_b128 of the f32 kernel carry over unchanged.
// Each wave owns TWO sets of 32 queries (B operands: 64 VGPRs), so an A fragment feeds two MFMAs and a workgroup covers 256
// queries (half the L2 -> LDS traffic of the f32 kernel).  The fold of step n runs in the shadow of step n+1's MFMAs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef ESFM_BF16_DROP
#define ESFM_BF16_DROP 0         // (experiment: mantissa bits dropped from the bf16 operands -- fewer toggling bits, a higher clock? the bounds follow the stored values)
#endif
__device__ __forceinline__ uint32_t bf16_rne_bits(float a)
{
    const uint32_t u = __float_as_uint(a);
    constexpr int S = 16 + ESFM_BF16_DROP;
    return ((u + ((1u << (S - 1)) - 1u) + ((u >> S) & 1u)) >> S) << ESFM_BF16_DROP;
}
// hi / lo halves of 2 consecutive floats packed into one dword each (element 0 in the low half)
__device__ __forceinline__ void bf16_split2(float a0, float a1, uint32_t &hi, uint32_t &lo)
{
    const uint32_t h0 = bf16_rne_bits(a0), h1 = bf16_rne_bits(a1);
    const float r0 = __fsub_rn(a0, __uint_as_float(h0 << 16)), r1 = __fsub_rn(a1, __uint_as_float(h1 << 16));   // exact
    hi = h0 | (h1 << 16);
    lo = bf16_rne_bits(r0) | (bf16_rne_bits(r1) << 16);
}

// One thread per 16-B piece of a row (4 floats): the load and both stores of a wave are contiguous kilobytes.  A 16-B piece of
// the image holds the hi (or lo) halves of EIGHT floats, so neighbouring lanes swap what the other one assembles: the even lane
// of a pair stores the hi piece, the odd lane the lo piece -- slots 0, 2, 1, 3 of the 64-B group for four consecutive lanes.
// Twice (train image, query image = the same split of -2 x); the 16 lanes of a row also leave |row|^2 (the approximate pass and
// the certificate only need it to 64 u: the summation order is free).  The launch also zeroes the pass's counters (the global
// list's and one per pair): two memset launches less per call.
// (Round 1: one thread per 16-feature group, four loads and eight stores of 16 B at a 64-B lane stride: 31 us per 25 x 4096 rows.)
__global__ __launch_bounds__(256) void l2_split_bf16_kernel(const float4 *__restrict__ desc, long long n_pieces, u32x4 *__restrict__ out,
                                                            u32x4 *__restrict__ out_q, float *__restrict__ norms,
                                                            int32_t *__restrict__ counters, int32_t *__restrict__ pair_cnt, int n_pairs,
                                                            u32x4 *__restrict__ hi_t, u32x4 *__restrict__ hi_q, float *__restrict__ rho_t,
                                                            float *__restrict__ rho_q, int32_t *__restrict__ pair_cnt2)
{
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f < 16) counters[f] = 0;
    if (f < n_pairs) { pair_cnt[f] = 0; if (pair_cnt2) pair_cnt2[f] = 0; }
    const bool ok = f < n_pieces;
    const float4 v = ok ? desc[f] : make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    const bool odd = (threadIdx.x & 1) != 0;
    const long long g = f >> 2;                                  // 16-feature group
    const int slot = (odd ? 2 : 0) + (int)((f >> 1) & 1);        // hi pieces: slots 0, 1; lo pieces: 2, 3
#pragma unroll
    for (int img = 0; img < 2; ++img) {
        const float sc = img == 0 ? 1.f : -2.f;                 // scaling by -2 is exact and commutes with the split
        uint32_t h0, l0, h1, l1;
        bf16_split2(sc * v.x, sc * v.y, h0, l0);
        bf16_split2(sc * v.z, sc * v.w, h1, l1);
        // the even lane needs its partner's hi halves, the odd lane its partner's lo halves
        const uint32_t r0 = __shfl_xor(odd ? h0 : l0, 1), r1 = __shfl_xor(odd ? h1 : l1, 1);
        const u32x4 piece = odd ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1};
        if (ok && out) (img == 0 ? out : out_q)[4 * g + slot] = piece;      // (the hi / lo images: the three-product pass's operands only)
        if (hi_t) {
            // The one-product pass (l2_knn_bf16x1_kernel) multiplies the hi halves only.  Its images are dense -- 128 B per row, the
            // even lane's piece IS the 16-B slot of eight consecutive features -- and its certificate needs |x - hi(x)|_2 of every
            // row in both roles (x = t and x = -2 q: the same number times two, except for denormals).  The residuals are exact in
            // f32; the sum is rounded up by more than its 64-term error.
            if (ok && !odd) (img == 0 ? hi_t : hi_q)[f >> 1] = piece;
            const float e0 = __fsub_rn(sc * v.x, __uint_as_float(h0 << 16)), e1 = __fsub_rn(sc * v.y, __uint_as_float(h0 & 0xFFFF0000u));
            const float e2 = __fsub_rn(sc * v.z, __uint_as_float(h1 << 16)), e3 = __fsub_rn(sc * v.w, __uint_as_float(h1 & 0xFFFF0000u));
            // (summed in double: the squares of residuals below ~1e-19 are denormal or zero in f32, and a residual norm that comes out
            // too small would make the certificate's bound too small)
            double r = (double)e0 * (double)e0 + (double)e1 * (double)e1 + (double)e2 * (double)e2 + (double)e3 * (double)e3;
            r += __shfl_xor(r, 1);
            r += __shfl_xor(r, 2);
            r += __shfl_xor(r, 4);
            r += __shfl_xor(r, 8);
            if (ok && (f & 15) == 0) {
                const double rd = sqrt(r) * 1.0005;
                float rf = (float)rd;
                if ((double)rf < rd) rf = nextafterf(rf, FLT_MAX);     // rounded up
                (img == 0 ? rho_t : rho_q)[f >> 4] = rf;
            }
        }
    }
    if (ok && (f & 15) == 0) norms[f >> 4] = s;
}

__global__ __launch_bounds__(256, 2) void l2_knn_bf16_kernel(const float *__restrict__ desc, const u32x4 *__restrict__ split,
                                                             const u32x4 *__restrict__ split_q, const float *__restrict__ norms, const PairDesc *__restrict__ pairs,
                                                             int n_pairs, int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist,
                                                             int32_t *__restrict__ flagged, int32_t *__restrict__ counters, int flag_cap,
                                                             int32_t *__restrict__ pair_cnt, int32_t *__restrict__ pair_list)
{
    constexpr int TT = 128, NS = 2, GRP = 4;                     // train rows per LDS tile, query sets of 32 per wave, rows per fold group
    constexpr int DIM = 64, QB = 128 * NS, SLOTS = 16, KS = 4;
    constexpr int NDMA = TT / 16;             // LDS-DMA instructions per wave per tile (4 rows = 1 KiB each)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *lds_tile = reinterpret_cast<u32x4 *>(smem);                         // [2][TT*SLOTS]
    float *lds_norm = reinterpret_cast<float *>(smem + 2 * TT * SLOTS * 16);   // [2][TT]   (the asm segment assumes norms right behind the tiles)
    float *lds_red = lds_norm + 2 * TT;                                        // [4]
    float *lds_master = lds_red + 4;                                           // [NS][6][256]: per-thread master top-3 (keys, segments)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int pi = find_pair_by_block(pairs, n_pairs, lb);
    const PairDesc pd = pairs[pi];
    const int nq = pd.nq, nt = pd.nt;
    const float *__restrict__ Q = desc + (size_t)pd.q_row0 * DIM;
    const float *__restrict__ T = desc + (size_t)pd.t_row0 * DIM;
    const float *__restrict__ tn = norms + pd.t_row0;
    const int qbase = (lb - pd.blk_off) * QB + wave * 32 * NS;
    auto row_of_slot = [&](int q) { return q; };

    // Running top-3 per query set, TWO levels deep in the hot loop (l2_segment_gfx950.inc).  A lane's 16 results of a 32-train
    // step are four groups of four consecutive train rows (accumulator registers 4g .. 4g+3 = rows 8g + 4h + 0..3).  Per group:
    // the minimum of the four raw scores (two v_min3_f32 seeded with kBig: a NaN score loses every minimum), the 8-bit position
    // code (6 bits step in segment, 2 bits group) into the low mantissa bits of that minimum (one v_and_or_b32), and the
    // three-smallest network on the group key (three v_med3_f32): 6 VALU per 4 results instead of 16.  The two nearest trains of a
    // query lie in the (at most two) groups with the smallest minima; the third group key bounds every row outside the kept
    // groups, which is what the certificate needs.  The tail re-ranks the kept groups' rows exactly -- four consecutive 256-B rows
    // per group.
    // (Measured on MI355X, scratch/ubench/mfma_issue: in SHADER CYCLES up to six VALU instructions hide behind every bf16 MFMA --
    // the 5.33-per-MFMA fold of round 1 included; what they cost is POWER: the chip is clock-limited on random operands, 1660 TFLOP/s
    // with the 4-per-result fold beside the MFMAs against 1805 with this one and 1690-1940 with none.)
    constexpr float kBig = 3.0e38f;
    constexpr int NG = 16 / GRP;              // groups per lane per 32-train step
    constexpr int kSegSub = 256 / NG;         // steps per segment: the 8-bit code is (step in segment) * NG + group
    constexpr int kSegTiles = kSegSub / (TT / 32);
    // The master top-3 (key, first step of the key's segment) is touched once per segment (2048 trains): it lives in LDS, a
    // private column per thread, so that the main loop's registers go to the pipeline.
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int m = 0; m < 3; ++m) { lds_master[(6 * s + m) * 256 + tid] = kBig; lds_master[(6 * s + 3 + m) * 256 + tid] = __int_as_float(-1); }
    }
    float tmax;
    // the master keeps (key, first step of the key's segment); the group's rows are decoded from the two once, at the end
    struct Master { float v0, v1, v2; int c0, c1, c2; };
    auto master_load = [&](int s) {
        Master m;
        m.v0 = lds_master[(6 * s + 0) * 256 + tid]; m.v1 = lds_master[(6 * s + 1) * 256 + tid]; m.v2 = lds_master[(6 * s + 2) * 256 + tid];
        m.c0 = __float_as_int(lds_master[(6 * s + 3) * 256 + tid]); m.c1 = __float_as_int(lds_master[(6 * s + 4) * 256 + tid]);
        m.c2 = __float_as_int(lds_master[(6 * s + 5) * 256 + tid]);
        return m;
    };
    auto master_store = [&](int s, const Master &m) {
        lds_master[(6 * s + 0) * 256 + tid] = m.v0; lds_master[(6 * s + 1) * 256 + tid] = m.v1; lds_master[(6 * s + 2) * 256 + tid] = m.v2;
        lds_master[(6 * s + 3) * 256 + tid] = __int_as_float(m.c0); lds_master[(6 * s + 4) * 256 + tid] = __int_as_float(m.c1);
        lds_master[(6 * s + 5) * 256 + tid] = __int_as_float(m.c2);
    };
    auto master_insert = [&](Master &m, float key, int seg_sub0 /* wave-uniform */) {
        const bool live = key < 1.0e38f;
        const bool l2 = live && key < m.v2, l1 = live && key < m.v1, l0 = live && key < m.v0;
        const int t2 = l2 ? seg_sub0 : m.c2;
        const int t1 = l1 ? seg_sub0 : m.c1;
        m.c2 = l1 ? m.c1 : t2;
        m.c1 = l0 ? m.c0 : t1;
        m.c0 = l0 ? seg_sub0 : m.c0;
        const float n2 = l2 ? key : m.v2;
        const float n1 = l1 ? key : m.v1;
        m.v2 = l1 ? m.v1 : n2;
        m.v1 = l0 ? m.v0 : n1;
        m.v0 = l0 ? key : m.v0;
    };
    // first of the GRP consecutive train rows of the group a key names (-1: empty slot): accumulator register r holds row
    // (r & 3) + 8 (r >> 2) + 4 h of its step
    auto group_row0_of = [&](float key, int seg_sub0) {
        const int code = (int)(__float_as_uint(key) & 0xFFu);
        const int r0 = GRP * (code % NG);
        return key < 1.0e38f ? (seg_sub0 + code / NG) * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * h : -1;
    };

    const int ntiles = (nt + TT - 1) / TT;
    // Staging is LDS-DMA (buffer_load_dwordx4 ... lds): a wave instruction moves 4 train rows (1 KiB) straight into LDS, lane l
    // to byte 16 l of the destination, so the XOR swizzle is applied on the SOURCE side (lane l fetches slot (l & 15) ^ (row & 15)
    // of its row) -- no staging VGPRs, no ds_write pass.  Rows past nt read as zeros through the buffer descriptor; their norm
    // is kBig.  Tiles 0 and 1 are issued here, tile t + 2 by the segment code when tile t hands its buffer over.
    const u32x4 trsrc = raw_buffer_rsrc(split + (size_t)pd.t_row0 * SLOTS, (uint32_t)nt * (DIM * 4));   // reads past it return 0
    const u32x4 nrsrc = raw_buffer_rsrc(tn, (uint32_t)nt * 4u);
    const uint32_t lds_tile_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_tile);   // LDS byte address
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wrow0 = wave_s * (TT / 4);                                   // this wave stages rows [wrow0, wrow0 + TT / 4) of a tile
    auto dma_tile = [&](int tile, int buf) {
        int voff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wrow0 + 4 * i + (lane >> 4);
            voff[i] = row * (DIM * 4) + (((lane & 15) ^ (row & 15)) * 16);   // rows 16 apart share the swizzle: i and i + 4
        }
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const uint32_t dst = lds_tile_addr + (uint32_t)((buf * TT * SLOTS + (wrow0 + 4 * i) * SLOTS) * 16);
            const int soff = (tile * TT + (i >= 4 ? 16 : 0)) * (DIM * 4);    // wave-uniform
            lds_dma_b128(dst, voff[i & 3], trsrc, soff);
        }
    };
    auto norm_load = [&](int tile) {
        const int t = tile * TT + tid;
        return (tid < TT && t < nt) ? tn[t] : kBig;
    };
    auto norm_store = [&](int buf, float nv) { if (tid < TT) lds_norm[buf * TT + tid] = nv; };

    // rows past nt of the last tile are not transferred (their norm kBig keeps them out of every top-3): what they hold must
    // at least be finite, so the buffers start out zeroed (NaN keys would corrupt the v_med3 network)
    if (ntiles * TT != nt) {
        for (int i = tid; i < 2 * TT * SLOTS; i += 256) lds_tile[i] = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
    if (ntiles > 0) {
        norm_store(0, norm_load(0));
        dma_tile(0, 0);
        if (ntiles > 1) { norm_store(1, norm_load(1)); dma_tile(1, 1); }
    }

    // (issued after the first two tiles' DMA so that their latencies overlap)
    // B operands: -2 q split into hi and lo (the query image of l2_split_bf16_kernel), this lane's 8 features of every K-step
    u32x4 bhi[NS][KS], blo[NS][KS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int qrow = row_of_slot(qbase + 32 * s + j);
        const bool ok = qrow < nq;
        const u32x4 *qp = split_q + ((size_t)pd.q_row0 + (ok ? qrow : 0)) * SLOTS + h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 hi = qp[4 * ks], lo = qp[4 * ks + 2];
            if (!ok) { hi = u32x4{0u, 0u, 0u, 0u}; lo = hi; }
            bhi[s][ks] = hi;
            blo[s][ks] = lo;
        }
    }

    // max |t|^2 over the train set (the certificate's error bound needs it in the tail): reduced here, while the first tiles are
    // on their way, and published through LDS -- the segment code's first barrier orders it for the whole workgroup
    {
        float m = 0.f;
        for (int t = tid; t < nt; t += 256) m = fmaxf(m, tn[t]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) lds_red[wave] = m;
    }

    // Main loop: one hand-scheduled asm block per segment of <= 16 tiles (gen_l2_segment_asm.py has the schedule: software-pipelined
    // by two K-steps, LDS-DMA of tile t + 2 issued when tile t hands its buffer over, one barrier per tile).  It returns the
    // segment's three smallest group keys per query set; they go into the master top-3 between segments.
    for (int t0 = 0; t0 < ntiles; t0 += kSegTiles) {
        const int t1 = min(t0 + kSegTiles, ntiles);
        float k0[NS], k1[NS], k2[NS];
        asm volatile(ESFM_L2_SEGMENT_ASM
                     : "=&v"(k0[0]), "=&v"(k1[0]), "=&v"(k2[0]), "=&v"(k0[1]), "=&v"(k1[1]), "=&v"(k2[1])
                     : "v"(bhi[0][0]), "v"(bhi[0][1]), "v"(bhi[0][2]), "v"(bhi[0][3]), "v"(bhi[1][0]), "v"(bhi[1][1]), "v"(bhi[1][2]), "v"(bhi[1][3]),
                       "v"(blo[0][0]), "v"(blo[0][1]), "v"(blo[0][2]), "v"(blo[0][3]), "v"(blo[1][0]), "v"(blo[1][1]), "v"(blo[1][2]), "v"(blo[1][3]),
                       "s"(t0), "s"(t1), "s"(nt), "s"(trsrc), "s"(nrsrc), "s"(lds_tile_addr), "s"(wave_s)
                     : ESFM_L2_SEGMENT_CLOBBERS);
        const int seg_sub0 = t0 * (TT / 32);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            Master m = master_load(s);
            master_insert(m, k0[s], seg_sub0); master_insert(m, k1[s], seg_sub0); master_insert(m, k2[s], seg_sub0);
            master_store(s, m);
        }
    }

    if (ntiles == 0) __syncthreads();   // no segment ran, so no barrier has published lds_red yet
    tmax = fmaxf(fmaxf(lds_red[0], lds_red[1]), fmaxf(lds_red[2], lds_red[3]));

    // ---- exact re-rank of the kept groups' rows in the oracle's order, certificate ----
    // A query's six kept groups (three per half-wave lane) are ranked by key across the two lanes and dealt out alternately --
    // global rank 2 r + h goes to lane half h in round r -- so the two groups that usually matter cost ONE round of four rows
    // whichever lanes found them.  A group is skipped when it provably cannot hold one of the two nearest: with ka <= kb the two
    // smallest of the six keys (two groups, hence two different rows: the groups' minima), both of those rows' exact d^2 are
    // <= U = |q|^2 + kb + E(kb), E(k) = 2^-15 (|q|^2 + max|t|^2) + 2^-15 |k| being the certificate's bound on |(|q|^2 + key) - d^2|;
    // every row of a group with |q|^2 + k - E(k) > U (1 + 2^-20) -- k its minimum -- is farther than both even after sqrtf's
    // rounding.  Keys only grow with the rank, so the rounds stop at the first one no lane of the wave needs.
    //
    // The rows come in by LDS-DMA (round 2, second half).  With one row per lane a load instruction touches 64 cache lines and the
    // L1 looks up about one line per clock: the 160 such instructions per wave kept the texture path busy for ~21 us per workgroup
    // (measured: 0.40 ms per launch with one workgroup per CU, 0.22 ms with two) and the OTHER workgroup's tile transfers queued
    // behind them -- with wave-uniform (coalesced) addresses in the same instructions the kernel ran 0.10 ms faster.  Now 16 lanes
    // fetch one 256-B row (4 rows = 1 KiB per wave instruction, every line touched once) into the wave's quarter of the idle tile
    // area, XOR-swizzled on the source side like the tiles, and lane l reads "its" row back with 16 conflict-free ds_read_b128; the
    // row of sub-round u + 1 is in flight while row u is compared.  A wave's chain is now latency-bound (ten sub-rounds of
    // ~1.3 us), which costs little: the other workgroup of the CU alone keeps the matrix pipe 93 % busy (measured, one workgroup
    // per CU without tail: 1.39 ms against 1.29).  Measured: 1.50-1.52 -> 1.42-1.44 ms per launch.
    // (Measured and dropped: s_setprio 3 for the main loop / 0 for the tail, 1.50 ms; the query rows by per-lane loads in the
    // shadow of the first row transfer instead of their own sub-round, 1.50 ms -- 32 lines per instruction are enough to disturb
    // the tile transfers again; starting the second workgroup of every CU half a run time late, no gain.)
    // The segment code issues the transfer of tile t + 2 unconditionally (a tile that does not exist reads zeros through the
    // descriptor): the last two of them are still in flight, aimed at rows of the tile area that now become OTHER waves' landing
    // zones -- every wave drains its own before the barrier.
    lds_dma_wait();
    __syncthreads();   // every wave is through its last tile: the tile area becomes four private 16-KiB landing zones
    const u32x4 frsrc_t = raw_buffer_rsrc(T, (uint32_t)nt * 256u);   // rows past the set read as zeros, no memory access
    const u32x4 frsrc_q = raw_buffer_rsrc(Q, (uint32_t)nq * 256u);
    const uint32_t lds_land = lds_tile_addr + (uint32_t)wave_s * 16384u;
    const float4 *land = reinterpret_cast<const float4 *>(smem) + (size_t)wave * 1024;
    int swz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) swz[i] = (4 * i + (lane >> 4)) * 256 + (((lane & 15) ^ ((4 * i + (lane >> 4)) & 15)) * 16);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int qrow = row_of_slot(qbase + 32 * s + j);
        const bool qvalid = qrow < nq;
        // the two best (distance, index, d^2) as plain scalars, updated without a branch (the struct form went through scratch
        // memory here, and every scratch access waits for the row transfer in flight)
        float b0d = FLT_MAX, b1d = FLT_MAX, b0q = 0.f, b1q = 0.f; int b0i = -1, b1i = -1;
        auto insert2 = [&](bool valid, float d, int i, float d2) {
            const bool c1 = valid && (d < b1d || (d == b1d && i < b1i));     // (an empty slot holds FLT_MAX: +inf and NaN never enter, like the oracle's `d < d1`)
            const bool c0 = valid && (d < b0d || (d == b0d && i < b0i));
            b1d = c0 ? b0d : (c1 ? d : b1d); b1i = c0 ? b0i : (c1 ? i : b1i); b1q = c0 ? b0q : (c1 ? d2 : b1q);
            b0d = c0 ? d : b0d; b0i = c0 ? i : b0i; b0q = c0 ? d2 : b0q;
        };
        // the 32 query rows of this set -> landing slots 0..31 (lanes j and j + 32 read the same slot); the group ranking below
        // runs in the transfer's shadow
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
            lds_dma_b128(lds_land + (uint32_t)i * 1024u, (qbase + 32 * s) * 256 + (i >> 2) * 4096 + swz[i & 3], frsrc_q, 0);
        const Master mst = master_load(s);
        const float qnorm_s = norms[pd.q_row0 + (qvalid ? qrow : 0)];
        const float vk[3] = {mst.v0, mst.v1, mst.v2};
        const int g0[3] = {group_row0_of(mst.v0, mst.c0), group_row0_of(mst.v1, mst.c1), group_row0_of(mst.v2, mst.c2)};
        float pk[3]; int pg[3], rank_own[3], rank_par[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { pk[i] = __shfl_xor(vk[i], 32); pg[i] = __shfl_xor(g0[i], 32); }
#pragma unroll
        for (int i = 0; i < 3; ++i) {        // ties between the halves: half 0 first (both lanes must agree on the order)
            rank_own[i] = i; rank_par[i] = i;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                rank_own[i] += (pk[k] < vk[i] || (pk[k] == vk[i] && h == 1)) ? 1 : 0;
                rank_par[i] += (vk[k] < pk[i] || (vk[k] == pk[i] && h == 0)) ? 1 : 0;
            }
        }
        const float kb = fminf(fmaxf(vk[0], pk[0]), fminf(vk[1], pk[1]));
        const double qn = (double)qnorm_s;
        const double e1 = (qn + (double)tmax) * (1.0 / 32768.0);
        constexpr double kTrunc = 1.0001 / 32768.0;
        const double U = (qn + (double)kb + e1 + fabs((double)kb) * kTrunc) * (1.0 + 1.0 / 1048576.0);
        float4 qv[16];
        lds_dma_wait();
#pragma unroll
        for (int c = 0; c < 16; ++c) qv[c] = land[j * 16 + (c ^ (j & 15))];
        for (int r = 0; r < 3; ++r) {
            const int want = 2 * r + h;
            float key = kBig; int row0 = -1;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (rank_own[i] == want) { key = vk[i]; row0 = g0[i]; }
                if (rank_par[i] == want) { key = pk[i]; row0 = pg[i]; }
            }
            const bool cannot = (qn + (double)key - e1 - fabs((double)key) * kTrunc) > U;   // false on NaN: re-rank
            const bool need = row0 >= 0 && qvalid && !cannot;
            if (__ballot(need) == 0ull) break;
            // 16 lanes fetch one 256-B row: DMA instruction i serves the lanes 4 i .. 4 i + 3 (their row of sub-round u)
            const int rsel = need ? row0 : nt;          // nt: past the descriptor, zeros
            int rowsrc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) rowsrc[i] = __builtin_amdgcn_ds_bpermute((4 * i + (lane >> 4)) * 4, rsel) * 256 + (swz[i & 3] & 255);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the zone's previous contents are in registers
#pragma unroll
            for (int i = 0; i < 16; ++i) lds_dma_b128(lds_land + (uint32_t)i * 1024u, rowsrc[i], frsrc_t, 0);
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                float4 ra_[16];
                lds_dma_wait();
#pragma unroll
                for (int c = 0; c < 16; ++c) ra_[c] = land[lane * 16 + (c ^ (lane & 15))];
                if (u + 1 < GRP) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int i = 0; i < 16; ++i) lds_dma_b128(lds_land + (uint32_t)i * 1024u, rowsrc[i] + (u + 1) * 256, frsrc_t, 0);
                }
                const float da = l2sqr64_canonical_regs(qv, ra_);
                const int ta_ = row0 + u;
                insert2(need && ta_ < nt, sqrt_rn_f32(da), ta_, da);
            }
        }
        {
            const float pd0 = __shfl_xor(b0d, 32), pq0 = __shfl_xor(b0q, 32), pd1 = __shfl_xor(b1d, 32), pq1 = __shfl_xor(b1q, 32);
            const int pi0 = __shfl_xor(b0i, 32), pi1 = __shfl_xor(b1i, 32);
            insert2(pi0 >= 0, pd0, pi0, pq0);
            insert2(pi1 >= 0, pd1, pi1, pq1);
        }
        const float tau = fminf(mst.v2, __shfl_xor(mst.v2, 32));
        if (qvalid && h == 0) {
            const size_t o = 2 * ((size_t)pd.out_off + qrow);
            knn_idx[o] = b0i; knn_idx[o + 1] = b1i;
            knn_dist[o] = b0d; knn_dist[o + 1] = b1d;
            bool certified = (tau >= 1.0e38f);       // the empty-slot sentinel; a NaN tau compares false and goes to the re-scan
            if (!certified && b1i >= 0) {
                const double eps = (qn + (double)tmax) * (1.0 / 32768.0) + fabs((double)tau) * (1.0001 / 32768.0);
                certified = (qn + (double)tau - eps) > (double)b1q * (1.0 + 1.0 / 2097152.0);
            }
            if (!certified) {
                const int slot = atomicAdd(&counters[0], 1);
                if (slot < flag_cap) { flagged[2 * slot] = pi; flagged[2 * slot + 1] = qrow; }
                pair_list[pd.out_off + atomicAdd(&pair_cnt[pi], 1)] = qrow;
            }
        }
    }
}

// max |row|^2 and max rho_t of every 256-row block of the bank (l2_knn_bf16x1_kernel takes the maxima over a train set from here: the
// whole blocks inside the set from this table, the rows in front of and behind them one by one).  Launched behind l2_split_bf16_kernel.
__global__ __launch_bounds__(256) void l2_blockmax_kernel(const float *__restrict__ norms, const float *__restrict__ rho_t, long long total_rows,
                                                          float2 *__restrict__ blkmax)
{
    __shared__ float red[8];
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    float m = row < total_rows ? norms[row] : 0.f, r = row < total_rows ? rho_t[row] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); r = fmaxf(r, __shfl_xor(r, o)); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; red[4 + (threadIdx.x >> 6)] = r; }
    __syncthreads();
    if (threadIdx.x == 0)
        blkmax[blockIdx.x] = make_float2(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
}

// The one-product pass's bound E1 on |(|q|^2 + score) - D| for every train row (D = the canonical float d^2; see the comment of
// l2_knn_bf16x1_kernel): operand rounding (rB T + (2 |q| + rB) R), the three-product pass's 2^-15 (|q|^2 + max |t|^2) for norms, MFMA
// accumulation and the canonical distance, and an ABSOLUTE floor of 2^-118 for whatever is flushed to zero or loses bits as a
// denormal on the way (64 products and sums below 2^-126 each, in the norms, the matrix pipe and the residual norms: descriptors of
// magnitude ~1e-20 and less).  Used by the first pass's certificate and ratio screen and by the threshold-filter pass: one formula.
__device__ __forceinline__ double l2x1_e1(double qn, double rq, double sqrt_tmax, double tmax, double rmax)
{
    return (rq * sqrt_tmax + (2.0 * sqrt(qn) + rq) * rmax) * (1.0 + 1.0 / 512.0) + (qn + tmax) * (1.0 / 32768.0) + 0x1p-118;
}

// ---------------------------------------------------------------------------------------------
// The ONE-product distance pass (round 3; reshaped in round 4): q.t ~ bf16(q).bf16(t), four v_mfma_f32_32x32x16_bf16 per 32 x 32 x 64
// tile instead of the three-product pass's twelve, a fused fold that keeps the K = ESFM_L2X1_KEEP smallest GROUP keys per lane and
// query set (groups of ESFM_L2X1_GRP results, the position in the low mantissa bits), and a ratio screen on those keys.
//  * the operand rounding is part of every bound.  With B = bf16(-2 q), a = bf16(t), rB = |(-2 q) - B|_2 and
//    rho_t = |t - a|_2 (both measured per row by l2_split_bf16_kernel):  |(-2 q).t - B.a| <= rB |t| + |B| rho_t, so
//        E1 = (rB T + (2 |q| + rB) R) (1 + 2^-9) + 2^-15 (|q|^2 + max|t|^2) + 2^-118,   T = max |t|,  R = max rho_t  over the train set,
//    bounds |(|q|^2 + score) - d^2| for every train row (l2x1_e1; the 2^-15 term is the three-product pass's whole budget: norms, MFMA
//    accumulation, the canonical distance).  For unit-norm descriptors E1 ~ 0.008 against 6e-5: K = 4 groups per lane push tau -- the
//    bound on every row outside the kept groups -- about as many ranks out as the larger error needs (simulated on M-SURF-4k,
//    scratch/sim_bf16x1_cert.py: K = 3 leaves 6.9 % of the queries uncertified, K = 4 0.6 %, K = 6 0.01 %; the reference's own
//    fountain descriptors 37 % / 15 % / 3.6 %);
//  * this kernel ends with the keys: the screen drops the queries that provably fail the ratio test, every other query leaves a
//    48-byte survivor entry.  l2_finish_kernel does the rest -- exact re-rank of the kept groups in the oracle's order, certificate,
//    threshold-filter pass over what stays uncertified, brute force of what overflows that, ratio test and compaction -- so the
//    result stays bit-identical to the oracle whatever the data;
//  * 512 queries per item (four sets of 32 per wave), a ring of two 32-KiB tiles of bf16(t) rows fed by LDS-DMA, 12-bit position
//    codes in the group keys: l2x1_segment_gfx950.inc (gen_l2x1_segment_asm.py) is the whole main loop.  Train sets the code cannot
//    number skip this pass (l2_x1_supported).
constexpr int l2x1_query_block_c = 128 * ESFM_L2X1_SETS;
// Cross-lane moves without an address register (__shfl_xor goes through ds_bpermute_b32, whose lane addresses the compiler hoists out
// of l2_knn_bf16x1_kernel's item loop and then has to keep in scratch memory across the main loop's asm block).
// max over the wave, the same value in every lane: rotations inside the rows of 16 lanes (DPP), then the four rows through SGPRs
__device__ __forceinline__ float wave_max_dpp(float x)
{
#define ESFM_ROR(n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + (n), 0xF, 0xF, false))
    x = fmaxf(x, ESFM_ROR(8)); x = fmaxf(x, ESFM_ROR(4)); x = fmaxf(x, ESFM_ROR(2)); x = fmaxf(x, ESFM_ROR(1));
#undef ESFM_ROR
    const int xi = __float_as_int(x);
    const float a = __int_as_float(__builtin_amdgcn_readlane(xi, 0)), b = __int_as_float(__builtin_amdgcn_readlane(xi, 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(xi, 32)), d = __int_as_float(__builtin_amdgcn_readlane(xi, 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
// the value of lane ^ 32 (v_permlane32_swap, gfx950: the upper half of its first operand changes places with the lower half of the
// second); upper = this lane is in the upper half
__device__ __forceinline__ float other_half(float x, bool upper)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(upper ? r[0] : r[1]);
}
// One unit of the one-product pass's work: 512 queries (block qblk) of pair pi against the pair's whole train set.
struct X1Item { int32_t q_row0, nq, t_row0, nt; int64_t out_off; int32_t pi, qblk; };

__global__ __launch_bounds__(256, 2) void l2_knn_bf16x1_kernel(const float *__restrict__ desc, const u32x4 *__restrict__ hi_t,
                                                               const u32x4 *__restrict__ hi_q, const float *__restrict__ norms,
                                                               const float *__restrict__ rho_t, const float *__restrict__ rho_q,
                                                               const float2 *__restrict__ blkmax,
                                                               const PairDesc *__restrict__ pairs, const int32_t *__restrict__ blk_pair, int n_blocks,
                                                               int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist,
                                                               int32_t *__restrict__ counters, int flag_cap,
                                                               int32_t *__restrict__ surv_cnt, float4 *__restrict__ surv_list,
                                                               double ratio2m, int markers, int32_t *__restrict__ rejected,
                                                               int32_t *__restrict__ zero_a, int32_t *__restrict__ zero_b, int zero_n,
                                                               int32_t *__restrict__ zero_counters)
{
    // The per-pair list counters and the global counters exist twice: this launch fills one phase and zeroes the other for the NEXT
    // call (whose finish kernel needs them immutable while it runs) -- no memset launch, no zeroing pass in front of this one.
    if (threadIdx.x == 0) for (int e = blockIdx.x; e < zero_n; e += gridDim.x) { zero_a[e] = 0; zero_b[e] = 0; }
    if (blockIdx.x == 0 && threadIdx.x < 16) zero_counters[threadIdx.x] = 0;
    constexpr int TT = ESFM_L2X1_TT, NS = ESFM_L2X1_SETS, K = ESFM_L2X1_KEEP, RING = ESFM_L2X1_RING;
    constexpr int QB = 128 * NS, HS = 8;                         // HS: 16-B slots per row of the hi images
    constexpr int TILE_BYTES = TT * HS * 16;
    static_assert(NS == 4, "operand list below is written for four query sets");
    static_assert(RING * TT == 2 * 256, "a thread stages two norms of the ring's first tiles");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *lds_tile = reinterpret_cast<u32x4 *>(smem);                         // [RING][TT * HS]: 64 KiB; the main loop leaves the keys here
    float *lds_norm = reinterpret_cast<float *>(smem + RING * TILE_BYTES);     // [RING][TT]   (the asm block assumes norms right behind the ring)
    float *lds_red = lds_norm + RING * TT;                                     // [2][8]: max |t|^2 and max rho_t of an item's train set, per wave
    float *lds_qn = lds_red + 16;                                              // [2][QB]: |q|^2 of an item's queries
    float *lds_rq = lds_qn + 2 * QB;                                           // [2][QB]: their residual norms

    // Nothing that depends on the thread index may stay live across the main loop's asm block (64 operand registers in, 184
    // clobbered: whatever the compiler keeps it keeps in scratch memory and fetches back in the tail's critical path): the thread's
    // coordinates are re-derived from an opaque copy of the index behind every pass of the block.
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto rederive = [&]() {
        int l;             // (the lane index from scratch: not even the thread index has to survive the block)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        lane = l; wave = wave_s; tid = wave_s * 64 + l; j = l & 31; h = l >> 5;
    };
    const uint32_t lds_tile_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_tile);   // LDS byte address

    // ---- A workgroup takes the 512-query blocks b, b + G, b + 2 G, ... of the launch (G = gridDim.x, a multiple of 8 when there is
    // more than one round: all of a workgroup's blocks map to its XCD's contiguous range of the numbering, xcd_remap) and overlaps the
    // memory round trips of block k + 1's set-up -- query operands, the ring's first tiles, norms, maxima -- with block k's tail.
    // The launcher's default is G = number of blocks (one block per workgroup, the loop runs once): persistent workgroups, two per
    // CU, were measured 2 - 4 % SLOWER on the metric's workload and on config 4 (0.517 - 0.540 against 0.498 - 0.513 ms; 191.5 against
    // 187.9 ms) although they take a block's set-up from 16 us to one round trip -- the hardware's dispatcher refills a CU the
    // moment a workgroup leaves, whatever the other one is doing, and a lone workgroup runs its main loop 60 % faster, so the set-up
    // was hidden already; what the static schedule adds is the last round's imbalance.  (ESFM_X1_GRID sets G for measurements; what
    // the restructuring did buy is the short set-up itself: a block -> pair table instead of nine dependent loads of a binary search,
    // maxima from a per-256-row table, 5 % on the launch.)
    const int G = gridDim.x;
    auto lb_of = [&](int k) -> int {
        const long long v = (long long)blockIdx.x + (long long)k * G;
        return v < (long long)n_blocks ? xcd_remap((int)v, n_blocks) : -1;
    };
    auto make_item = [&](int lb, int pi) -> X1Item {
        X1Item it = {0, 0, 0, 0, 0, -1, 0};
        if (pi >= 0) {
            const PairDesc d = pairs[pi];
            it = X1Item{__builtin_amdgcn_readfirstlane(d.q_row0), __builtin_amdgcn_readfirstlane(d.nq), __builtin_amdgcn_readfirstlane(d.t_row0),
                        __builtin_amdgcn_readfirstlane(d.nt),
                        (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)d.out_off >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)d.out_off)),
                        pi, lb - __builtin_amdgcn_readfirstlane(d.blk_off2)};
        }
        return it;
    };

    // what the set-up of an item leaves in registers until its round trip is over
    u32x4 bq[NS][4];                     // B operands: bf16(-2 q), this lane's 8 features of every K-step
    float st_norm[2], st_qn[2], st_rq[2], st_m, st_r;
    // set-up, part 1: every load of the item goes out (and the ring's first tiles: LDS-DMA, wave w rows [TT/4 w, TT/4 (w + 1)) of a
    // tile, 8 rows = 1 KiB per instruction, lane l -> row l >> 3, physical slot l & 7 = logical slot (l & 7) ^ ((row >> 1) & 7))
    auto stage_issue = [&](const X1Item &it) {
        const int nt = it.nt, nq = it.nq;
        const int ntiles = (nt + TT - 1) / TT;
        const u32x4 trsrc = raw_buffer_rsrc(hi_t + (size_t)it.t_row0 * HS, (uint32_t)nt * (HS * 16));   // reads past it return 0
        // LDS under rows that are never transferred (past nt in the last tile) must not hold huge values (the previous item's keys)
        if (ntiles * TT != nt || ntiles < RING) {
            uint32_t z;          // (as a hoisted constant vector the zeros would be carried through the main loop's asm block)
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));
            for (int i = tid; i < RING * TT * HS; i += 256) lds_tile[i] = u32x4{z, z, z, z};
            __syncthreads();
        }
#pragma unroll
        for (int b = 0; b < RING; ++b) {
#pragma unroll
            for (int i = 0; i < TT / 32; ++i) {
                const int row = wave_s * (TT / 4) + 8 * i + (lane >> 3);
                const int voff = row * (HS * 16) + (((lane & 7) ^ ((row >> 1) & 7)) * 16);
                lds_dma_b128(lds_tile_addr + (uint32_t)(b * TILE_BYTES + (wave_s * (TT / 4) + 8 * i) * (HS * 16)), voff, trsrc, b * TILE_BYTES);
            }
        }
        const int qbase = it.qblk * QB + wave * 32 * NS;
        const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(hi_q + (size_t)it.q_row0 * HS), 0, nq * (HS * 16), 0x00020000);   // rows past nq read as zeros
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int voff = (qbase + 32 * s + j) * (HS * 16) + h * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bq[s][ks] = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, voff + 32 * ks, 0, 0);
        }
        const float *__restrict__ tn = norms + it.t_row0;
        const float *__restrict__ tr = rho_t + it.t_row0;
        float big;           // (kBig out of an SGPR written here: as a hoisted constant it would be one more register to carry through the block)
        asm volatile("s_mov_b32 %0, 0x7f61b1e6" : "=s"(big));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + 256 * u, q = it.qblk * QB + t;
            st_norm[u] = t < nt ? tn[t] : big;            // norms of the ring's first tiles, rows past nt as kBig
            st_qn[u] = norms[it.q_row0 + (q < nq ? q : 0)];
            st_rq[u] = rho_q[it.q_row0 + (q < nq ? q : 0)];
        }
        // max |t|^2 and max rho_t over the train set (the certificate's bound needs both in the tail): whole 256-row blocks of the
        // bank from l2_blockmax_kernel's table, the rows in front of and behind them one by one
        float m = 0.f, r = 0.f;
        const int t0 = it.t_row0, t1 = t0 + nt, b0 = (t0 + 255) >> 8, b1 = t1 >> 8;
        if (b0 >= b1) {
            for (int t = tid; t < nt; t += 256) { m = fmaxf(m, tn[t]); r = fmaxf(r, tr[t]); }      // < 512 rows
        } else {
            if (tid < b0 * 256 - t0) { m = fmaxf(m, tn[tid]); r = fmaxf(r, tr[tid]); }
            if (tid < t1 - b1 * 256) { m = fmaxf(m, norms[b1 * 256 + tid]); r = fmaxf(r, rho_t[b1 * 256 + tid]); }
            for (int b = b0 + tid; b < b1; b += 256) { const float2 v = blkmax[b]; m = fmaxf(m, v.x); r = fmaxf(r, v.y); }
        }
        st_m = m; st_r = r;
    };
    // set-up, part 2 (every load has landed): what the main loop and the tail read from LDS
    auto stage_commit = [&](int par) {
        lds_norm[tid] = st_norm[0]; lds_norm[tid + 256] = st_norm[1];
        lds_qn[par * QB + tid] = st_qn[0]; lds_qn[par * QB + tid + 256] = st_qn[1];
        lds_rq[par * QB + tid] = st_rq[0]; lds_rq[par * QB + tid + 256] = st_rq[1];
        const float m = wave_max_dpp(st_m), r = wave_max_dpp(st_r);
        if (lane == 0) { lds_red[par * 8 + wave] = m; lds_red[par * 8 + 4 + wave] = r; }
    };

    // items k, k + 1 and the pair index of item k + 2 (the table look-ups run two items ahead, the pair descriptors one)
    int lb_c = lb_of(0), lb_n = lb_of(1), lb_nn = lb_of(2);
    if (lb_c < 0) return;
    int pi_nn = lb_nn >= 0 ? blk_pair[lb_nn] : -1;
    X1Item cur = make_item(lb_c, blk_pair[lb_c]);
    X1Item nxt = make_item(lb_n, lb_n >= 0 ? blk_pair[lb_n] : -1);
#ifdef ESFM_X1_TRACE
    uint64_t trA = __builtin_amdgcn_s_memrealtime();
    const uint64_t tr_first = trA;
    int tr_setup = 0, tr_loop = 0, tr_tail = 0, tr_clk = 0;
#endif
    stage_issue(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage_commit(0);
    __syncthreads();

    static_assert(K == 4, "a survivor entry carries a lane's keys as one 16-byte piece");
    constexpr double kTrunc = 1.0001 / (double)(1 << (23 - ESFM_L2X1_CODE_BITS));     // the key's mantissa bits under the position code
    for (int k = 0;; ++k) {
        const int par = k & 1;
        const int nq = cur.nq, nt = cur.nt, pi = cur.pi;
        const int ntiles = (nt + TT - 1) / TT;
#ifdef ESFM_X1_TRACE
        { const uint64_t t = __builtin_amdgcn_s_memrealtime(); tr_setup += (int)(t - trA); trA = t; }
        const uint64_t clk0 = __builtin_amdgcn_s_memtime();
#endif
        if (ntiles > 0) {
            const u32x4 trsrc = raw_buffer_rsrc(hi_t + (size_t)cur.t_row0 * HS, (uint32_t)nt * (HS * 16));
            const u32x4 nrsrc = raw_buffer_rsrc(norms + cur.t_row0, (uint32_t)nt * 4u);
            asm volatile(ESFM_L2X1_SEGMENT_ASM
                         :
                         : "v"(bq[0][0]), "v"(bq[0][1]), "v"(bq[0][2]), "v"(bq[0][3]), "v"(bq[1][0]), "v"(bq[1][1]), "v"(bq[1][2]), "v"(bq[1][3]),
                           "v"(bq[2][0]), "v"(bq[2][1]), "v"(bq[2][2]), "v"(bq[2][3]), "v"(bq[3][0]), "v"(bq[3][1]), "v"(bq[3][2]), "v"(bq[3][3]),
                           "s"(ntiles), "s"(nt), "s"(trsrc), "s"(nrsrc), "s"(lds_tile_addr), "s"(wave_s)
                         : ESFM_L2X1_SEGMENT_CLOBBERS);
        }
        rederive();
        st_norm[0] = st_norm[1] = st_qn[0] = st_qn[1] = st_rq[0] = st_rq[1] = st_m = st_r = 0.f;      // (dead here: not carried through the block)
#ifdef ESFM_X1_TRACE
        { const uint64_t t = __builtin_amdgcn_s_memrealtime(); tr_loop += (int)(t - trA); trA = t; tr_clk += (int)((__builtin_amdgcn_s_memtime() - clk0) >> 4); }
#endif
        // the block left this thread's keys in LDS: key i of set s at float (K s + i) * 256 + tid  (replaced when no tile ran)
        float keys[NS][K];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                keys[s][i] = reinterpret_cast<const float *>(smem)[(K * s + i) * 256 + tid];
            }
        }
        if (ntiles == 0) {
            float big;
            asm volatile("s_mov_b32 %0, 0x7f61b1e6" : "=s"(big));
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int i = 0; i < K; ++i) keys[s][i] = big;
        }
        const float tmax = fmaxf(fmaxf(lds_red[par * 8 + 0], lds_red[par * 8 + 1]), fmaxf(lds_red[par * 8 + 2], lds_red[par * 8 + 3]));
        const float rmax = fmaxf(fmaxf(lds_red[par * 8 + 4], lds_red[par * 8 + 5]), fmaxf(lds_red[par * 8 + 6], lds_red[par * 8 + 7]));
        const int qbase = cur.qblk * QB + wave * 32 * NS;
        float qn_s[NS], rq_s[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { qn_s[s] = lds_qn[par * QB + wave * 32 * NS + 32 * s + j]; rq_s[s] = lds_rq[par * QB + wave * 32 * NS + 32 * s + j]; }
        __syncthreads();                  // the ring (and the keys in it) is free: every wave has left the main loop and read its keys
#ifdef ESFM_X1_NOTAIL
        const bool notail = n_blocks >= 0;             // (timing experiments: the kernel without its tail)
#else
        const bool notail = false;
#endif

        // ---- the RATIO SCREEN (round 4), all that is left of this kernel's tail.  The reference keeps a query only if d0 < ratio d1
        // (feature_matching.cpp:133); everything else is dropped one kernel later, and on the metric's workload that is > 90 % of the
        // queries.  With k0 <= kb the two smallest of a query's 2 K group keys (two groups, hence two different train rows: the groups'
        // minima) and E1 the pass's bound on |(|q|^2 + score) - D| (D = the canonical float d^2), kTrunc the key's truncation:
        //     every train row has   D >= L0 = |q|^2 + k0 - kTrunc |k0| - E1        (k0 is the smallest key of all groups),
        //     two rows have         D <= U1 = |q|^2 + kb + kTrunc |kb| + E1,
        // so the nearest has D0 >= L0 and the second nearest D1 <= U1.  If L0 >= ratio^2 (1 + 2^-20) U1 then sqrtf(D0) >= ratio sqrtf(D1)
        // whatever the two roundings of sqrtf and the double product do (their relative error is < 2^-22 together): the query cannot
        // pass the test.  It gets a marker (train index -2) and nothing more is done for it.  A SURVIVOR leaves one 48-byte entry -- its
        // 2 K keys, its row, |q|^2 and E1 (as a float, rounded up) -- in the pair's slice of surv_list; l2_finish_kernel re-ranks the
        // survivors' kept groups exactly.  (Until the middle of round 4 the re-rank ran here, per wave, beside the other workgroup's
        // main loop whose VALU and LDS ports it shares: 0.06 - 0.075 of 0.605 ms, measured against a build without it; in a kernel of its
        // own it has the chip to itself.)  ratio2m = ratio^2 (1 + 2^-20); +inf switches the screen off (the knn2 entry points).
        const double sqrt_tmax = sqrt((double)tmax);
        int nsurv = 0;
        int myslot[NS];
        float e1_s[NS];
        uint32_t rejmask = 0;
        if (!notail) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int qrow = qbase + 32 * s + j;
                const bool qvalid = qrow < nq;
                const float v0 = keys[s][0], v1 = keys[s][1];
                const float p0 = other_half(v0, h != 0), p1 = other_half(v1, h != 0);
                const float k0 = fminf(v0, p0), kb = fminf(fmaxf(v0, p0), fminf(v1, p1));     // the two smallest of the 2 K keys
                const double qn = (double)qn_s[s];
                const double e1 = l2x1_e1(qn, (double)rq_s[s], sqrt_tmax, (double)tmax, (double)rmax);
                const double L0 = qn + (double)k0 - e1 - fabs((double)k0) * kTrunc;
                const double U1 = qn + (double)kb + e1 + fabs((double)kb) * kTrunc;
                const bool rej = qvalid && (L0 >= ratio2m * U1);                              // false on NaN / inf: re-rank
                const bool surv = qvalid && !rej;
                const uint32_t m = (uint32_t)__ballot(surv);                                  // lanes 0 .. 31 (both halves agree)
                myslot[s] = surv ? nsurv + __popc(m & ((1u << j) - 1u)) : -1;
                float ef = (float)e1;
                if ((double)ef < e1) ef = __uint_as_float(__float_as_uint(ef) + 1u);          // rounded up: e1 > 0, and ef < e1 makes ef finite (NaN: every compare false)
                e1_s[s] = ef;
                if (rej) rejmask |= 1u << s;
                nsurv += __popc(m);
            }
        }
        // the one round trip of the tail -- the survivors' place in the pair's slice -- and the next item's set-up share their latency
        int base = 0;
        if (nsurv > 0 && lane == 0) { base = atomicAdd(&surv_cnt[pi], nsurv); atomicAdd(&counters[2], nsurv); }
        const bool more = nxt.pi >= 0;
        const int lb_3 = lb_of(k + 3);
        int pi_3 = -1;
        X1Item nn = {0, 0, 0, 0, 0, -1, 0};
        if (more) {
            stage_issue(nxt);
            nn = make_item(lb_nn, pi_nn);
            pi_3 = lb_3 >= 0 ? blk_pair[lb_3] : -1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more) stage_commit(par ^ 1);
        // the tail's stores go out last: nothing waits for them (the main loop's first hand-over, eight steps on, finds them done)
        if (!notail) {
            float fltmax;        // (FLT_MAX out of an SGPR written here, like kBig in the set-up)
            asm volatile("s_mov_b32 %0, 0x7f7fffff" : "=s"(fltmax));
            int minus2;
            asm volatile("s_mov_b32 %0, -2" : "=s"(minus2));
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int qrow = qbase + 32 * s + j;
                if (markers && h == 0 && ((rejmask >> s) & 1u)) {
                    const size_t o = 2 * ((size_t)cur.out_off + qrow);
                    *reinterpret_cast<int2 *>(knn_idx + o) = make_int2(minus2, minus2);
                    *reinterpret_cast<float2 *>(knn_dist + o) = make_float2(fltmax, fltmax);
                    if (rejected) {          // audit of the screen: what it dropped, on the global list
                        const int slot = atomicAdd(&counters[0], 1);
                        if (slot < flag_cap) { rejected[2 * slot] = pi; rejected[2 * slot + 1] = qrow; }
                    }
                }
            }
            if (nsurv > 0) {
                base = __builtin_amdgcn_readfirstlane(base);
                float4 *ent = surv_list + 3 * ((size_t)cur.out_off + base);                   // (a pair's slice holds nq entries: it cannot overflow)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (myslot[s] >= 0) {
                        ent[3 * myslot[s] + h] = make_float4(keys[s][0], keys[s][1], keys[s][2], keys[s][3]);
                        if (h == 0) ent[3 * myslot[s] + 2] = make_float4(__int_as_float(qbase + 32 * s + j), qn_s[s], e1_s[s], 0.f);
                    }
                }
            }
        }
#ifdef ESFM_X1_TRACE
        { const uint64_t t = __builtin_amdgcn_s_memrealtime(); tr_tail += (int)(t - trA); trA = t; }
#endif
        if (!more) break;
        __syncthreads();                  // the next item's norms, maxima and query norms are in LDS, its first tile has landed
        cur = nxt; nxt = nn; lb_nn = lb_3; pi_nn = pi_3;
    }
#ifdef ESFM_X1_TRACE
    if (lane == 0) {     // per-wave stage times, 10-ns ticks (scratch/x1_trace.py)
        atomicAdd(&counters[8], tr_setup); atomicAdd(&counters[9], tr_loop); atomicAdd(&counters[10], tr_tail); atomicAdd(&counters[11], 1);
        atomicMax(&counters[12], (int)(trA & 0x3fffffffu)); atomicMax(&counters[13], 0x40000000 - (int)(tr_first & 0x3fffffffu));
        atomicMax(&counters[14], (int)(trA - tr_first)); atomicAdd(&counters[15], tr_clk >> 4);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// coherent (agent-scope, relaxed) accesses to what one workgroup of a pair writes and another reads inside l2_finish_kernel's launch
__device__ __forceinline__ void st_coh_i(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh_f(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t ld_coh_i(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_coh_f(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// N x 16 contiguous bytes (dword aligned) with agent-scope coherence (`sc1`: the load is served past this XCD's L2), all in flight at once
template <int N>
__device__ __forceinline__ void ld_coh_block(const void *p, uint32_t *out /* 4 N */)
{
    static_assert(N == 4 || N == 8 || N == 2, "offsets below are immediates");
    u32x4 v[N];
    if constexpr (N == 8)
        asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
                     "global_load_dwordx4 %3, %8, off offset:48 sc1\n\tglobal_load_dwordx4 %4, %8, off offset:64 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
                     "global_load_dwordx4 %6, %8, off offset:96 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:112 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p) : "memory");
    else if constexpr (N == 4)
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                     "global_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]) : "v"(p) : "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) { out[4 * i] = v[i][0]; out[4 * i + 1] = v[i][1]; out[4 * i + 2] = v[i][2]; out[4 * i + 3] = v[i][3]; }
}

// ---------------------------------------------------------------------------------------------
// Lowe ratio test (feature_matching.cpp:88 / :133: float < double * float, i.e. in double) and an order-preserving compaction of one
// pair's survivors, by the THREADS threads of a workgroup, THREADS * kRatioPer queries per sweep (one round of loads for a 4096-row
// set in either instantiation).  A thread takes kRatioPer CONSECUTIVE queries (their 2-NN records are
// 32 + 32 contiguous bytes), so the survivors' order is thread order, then query order inside the thread: an exclusive scan of the
// threads' counts places them.  A train index < 0 (no neighbour; -2: dropped by the one-product pass's ratio screen) never passes; a
// SECOND index of -3 says that pass has proved d0 < ratio d1 without looking for the second neighbour.
// (Round 1: 256 threads, one query each, 16 sweeps of three barriers for a 4096-row set: 14 us per launch.)
template <int THREADS, int kRatioPer, bool COHERENT = false>
__device__ __forceinline__ void ratio_compact_pair(const PairDesc &pd, const int32_t *__restrict__ knn_idx, const float *__restrict__ knn_dist,
                                                   double ratio, int32_t *__restrict__ query_idx, int32_t *__restrict__ train_idx,
                                                   float *__restrict__ distance, int32_t *__restrict__ n_out_p, int *s_wave /* [THREADS / 64] */,
                                                   int *s_base)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *s_base = 0;
    __syncthreads();
    for (int q0 = 0; q0 < pd.nq; q0 += THREADS * kRatioPer) {
        const int qa = q0 + tid * kRatioPer;
        int ti[kRatioPer]; float d0[kRatioPer]; bool pass[kRatioPer];
        int cnt = 0;
        // COHERENT: the records may have been written by another workgroup of this launch (l2_finish_kernel, through write-through
        // stores): they are read past this XCD's L2 -- `sc1` loads -- sixteen bytes at a time (one asm statement per array: as
        // 4-byte atomic loads the same reads took 80 us per launch)
        int iv[2 * kRatioPer]; float dv[2 * kRatioPer];
        if (COHERENT && (kRatioPer % 2) == 0 && qa + kRatioPer <= pd.nq) {
            ld_coh_block<kRatioPer / 2>(knn_idx + 2 * ((size_t)pd.out_off + qa), reinterpret_cast<uint32_t *>(iv));
            ld_coh_block<kRatioPer / 2>(knn_dist + 2 * ((size_t)pd.out_off + qa), reinterpret_cast<uint32_t *>(dv));
        } else {
#pragma unroll
            for (int u = 0; u < kRatioPer; ++u) {
                const size_t o = 2 * ((size_t)pd.out_off + min(qa + u, max(pd.nq - 1, 0)));
                iv[2 * u] = COHERENT ? ld_coh_i(knn_idx + o) : knn_idx[o]; iv[2 * u + 1] = COHERENT ? ld_coh_i(knn_idx + o + 1) : knn_idx[o + 1];
                dv[2 * u] = COHERENT ? ld_coh_f(knn_dist + o) : knn_dist[o]; dv[2 * u + 1] = COHERENT ? ld_coh_f(knn_dist + o + 1) : knn_dist[o + 1];
            }
        }
#pragma unroll
        for (int u = 0; u < kRatioPer; ++u) {
            const int q = qa + u;
            const int i0 = iv[2 * u], i1 = iv[2 * u + 1];
            const float d1 = dv[2 * u + 1];
            d0[u] = dv[2 * u]; ti[u] = i0;
            pass[u] = q < pd.nq && (i0 >= 0) && (i1 == -3 || (i1 >= 0 && (double)d0[u] < ratio * (double)d1));     // -3: the one-product pass proved the test
            cnt += pass[u] ? 1 : 0;
        }
        // exclusive scan of cnt over the workgroup: inside the wave by shuffles, across waves through LDS
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int off = *s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        size_t o = (size_t)pd.out_off + off + (incl - cnt);
#pragma unroll
        for (int u = 0; u < kRatioPer; ++u) {
            if (pass[u]) { query_idx[o] = qa + u; train_idx[o] = ti[u]; distance[o] = d0[u]; ++o; }
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < THREADS / 64; ++w) t += s_wave[w]; *s_base += t; }
        __syncthreads();
    }
    if (tid == 0) *n_out_p = *s_base;
}

// The same over the ratio screen's SURVIVORS only (l2_finish_kernel when the screen ran): whatever is not on the pair's survivor list
// has been dropped by the screen and has no record at all (the one-product pass writes no markers outside the audit modes: 16 bytes
// per query it does not store and this stage does not read -- 19.6 of 19.7 MB per step on the metric's workload).  A sweep of
// THREADS * 16 queries: the survivors' rows set bits in an LDS bitmap, a thread looks at its 16 consecutive queries' bits and reads
// the records of the set ones; order and compaction as above.
template <int THREADS>
__device__ __forceinline__ void ratio_compact_pair_sparse(const PairDesc &pd, const float4 *__restrict__ ent, int nsv,
                                                          const int32_t *__restrict__ knn_idx, const float *__restrict__ knn_dist,
                                                          double ratio, int32_t *__restrict__ query_idx, int32_t *__restrict__ train_idx,
                                                          float *__restrict__ distance, int32_t *__restrict__ n_out_p,
                                                          uint32_t *s_bits /* [THREADS / 2] */, int *s_wave /* [THREADS / 64] */, int *s_base)
{
    constexpr int PER = 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *s_base = 0;
    for (int q0 = 0; q0 < pd.nq; q0 += THREADS * PER) {
        for (int w = tid; w < THREADS / 2; w += THREADS) s_bits[w] = 0u;
        __syncthreads();
        for (int k = tid; k < nsv; k += THREADS) {
            const uint32_t r = (uint32_t)(__float_as_int(ent[3 * (size_t)k + 2].x) - q0);
            if (r < (uint32_t)(THREADS * PER)) atomicOr(&s_bits[r >> 5], 1u << (r & 31));
        }
        __syncthreads();
        const int qa = q0 + tid * PER;
        uint32_t bits = (s_bits[tid >> 1] >> ((tid & 1) * PER)) & 0xFFFFu;
        int ti[PER]; float d0[PER];
        uint32_t passm = 0;
        for (uint32_t b = bits; b; b &= b - 1) {
            const int u = __ffs(b) - 1;
            const size_t o = 2 * ((size_t)pd.out_off + qa + u);
            uint32_t iv[4];          // (records written by other workgroups of this launch: read past this XCD's L2)
            asm volatile("global_load_dwordx2 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(*reinterpret_cast<uint2 *>(iv)), "=&v"(*reinterpret_cast<uint2 *>(iv + 2)) : "v"(knn_idx + o), "v"(knn_dist + o) : "memory");
            const int i0 = (int)iv[0], i1 = (int)iv[1];
            const float dd0 = __uint_as_float(iv[2]), dd1 = __uint_as_float(iv[3]);
            const bool pass = qa + u < pd.nq && (i0 >= 0) && (i1 == -3 || (i1 >= 0 && (double)dd0 < ratio * (double)dd1));
            // (static indexing keeps ti / d0 in registers)
#pragma unroll
            for (int e = 0; e < PER; ++e) if (e == u) { ti[e] = i0; d0[e] = dd0; }
            passm |= pass ? (1u << u) : 0u;
        }
        const int cnt = __popc(passm);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int off = *s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        size_t o = (size_t)pd.out_off + off + (incl - cnt);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if ((passm >> u) & 1u) { query_idx[o] = qa + u; train_idx[o] = ti[u]; distance[o] = d0[u]; ++o; }
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < THREADS / 64; ++w) t += s_wave[w]; *s_base += t; }
        __syncthreads();
    }
    if (tid == 0) *n_out_p = *s_base;
}

// ---------------------------------------------------------------------------------------------
// Everything behind the one-product pass's main loop and ratio screen in ONE launch (round 4; round 3 had three: a threshold-filter kernel,
// l2_rescan64_pairs_kernel, ratio_compact_kernel, and every launch boundary costs 5 - 10 us on this part):
//   (1) the exact RE-RANK of the screen's survivors (surv_cnt / surv_list: one 48-byte entry per survivor): the rows of the kept
//       groups in the oracle's order, the certificate, and the ratio verdicts that make most second neighbours unnecessary
//       (finish_rerank_vset below).  Uncertified and undecided queries go on the pair's list (unc_cnt / unc_list, knn_d2);
//   (2) the second pass over that list -- a THRESHOLD FILTER instead of a second top-k.  For such a query (1) has left the exact
//       two best of its candidates; U = an upper bound of the second-best d^2.  Every train row that can still change the answer
//       has d^2 <= U, hence a one-product score s <= U - |q|^2 + E1 (E1: l2x1_e1).  So: the same bf16(-2 q).bf16(t) product on the
//       matrix cores over the whole train set, a compare of every score with the query's threshold, the few rows that pass (0.1 -
//       1.2 per query on the data simulated in scratch/sim_bf16x1_cert.py) evaluated exactly in the oracle's order and merged;
//   (3) the exact brute force of a chunk whose hit list overflows (adversarial inputs: every row inside the error);
//   (4) the ratio test and the ordered compaction of the pair's survivors (ratio_compact_pair).
// Work split: S workgroups per pair (blockIdx = slice * n_pairs + pair).  Stage (1) is shared: the pair's virtual sets of seven
// survivors are dealt out over the S x 4 waves.  Then every workgroup arrives at the pair's counter and the LAST one runs (2) - (4)
// alone (after the screen and the verdicts (2) and (3) see a handful of queries per launch).  Nobody waits for anybody: no
// assumption about which workgroups are resident.  What stage (1) writes for the last workgroup -- results, list entries -- travels
// through relaxed agent-scope atomics (write-through stores, L2-bypassing loads), not through agent-scope fences: on this part a
// release is a write-back of the XCD's whole L2, an acquire an invalidation, and hundreds of workgroups would queue for them.
#ifndef ESFM_FIN_THREADS
#define ESFM_FIN_THREADS 256
#endif
constexpr int kFinThreads = ESFM_FIN_THREADS, kFinCap = 2048, kFinWaves = kFinThreads / 64;

// exact 2-NN of up to 32 listed queries of one pair by the whole workgroup, the oracle's arithmetic and (distance, index) order:
// thread = train row (16 x 16 B in registers), the queries as LDS broadcasts; the 64 keys of a wave's rows are reduced to the two
// smallest by shuffles and merged into lane k's running pair for query k.  The fallback of the fallback: ~3 us per query.
__device__ __forceinline__ void finish_bruteforce_chunk(const float *__restrict__ desc, const PairDesc &pd, const int32_t *__restrict__ qrows /* LDS, nqc */,
                                                        int nqc, float4 (*s_q)[16] /* LDS [32][16] */, unsigned long long (*s_keys)[32][2] /* LDS [waves][32][2] */,
                                                        int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist)
{
    typedef unsigned long long u64;
    constexpr u64 kEmpty = ~0ull;
    auto key_of = [](float d, int t) { return d < FLT_MAX ? (((u64)__float_as_uint(d) << 32) | (u64)(uint32_t)t) : ~0ull; };   // FLT_MAX, +inf, NaN: never a neighbour
    auto insert2 = [](u64 &b0, u64 &b1, u64 k) {
        const u64 hi = k > b0 ? k : b0;
        b0 = k > b0 ? b0 : k;
        b1 = hi < b1 ? hi : b1;
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 *Q = reinterpret_cast<const float4 *>(desc + (size_t)pd.q_row0 * 64);
    const float4 *T = reinterpret_cast<const float4 *>(desc + (size_t)pd.t_row0 * 64);
    for (int e = tid; e < nqc * 16; e += kFinThreads) s_q[e >> 4][e & 15] = Q[(size_t)qrows[e >> 4] * 16 + (e & 15)];
    __syncthreads();
    u64 m0 = kEmpty, m1 = kEmpty;                    // lane k: query k's two best over this wave's rows
    for (int t0 = wave * 64; t0 < pd.nt; t0 += kFinThreads) {
        const int t = t0 + lane;
        float4 ta[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) ta[c] = t < pd.nt ? T[(size_t)t * 16 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < nqc; ++k) {
            float4 qa[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) qa[c] = s_q[k][c];
            const float d2 = l2sqr64_canonical_regs(qa, ta);
            u64 x0 = t < pd.nt ? key_of(sqrt_rn_f32(d2), t) : kEmpty, x1 = kEmpty;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const u64 y0 = __shfl_xor(x0, o), y1 = __shfl_xor(x1, o);
                insert2(x0, x1, y0);
                insert2(x0, x1, y1);
            }
            if (lane == k) { insert2(m0, m1, x0); insert2(m0, m1, x1); }
        }
    }
    if (lane < 32) { s_keys[wave][lane][0] = m0; s_keys[wave][lane][1] = m1; }
    __syncthreads();
    if (tid < nqc) {
        u64 x0 = kEmpty, x1 = kEmpty;
        for (int w = 0; w < kFinWaves; ++w) { insert2(x0, x1, s_keys[w][tid][0]); insert2(x0, x1, s_keys[w][tid][1]); }
        const size_t o = 2 * ((size_t)pd.out_off + qrows[tid]);
        const int i0 = (int)(uint32_t)x0, i1 = (int)(uint32_t)x1;
        st_coh_i(knn_idx + o, i0); st_coh_i(knn_idx + o + 1, i1);           // (read back by the ratio stage past the L2)
        st_coh_f(knn_dist + o, i0 >= 0 ? __uint_as_float((uint32_t)(x0 >> 32)) : FLT_MAX);
        st_coh_f(knn_dist + o + 1, i1 >= 0 ? __uint_as_float((uint32_t)(x1 >> 32)) : FLT_MAX);
    }
    __syncthreads();
}

// Stage (1) for ONE virtual set of up to QV = 7 survivors, by one wave, DENSE: eight lanes per query.  In round r lane l (query slot
// l >> 3, i = l & 7) evaluates row i of the query's group of rank r -- a kept group is eight rows: two runs of four, eight apart -- so
// a round is ONE batch of row transfers into the wave's 16-KiB landing zone (56 candidate rows, 16 lanes per 256-B row: every cache
// line touched once; round 0 also brings the seven query rows, which stay in the slots 56 .. 62) and every lane computes one
// distance, straight from LDS.
//  * a candidate is ONE 64-bit key, (bits of the distance) << 32 | train row: key order is the oracle's (distance, index) order; the
//    two best of a query's eight are reduced over its eight lanes by min / max exchanges and merged into (m0, m1), which all eight
//    lanes carry.  The bounds need d^2, not the canonical float D the distance is the root of: sqrtf is correctly rounded, so D lies
//    in d^2 (1 -+ 2^-22), and each use takes the side that keeps it conservative;
//  * round 0 needs the two smallest of the 2 K keys only (its group, the verdict's bound): the heads of the two sorted key lists; the
//    ranking by counting is made in the later rounds -- when a query is still undecided after its best group.
// Between the rounds the ratio test may already be DECIDED (the match entry points only; the reference emits queryIdx, trainIdx and d0
// of a survivor, never d1): with (m0, m1) the exact two best of the rows evaluated so far and lrest a lower bound on the D of every
// other row (the groups of the next ranks, everything outside the kept groups; rows of a group skipped as `cannot` are farther than
// the two rows that bound U anyway),
//   verdict 1, cannot pass:  D1 <= m1 and D0 >= min(m0, lrest) >= ratio^2 (1 + 2^-20) m1;
//   verdict 2, passes:       lrest > m0 (1 + 2^-20), so m0 IS the nearest row, and m0 (1 + 2^-18) < ratio^2 (1 + 2^-20) min(m1, lrest),
//                            so sqrtf(m0) < ratio sqrtf(D1) whatever the second nearest turns out to be -- it is not looked for
//                            (second index -3: "not determined, the test passes").
// On the metric's workload nearly every survivor is a true match whose best group alone decides it: one transfer round trip.  A
// query that ends neither certified nor decided goes on the pair's list for the threshold filter, with an upper bound of its
// second-best d^2 as the filter's threshold.
#ifdef ESFM_FIN_TRACE
__shared__ int s_fin_tr[4][4];         // per wave: ticks waiting for transfers, ticks behind the wait, rounds (flushed once per wave: atomics per round distort what they measure)
#endif
struct FinRerankArgs {
    const float4 *ent;                 // the pair's survivor entries
    int nsv, per;                      // ... their number; entries per virtual set (<= kFinQV)
    PairDesc pd; int p;
    __amdgpu_buffer_rsrc_t frsrc_t, frsrc_q;   // buffer descriptors of the train / query set's float rows (rows past a set read as zeros)
    double ratio2m;
    int32_t *knn_idx; float *knn_dist; float *knn_d2;
    int32_t *unc_cnt, *unc_list;       // the pair's list of queries for the threshold filter
    int32_t *counters, *audit_unc, *audit_rej; int flag_cap;
};
constexpr int kFinQV = 7;
// What the re-rank keeps of a query between two rounds: its survivor entry (two half-waves' K keys, row / |q|^2 / E1) and the exact
// two best rows so far.  64 bytes.
struct FinPending { float4 ka, kb, mi; unsigned long long m0, m1; };
#ifndef ESFM_FIN_PEND
#define ESFM_FIN_PEND 112
#endif
constexpr int kFinPend = ESFM_FIN_PEND;        // pending queries of a wave (16 virtual sets' worth): 7 KiB of LDS per wave

// Stage (1) for ONE WAVE's share of a pair's survivors: the virtual sets v0, v0 + vstride, ... < nvs.
// Round 5 -- COMPACTION BY ROUND.  Until then a virtual set of seven ran all of its rounds together and advanced at the pace of its
// slowest query: on real, clustered descriptors (M-SURF-4k-hard) a set took 3.24 rounds for queries that needed 1.8 on average, and
// every round costs the same whatever it holds.  Now round 0 runs over the wave's sets as before; a query that is neither decided
// nor out of groups afterwards is parked in the wave's LDS list (FinPending); the later rounds are run over that list, seven
// queries at a time, each round re-packing what is still undecided.  A query's arithmetic depends on nothing but its own state, so
// the results are bit-identical; on the metric's workload (one set per wave, 1.6 rounds) nothing changes.
template <bool SINGLE>
__device__ __forceinline__ void finish_rerank_wave(const FinRerankArgs &A, int v0, int vstride, int nvs, FinPending *pend)
{
    typedef unsigned long long u64;
    constexpr int K = ESFM_L2X1_KEEP, GRP = ESFM_L2X1_GRP, NG = 16 / GRP, QV = kFinQV;
    static_assert(K == 4 && GRP == 8, "an entry carries 2 x 4 keys; eight lanes evaluate the eight rows of a group");
    constexpr u64 kNone = ~0ull;
    constexpr float kBig = 3.0e38f;
    constexpr uint32_t kCodeMask = (1u << ESFM_L2X1_CODE_BITS) - 1u;
    constexpr double kTrunc = 1.0001 / (double)(1 << (23 - ESFM_L2X1_CODE_BITS));
    const int lane = threadIdx.x & 63;
    const int ri = (lane & 3) + 8 * ((lane >> 2) & 1);     // row of the group this lane evaluates (lanes 8 e .. 8 e + 7: entry e of the virtual set)
    const int nt = A.pd.nt, nq = A.pd.nq;
    const double ratio2m = A.ratio2m;
    const bool screen = ratio2m < 1.0e300;
    auto row0_of = [&](float key, int hh) {
        const int code = (int)(__float_as_uint(key) & kCodeMask);
        return key < 1.0e38f ? (code / NG) * 32 + (32 / NG) * (code % NG) + 4 * hh : -1;
    };
    auto kmin = [](u64 x, u64 y) { return x < y ? x : y; };
    auto kmax = [](u64 x, u64 y) { return x < y ? y : x; };
    // ---- the state of the query this lane works for (the same in its eight lanes)
    int nv = 0;                                                  // wave-uniform: queries of the current set
    bool qvalid = false;
    int qrow = 0;
    double qn = 0.0, e1 = 0.0, U = 0.0;
    float a_[4], b_[4], tau = kBig;
    float4 s_ka, s_kb, s_mi;                                     // (the entry as it came: what is parked when the query stays undecided)
    u64 m0 = kNone, m1 = kNone;                                  // the query's exact two best so far
    int verdict = 0;
    // the 2 K keys' order, once per parked query (round 6): position p of the ascending (key, index) order holds key number
    // (ord >> 3 p) & 7 (0 .. 3: half 0's keys, 4 .. 7: half 1's); kOrdValid marks a computed order.  It travels with the parked query in
    // the entry's unused fourth word.  (Until then every later round ranked the eight keys by counting TWICE -- once to decide whether
    // to park, once when the round ran: 8 x 7 compares each, a third of the later rounds' instructions on clustered descriptors.)
    constexpr int kOrdValid = 1 << 24;
    int ord = 0;
    auto take = [&](float4 ka, float4 kb4, float4 mi, bool valid, u64 b0, u64 b1) {
        s_ka = ka; s_kb = kb4; s_mi = mi;
        ord = __float_as_int(mi.w);                                // 0 in an entry as the pass wrote it
        qvalid = valid;
        qrow = qvalid ? __float_as_int(mi.x) : nq;               // nq: past the descriptor, zeros
        qn = (double)mi.y; e1 = (double)mi.z;
        a_[0] = qvalid ? ka.x : kBig; a_[1] = qvalid ? ka.y : kBig; a_[2] = qvalid ? ka.z : kBig; a_[3] = qvalid ? ka.w : kBig;
        b_[0] = qvalid ? kb4.x : kBig; b_[1] = qvalid ? kb4.y : kBig; b_[2] = qvalid ? kb4.z : kBig; b_[3] = qvalid ? kb4.w : kBig;
        tau = fminf(a_[3], b_[3]);
        // the two smallest keys (ties: half 0 first -- any fixed rule will do, the eight lanes only have to agree)
        const bool c0 = a_[0] <= b_[0];
        const float r1k = fminf(c0 ? a_[1] : a_[0], c0 ? b_[0] : b_[1]);
        U = (qn + (double)r1k + e1 + fabs((double)r1k) * kTrunc) * (1.0 + 1.0 / 1048576.0);
        m0 = b0; m1 = b1; verdict = 0;
    };
    // the group of rank r among the query's 2 K keys (by counting), and the smallest key behind it
    auto rank_group = [&](int r, float &key, int &row0, float &nkey) {
        key = kBig; nkey = kBig; row0 = -1;
#pragma unroll
        for (int x = 0; x < 2 * K; ++x) {
            const float kx = x < K ? a_[x & 3] : b_[x & 3];
            int rank = 0;
#pragma unroll
            for (int y = 0; y < 2 * K; ++y) {
                const float ky = y < K ? a_[y & 3] : b_[y & 3];
                if (y != x) rank += (ky < kx || (ky == kx && y < x)) ? 1 : 0;
            }
            if (rank == r) { key = kx; row0 = row0_of(kx, x < K ? 0 : 1); }
            if (rank == r + 1) nkey = kx;
        }
    };
    auto compute_ord = [&]() {
        int o = kOrdValid;
#pragma unroll
        for (int x = 0; x < 2 * K; ++x) {
            const float kx = x < K ? a_[x & 3] : b_[x & 3];
            int rank = 0;
#pragma unroll
            for (int y = 0; y < 2 * K; ++y) {
                const float ky = y < K ? a_[y & 3] : b_[y & 3];
                if (y != x) rank += (ky < kx || (ky == kx && y < x)) ? 1 : 0;
            }
            o |= x << (3 * rank);
        }
        ord = o;
    };
    auto key_number = [&](int x) {           // key number x of the query's eight (a chain of selects: a register array indexed by data would go to scratch)
        float k = a_[0];
        k = x == 1 ? a_[1] : k; k = x == 2 ? a_[2] : k; k = x == 3 ? a_[3] : k;
        k = x == 4 ? b_[0] : k; k = x == 5 ? b_[1] : k; k = x == 6 ? b_[2] : k; k = x == 7 ? b_[3] : k;
        return k;
    };
    // rank_group from the stored order: the same (key, row0, nkey)
    auto ordered_group = [&](int r, float &key, int &row0, float &nkey) {
        const int x = (ord >> (3 * r)) & 7;
        key = key_number(x);
        row0 = row0_of(key, x >> 2);
        nkey = r + 1 < 2 * K ? key_number((ord >> (3 * (r + 1))) & 7) : kBig;
    };
    // does a round on the group (key, row0) still have to look at rows?  (false once: false for every later rank -- the keys ascend)
    auto wanted = [&](float key, int row0) {
        const bool cannot = (qn + (double)key - e1 - fabs((double)key) * kTrunc) > U;   // false on NaN: re-rank
        return row0 >= 0 && qvalid && !cannot && verdict == 0;
    };
    // one round: the group (key, row0) of the query, nkey = the smallest key of the groups the later rounds would fetch
    auto do_round = [&](bool last, float key, int row0, float nkey) __attribute__((always_inline)) {
        const bool need = wanted(key, row0);
        if (__ballot(need) == 0ull) return;
        const int trow = row0 + ri;
        const int rsel = need ? trow : nt;                       // nt: past the descriptor, zeros
        // 16 lanes fetch one 256-B row INTO REGISTERS: load i brings the rows of the lanes 4 i .. 4 i + 3 (i < 2 nv: the candidate
        // rows), lane l of the wave its 16-byte piece l & 15 of the row of lane 4 i + (l >> 4); the query rows of the entries
        // 0 .. nv - 1 the same way (every 16-lane row holds a copy).  Distances across the 16 lanes of a row in the oracle's order
        // (l2sqr64_canonical_row16), handed to the lane that owns the candidate by one ds_bpermute per load.
        // (Until the end of round 4 the rows landed in a 16-KiB LDS zone per wave -- LDS-DMA -- and every lane summed its own row
        // from there: eight waves per CU, 56 queries in flight, and the zone idle for 60 % of a virtual set's 10 us.  Registers
        // hold the same 14 KiB per wave, but sixteen waves fit a CU.)
        int rs[14];
#pragma unroll
        for (int i = 0; i < 14; ++i) rs[i] = __builtin_amdgcn_ds_bpermute((4 * i + (lane >> 4)) * 4, rsel) * 256 + (lane & 15) * 16;
        u32x4 qv[QV], rowv[14];
#pragma unroll
        for (int e = 0; e < QV; ++e)
            if (e < nv) qv[e] = __builtin_amdgcn_raw_buffer_load_b128(A.frsrc_q, __builtin_amdgcn_readlane(qrow, 8 * e) * 256 + (lane & 15) * 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 14; ++i)
            if (i < 2 * nv) rowv[i] = __builtin_amdgcn_raw_buffer_load_b128(A.frsrc_t, rs[i], 0, 0);
#ifdef ESFM_FIN_TRACE
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) s_fin_tr[threadIdx.x >> 6][0] += (int)(rt1 - rt0);
#endif
        float da = 0.f;
        float got[14];
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            got[i] = 0.f;
            if (i < 2 * nv) {
#ifdef ESFM_FIN_DIST2X      // sizing build: the canonical-order distance evaluated TWICE (same results): what the kernel gains per evaluation removed
                float dr = l2sqr64_canonical_row16(qv[i >> 1], rowv[i]);
                {
                    u32x4 again = rowv[i];
                    asm volatile("" : "+v"(again));
                    const float dr2 = l2sqr64_canonical_row16(qv[i >> 1], again);
                    dr = dr2 != dr2 ? dr2 : dr;
                }
#else
                const float dr = l2sqr64_canonical_row16(qv[i >> 1], rowv[i]);
#endif
                got[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & 3) * 16 + 15) * 4, __float_as_int(dr)));
            }
            __builtin_amdgcn_sched_barrier(0);       // (one row group at a time: interleaved, the fourteen chains took 296 registers)
        }
#pragma unroll
        for (int i = 0; i < 14; ++i)
            if ((lane >> 2) == i) da = got[i];
        // this lane's candidate as a key (+inf, NaN, rows past the set: none); then the two best of the query's eight lanes
        const float dda = sqrt_rn_f32(da);
        u64 c0k = (need && trow < nt && dda < FLT_MAX) ? (((u64)__float_as_uint(dda) << 32) | (u64)(uint32_t)trow) : kNone, c1k = kNone;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const u64 p0 = __shfl_xor(c0k, o), p1 = __shfl_xor(c1k, o);
            const u64 lo = kmin(c0k, p0), hi = kmax(c0k, p0);
            c1k = kmin(hi, kmin(c1k, p1));
            c0k = lo;
        }
        {
            const u64 lo = kmin(m0, c0k), hi = kmax(m0, c0k);
            m1 = kmin(hi, kmin(m1, c1k));
            m0 = lo;
        }
#ifdef ESFM_FIN_TRACE
        if (lane == 0) { s_fin_tr[threadIdx.x >> 6][2] += 1; s_fin_tr[threadIdx.x >> 6][1] += (int)(__builtin_amdgcn_s_memrealtime() - rt1); }           // rounds, ticks after the wait
#endif
        if (screen && !last) {
            const float nk = fminf(nkey, tau);                   // the smallest key of anything not evaluated yet
            const double lrest = qn + (double)nk - e1 - fabs((double)nk) * kTrunc;
            const double d0 = (double)__uint_as_float((uint32_t)(m0 >> 32)), d1 = (double)__uint_as_float((uint32_t)(m1 >> 32));
            const double m0lo = d0 * d0 * (1.0 - 1.0 / 4194304.0), m0hi = d0 * d0 * (1.0 + 1.0 / 4194304.0);
            const double m1lo = d1 * d1 * (1.0 - 1.0 / 4194304.0), m1hi = d1 * d1 * (1.0 + 1.0 / 4194304.0);
            const bool two = m1 != kNone;
            const double r1 = ratio2m * m1hi;
            const bool fail = two && lrest >= r1 && m0lo >= r1;                                                    // (every compare false on NaN)
            const double dlo = lrest < m1lo ? lrest : m1lo;
            const bool pass = two && lrest > m0hi * (1.0 + 1.0 / 1048576.0) && m0hi * (1.0 + 1.0 / 262144.0) < ratio2m * dlo;
            if (verdict == 0 && qvalid) verdict = fail ? 1 : (pass ? 2 : 0);
        }
    };
    // the query is through: its record, and -- neither certified nor decided -- its place on the pair's list for the threshold filter
    auto finalize = [&]() {
        const size_t o = 2 * ((size_t)A.pd.out_off + qrow);
        const bool one = m0 != kNone, two = m1 != kNone;
        const double d0 = (double)__uint_as_float((uint32_t)(m0 >> 32)), d1 = (double)__uint_as_float((uint32_t)(m1 >> 32));
        const double m0lo = d0 * d0 * (1.0 - 1.0 / 4194304.0), m1hi = d1 * d1 * (1.0 + 1.0 / 4194304.0);
        bool certified = (tau >= 1.0e38f);       // the empty-slot sentinel: every train row is a candidate (a NaN tau compares false)
        double lmiss = 0.0;
        if (!certified && two) {
            const double eps = e1 + fabs((double)tau) * kTrunc;
            lmiss = qn + (double)tau - eps;                                                    // every row outside the kept groups has D >= lmiss
            certified = lmiss > m1hi * (1.0 + 1.0 / 2097152.0);     // false on NaN (e1 of non-finite rows)
        }
        // Not certified, but the ratio test is already decided: the true second-nearest has D1 <= m1, the true nearest
        // D0 >= min(m0, lmiss); if that is >= ratio^2 (1 + 2^-20) m1 the query cannot pass whatever the other rows are.
        const bool lost = verdict == 1 || (verdict == 0 && !certified && two && lmiss >= ratio2m * m1hi && m0lo >= ratio2m * m1hi);   // false on NaN
        if (lost) {
            st_coh_i(A.knn_idx + o, -2); st_coh_i(A.knn_idx + o + 1, -2);
            st_coh_f(A.knn_dist + o, FLT_MAX); st_coh_f(A.knn_dist + o + 1, FLT_MAX);
            if (A.audit_rej) {       // audit of the screen: what it dropped, on the global list
                const int slot = atomicAdd(&A.counters[0], 1);
                if (slot < A.flag_cap) { A.audit_rej[2 * slot] = A.p; A.audit_rej[2 * slot + 1] = qrow; }
            }
        } else {
            st_coh_i(A.knn_idx + o, one ? (int)(uint32_t)m0 : -1); st_coh_i(A.knn_idx + o + 1, verdict == 2 ? -3 : (two ? (int)(uint32_t)m1 : -1));
            st_coh_f(A.knn_dist + o, one ? __uint_as_float((uint32_t)(m0 >> 32)) : FLT_MAX);
            st_coh_f(A.knn_dist + o + 1, two ? __uint_as_float((uint32_t)(m1 >> 32)) : FLT_MAX);
        }
        if (!certified && !lost && verdict == 0) {
            atomicAdd(&A.counters[1], 1);
            float u2 = two ? (float)m1hi : FLT_MAX;              // the threshold filter's bound: an upper bound of the exact second best so far
            if (two && (double)u2 < m1hi) u2 = nextafterf(u2, FLT_MAX);
            st_coh_f(A.knn_d2 + A.pd.out_off + qrow, u2);
            if (A.audit_unc) {       // audit of THIS pass's certificate: its failures on the global list
                const int slot = atomicAdd(&A.counters[0], 1);
                if (slot < A.flag_cap) { A.audit_unc[2 * slot] = A.p; A.audit_unc[2 * slot + 1] = qrow; }
            }
            const int k = __hip_atomic_fetch_add(&A.unc_cnt[A.p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st_coh_i(A.unc_list + A.pd.out_off + k, qrow);
        }
    };
    // after round r: a query whose group of rank r + 1 is still wanted is parked at pend[*npend ...] (the leaders of the wave's
    // queries take consecutive slots), every other one is finalised
    auto park_or_finalize = [&](int r, int *npend) {
        float key = kBig, nkey; int row0 = -1;
        if (r + 1 < 2 * K) {
            if (!(ord & kOrdValid)) compute_ord();
            ordered_group(r + 1, key, row0, nkey);
        }
        const bool leader = qvalid && (lane & 7) == 0;
        const bool again = leader && r + 1 < 2 * K && wanted(key, row0);
        const unsigned long long bal = __ballot(again);
        if (again) {
            FinPending &dst = pend[*npend + __popcll(bal & ((1ull << lane) - 1ull))];
            dst.ka = s_ka; dst.kb = s_kb; dst.mi = make_float4(s_mi.x, s_mi.y, s_mi.z, __int_as_float(ord)); dst.m0 = m0; dst.m1 = m1;
        }
        if (leader && !again) finalize();
        *npend += __popcll(bal);
    };
    // the later rounds over the parked queries, seven at a time, re-packed in place after every round (a set is read into
    // registers before anything of it is written back, and what is written never passes what has been read)
    auto drain = [&](int npend) {
#pragma unroll 1
        for (int r = 1; r < 2 * K && npend > 0; ++r) {
            int nout = 0;
#pragma unroll 1
            for (int k = 0; k < npend; k += QV) {
                nv = min(QV, npend - k);
                const int e = lane >> 3;
                const FinPending src = pend[k + min(e, nv - 1)];
                __builtin_amdgcn_wave_barrier();
                take(src.ka, src.kb, src.mi, e < nv, src.m0, src.m1);
                float key, nkey; int row0;
                ordered_group(r, key, row0, nkey);                 // (a parked query carries its order)
                do_round(r + 1 == 2 * K, key, row0, nkey);
                park_or_finalize(r, &nout);
                __builtin_amdgcn_wave_barrier();
            }
            npend = nout;
        }
    };
    // ---- round 0 over the wave's virtual sets (the next set's entries are loaded while this one waits for its rows)
    const int eq = lane >> 3 < QV ? lane >> 3 : QV - 1;
    auto entry_of = [&](int v, int part) {
        const int e = v * A.per + eq;
        return (v < nvs && e < A.nsv) ? A.ent[3 * (size_t)e + part] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    int npend = 0;
    int v = v0;
    float4 e0 = entry_of(v, 0), e1v = entry_of(v, 1), e2 = entry_of(v, 2);
#pragma unroll 1
    for (; v < nvs; v += vstride) {
        const float4 n0 = entry_of(v + vstride, 0), n1 = entry_of(v + vstride, 1), n2 = entry_of(v + vstride, 2);
        nv = min(A.per, A.nsv - v * A.per);                      // wave-uniform: queries of this virtual set
        take(e0, e1v, e2, (lane >> 3) < nv, kNone, kNone);
        {
            const bool c0 = a_[0] <= b_[0];
            const float r0k = c0 ? a_[0] : b_[0];
            const float r1k = fminf(c0 ? a_[1] : a_[0], c0 ? b_[0] : b_[1]);
            do_round(false, r0k, row0_of(r0k, c0 ? 0 : 1), r1k);
        }
        if (SINGLE) {
            // the wave's only set (the metric's workload: a pair's survivors are one set per wave): its later rounds straight away, in
            // the registers the state is in -- parking and re-loading five queries costs more than their idle lanes do.  (SINGLE is a
            // template parameter, the kernel branches once per wave: compiled into one body with the parking form, this path -- round
            // 4's, instruction for instruction -- came out 3 us slower per launch.)
#pragma unroll 1
            for (int r = 1; r < 2 * K; ++r) {
                float key, nkey; int row0;
                rank_group(r, key, row0, nkey);
                if (__ballot(wanted(key, row0)) == 0ull) break;
                do_round(r + 1 == 2 * K, key, row0, nkey);
            }
            if (qvalid && (lane & 7) == 0) finalize();
        } else {
            park_or_finalize(0, &npend);
            if (npend + QV > kFinPend) { __builtin_amdgcn_wave_barrier(); drain(npend); npend = 0; }
        }
        e0 = n0; e1v = n1; e2 = n2;
    }
    if (!SINGLE) {
        __builtin_amdgcn_wave_barrier();
        drain(npend);
    }
}

// Stage (2)'s sweep over a wave's share [st0, st1) of the train set's 32-row steps, for the (up to) 64 queries of a sweep: the same
// bf16(-2 q) . bf16(t) product as the pass on the matrix cores, every score compared with its query's threshold, the rows that pass
// appended to the workgroup's hit list ((query slot) << 21 | train row).  Lane j of either half-wave owns the queries j and 32 + j.
// Branch-free loads through buffer descriptors (rows past nt read as zeros; their norms become kBig), the next step's eight loads in
// flight during this step's MFMAs.  (The first version guarded every load with `row < nt`: hipcc turned each into a branch and waited
// for every fragment before its MFMA -- 6.5 us per step.)  A function of its OWN, not inlined: inside l2_finish_kernel's body the
// register allocator -- 168 registers for three workgroups per CU, cut for the re-rank -- kept the operands of this loop in scratch
// memory and re-loaded them in front of every matrix instruction (6 us per step on M-SURF-4k-hard, round 5).
__device__ __noinline__ void finish_filter_sweep(const u32x4 *hi_rows /* the train set's bf16 images */, const float *tn /* its |row|^2 */, int st0, int st1, int nt_, int j,
                                                 int h, bool two, const bf16x8 (&bq_)[2][4], const float (&thr2_)[2], int *s_nhit, int *s_h)
{
    constexpr int HS = 8, CAP = kFinCap;
    constexpr float kBig = 3.0e38f;
    // (arguments of a non-inlined function arrive in vector registers: the descriptors are rebuilt from values the compiler can see
    // are wave-uniform, or every buffer load becomes a waterfall loop)
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)(uintptr_t)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<void *>((uintptr_t)(((unsigned long long)hi << 32) | lo));
    };
    const int nt = __builtin_amdgcn_readfirstlane(nt_);
    st0 = __builtin_amdgcn_readfirstlane(st0); st1 = __builtin_amdgcn_readfirstlane(st1);
    const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(hi_rows), 0, nt * (HS * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_n = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(tn), 0, nt * 4, 0x00020000);
    // the query operands and thresholds arrive by reference, i.e. in the caller's scratch memory: into registers ONCE (left as
    // references, every matrix instruction of every step re-loaded its operand with a flat load: 2.7 us per step)
    bf16x8 lbq[2][4]; float lthr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float tv = thr2_[u];
        asm volatile("" : "+v"(tv));
        lthr[u] = tv;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 v = __builtin_bit_cast(u32x4, bq_[u][ks]);
            asm volatile("" : "+v"(v));
            lbq[u][ks] = __builtin_bit_cast(bf16x8, v);
        }
    }
    const bf16x8 (&bq)[2][4] = lbq; const float (&thr2)[2] = lthr;
    auto load_step = [&](int st, u32x4 (&a)[4], u32x4 (&nv)[4]) {
        const int voff = (st * 32 + j) * (HS * 16) + h * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_t, voff + 32 * ks, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) nv[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_n, (st * 32 + 8 * g + 4 * h) * 4, 0, 0);
    };
    // one step: scores of the step's 32 rows against the sweep's queries, the rows under a query's threshold appended to the hit list
    auto step = [&](int st, const u32x4 (&ca)[4], const u32x4 (&cn)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;                       // (workgroup-uniform: a sweep of at most 32 queries runs one accumulator)
            floatx16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[4 * g + x] = (st * 32 + 8 * g + 4 * h + x) < nt ? __uint_as_float(cn[g][x]) : kBig;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ca[ks]), bq[u][ks], acc, 0, 0, 0);
            float m = kBig;
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fminf(m, acc[r]);       // (fminf drops NaN scores: never neighbours)
            if (__ballot(m <= thr2[u]) != 0ull) {
                uint32_t mask = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = st * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    mask |= (acc[r] <= thr2[u] && t < nt) ? (1u << r) : 0u;
                }
                if (mask) {                         // one atomic per lane with hits, then its slots in order
                    int k = atomicAdd(s_nhit, __popc(mask));
                    while (mask) {
                        const int r = __ffs(mask) - 1;
                        mask &= mask - 1;
                        if (k < CAP) s_h[k] = ((32 * u + j) << 21) | (st * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
                        ++k;
                    }
                }
            }
        }
    };
    // two operand sets: the loads of step s + 2 are issued right behind the use of step s.  (A ring of three was measured: the third
    // set pushes two of the query operands into scratch memory, re-loaded in front of their matrix instructions: slower.)
    u32x4 a0[4], n0[4], a1[4], n1[4];
    load_step(st0, a0, n0); load_step(st0 + 1, a1, n1);      // (steps past the set read zeros; past st1 they are not used)
    for (int st = st0; st < st1; st += 2) {
        step(st, a0, n0);
        if (st + 2 < st1) load_step(st + 2, a0, n0);
        if (st + 1 < st1) { step(st + 1, a1, n1); if (st + 3 < st1) load_step(st + 3, a1, n1); }
    }
}

constexpr size_t kFinTailLds = 8192 + 8192 + (size_t)kFinWaves * 32 * 2 * 8;        // the buffers of stages (2) - (4)
constexpr size_t kFinLdsBytes = kFinTailLds + (size_t)kFinWaves * kFinPend * sizeof(FinPending);   // + the re-rank's parked queries (its rows live in registers)

#ifndef ESFM_FIN_OCC
#define ESFM_FIN_OCC 3            // workgroups per CU the register budget is cut for: 3 = 168 registers, no spill in the re-rank (64.5 us per step;
                                  // 1: 342 registers, 113 us; 2: 76 us; 4: 128 registers, 42 spills in the re-rank, 80 - 87 us; the query rows parked in LDS: 66 / 84 us at 3 / 4)
#endif
__global__ __launch_bounds__(kFinThreads, ESFM_FIN_OCC) void l2_finish_kernel(const float *__restrict__ desc, const u32x4 *__restrict__ hi_t,
                                                                const u32x4 *__restrict__ hi_q, const float *__restrict__ norms,
                                                                const float *__restrict__ rho_t, const float *__restrict__ rho_q,
                                                                const PairDesc *__restrict__ pairs, const int32_t *__restrict__ pair_order, int n_pairs, int S,
                                                                const int32_t *__restrict__ surv_cnt, const float4 *__restrict__ surv_list,
                                                                int32_t *__restrict__ unc_cnt, int32_t *__restrict__ unc_list, float *__restrict__ knn_d2,
                                                                int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist,
                                                                int32_t *__restrict__ counters, int32_t *__restrict__ flagged, int flag_cap,
                                                                int32_t *__restrict__ done, int audit, int do_ratio, double ratio, double ratio2m,
                                                                int32_t *__restrict__ query_idx, int32_t *__restrict__ train_idx,
                                                                float *__restrict__ distance, int32_t *__restrict__ n_out)
{
    constexpr int CAP = kFinCap, HS = 8, NW = kFinWaves;
    constexpr float kBig = 3.0e38f;
    static_assert(kFinThreads == 256, "the later stages' buffers are laid out for four waves");
    extern __shared__ __attribute__((aligned(16))) char fin_smem[];         // kFinLdsBytes
    // stages (2) - (4) reuse the landing zones
    int *s_h = reinterpret_cast<int *>(fin_smem);                             // [CAP] hits of the sweep: (query slot in the sweep) << 21 | train row
    float4 (*s_q)[16] = reinterpret_cast<float4 (*)[16]>(fin_smem + 8192);   // [32][16]
    unsigned long long (*s_keys)[32][2] = reinterpret_cast<unsigned long long (*)[32][2]>(fin_smem + 8192 + 8192);   // [NW][32][2]
    __shared__ int s_nhit, s_last;
    __shared__ float s_red[2 * NW];
    __shared__ int s_qrows[64];
    __shared__ unsigned long long s_best[2][64];                              // stage 2: the running (distance, index) keys of a sweep's queries

    __shared__ int s_wave[NW], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    // Blocks are dispatched round-robin over the XCDs; xcd_remap gives every XCD a contiguous range of logical blocks, and the logical
    // order is pair-major over the pairs SORTED BY TRAIN SET: the S workgroups of a pair and the pairs of one train set run on one
    // XCD, whose L2 (4 MiB) then holds the one or two train sets their row fetches go to -- the re-rank is bound by those fetches
    // (51 k survivors x 9 rows x 256 B per step on the metric's workload).
#ifdef ESFM_FIN_TRACE
    const unsigned long long ft_in = __builtin_amdgcn_s_memrealtime();
#endif
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int sl = lb % S, p = pair_order[lb / S];
    const PairDesc pd = pairs[p];
    const int nq = pd.nq, nt = pd.nt;

    // ---- (1) re-rank of the survivors: virtual sets dealt out over the pair's S x NW waves
    {
        FinRerankArgs A;
        A.nsv = min(surv_cnt[p], nq);
        // the survivors dealt out EVENLY over the pair's S x NW waves while a wave's share fits one virtual set (a set costs its
        // latency chain whatever it holds: 25 sets of seven for 20 waves made five of them -- and their workgroups -- last twice as long)
        A.per = max(1, min(kFinQV, (A.nsv + S * NW - 1) / (S * NW)));
        A.ent = surv_list + 3 * (size_t)pd.out_off;
        A.pd = pd; A.p = p;
        A.frsrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(desc + (size_t)pd.t_row0 * 64), 0, nt * 256, 0x00020000);   // rows past the set read as zeros, no memory access
        A.frsrc_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(desc + (size_t)pd.q_row0 * 64), 0, nq * 256, 0x00020000);
        A.ratio2m = ratio2m;
        A.knn_idx = knn_idx; A.knn_dist = knn_dist; A.knn_d2 = knn_d2;
        A.unc_cnt = unc_cnt; A.unc_list = unc_list;
        A.counters = counters; A.audit_unc = audit == 3 ? flagged : nullptr; A.audit_rej = audit == 4 ? flagged : nullptr; A.flag_cap = flag_cap;
#ifdef ESFM_FIN_NOSTAGE1
        const int nvs = 0;                    // (timing experiments)
#else
        const int nvs = (A.nsv + A.per - 1) / A.per;
#endif
#ifdef ESFM_FIN_TRACE
        const unsigned long long ft0 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { s_fin_tr[wave][0] = 0; s_fin_tr[wave][1] = 0; s_fin_tr[wave][2] = 0; }
        const int nset = (nvs - (sl * NW + wave) + S * NW - 1) / (S * NW);
#endif
        if (sl * NW + wave + S * NW >= nvs)      // (wave-uniform) at most one virtual set for this wave
            finish_rerank_wave<true>(A, sl * NW + wave, S * NW, nvs, nullptr);
        else
            finish_rerank_wave<false>(A, sl * NW + wave, S * NW, nvs, reinterpret_cast<FinPending *>(fin_smem + kFinTailLds) + wave * kFinPend);
#ifdef ESFM_FIN_TRACE
        if (lane == 0) {      // (scratch/fin_trace.py: 10-ns ticks of stage 1 per wave, virtual sets, waves)
            atomicAdd(&counters[8], (int)(__builtin_amdgcn_s_memrealtime() - ft0)); atomicAdd(&counters[9], nset); atomicAdd(&counters[10], 1);
            atomicAdd(&counters[12], s_fin_tr[wave][0]); atomicAdd(&counters[13], s_fin_tr[wave][1]); atomicAdd(&counters[11], s_fin_tr[wave][2]);
            atomicAdd(&counters[14], (int)(ft0 - ft_in));
        }
#endif
    }
    // (Round 5, measured and not kept: every workgroup settling ITS OWN uncertified queries by exact brute force right here instead
    // of leaving them to the pair's last workgroup -- M-SURF-4k-hard: 16 695 such queries per step, the finish kernel 1.43 -> 1.71 ms:
    // 4096 exact distances per query are as many VALU instructions again as the whole re-rank; the threshold filter's MFMA pass is
    // what keeps the second pass cheap, and what it needed was a shorter serial tail, see stage (2).)
    // arrive; the last of the pair's S workgroups goes on alone.  Every wave waits for its own write-through stores to be
    // acknowledged before the barrier lets the arrival out.
#ifdef ESFM_FIN_TRACE
    const unsigned long long ft_arr = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (S > 1) {
        if (tid == 0) s_last = __hip_atomic_fetch_add(&done[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
        __syncthreads();
#ifdef ESFM_FIN_TRACE
        if (lane == 0) atomicAdd(&counters[15], (int)(__builtin_amdgcn_s_memrealtime() - ft_arr));
#endif
        if (!s_last) return;
        if (tid == 0) __hip_atomic_store(&done[p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (everybody has arrived: nobody touches it again in this launch)
    }
    if (audit == 3 || audit == 4) return;          // the first pass alone: its answers, its own lists

    // ---- (2), (3): the pair's uncertified queries, chunks of 32, the whole train set by this workgroup's four waves
#ifdef ESFM_FIN_NOSTAGE2
    const int cnt = 0;                        // (timing experiments: wrong results for uncertified queries)
#else
    const int cnt = min(ld_coh_i(unc_cnt + p), nq);
#endif
#ifndef ESFM_FIN_SMALL
#define ESFM_FIN_SMALL 8       // uncertified queries of a pair up to which the exact brute force beats the threshold filter's fixed ~100-us chain
#endif
    if (cnt > 0 && cnt <= ESFM_FIN_SMALL && audit != 1) {
        // a handful of queries: their exact 2-NN over the whole train set straight away (the threshold filter below is a chain of
        // nt / 128 dependent MFMA steps per chunk of 32 whatever the chunk holds: ~100 us for ONE query; this: a few us per query)
        if (tid < 32) s_qrows[tid] = tid < cnt ? ld_coh_i(unc_list + pd.out_off + tid) : 0;
        __syncthreads();
        if (tid < cnt) {
            const int sl2 = atomicAdd(&counters[0], 1);
            if (sl2 < flag_cap) { flagged[2 * sl2] = p; flagged[2 * sl2 + 1] = s_qrows[tid]; }
        }
        finish_bruteforce_chunk(desc, pd, s_qrows, cnt, s_q, s_keys, knn_idx, knn_dist);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else if (cnt > 0) {
        // Round 5: a sweep over the train set serves SIXTY-FOUR queries (two MFMA accumulators per operand load) and the hits are merged
        // by 64-bit LDS atomic minima of (distance, index) keys.  Until then a sweep took 32 queries and one thread per query walked the
        // whole hit list: on real descriptors (M-SURF-4k-hard: 55 uncertified queries per pair, 13 hits per query) the pair's last
        // workgroup spent 2.2 x (104 us of sweep + 68 us of hits) here while every other workgroup of the pair had left.
        typedef unsigned long long u64;
        const int nchunks = (cnt + 63) >> 6;
        const float *__restrict__ tn = norms + pd.t_row0;
        const float *__restrict__ tr = rho_t + pd.t_row0;
        {   // max |t|^2 and max rho_t over the train set
            float m = 0.f, r = 0.f;
            for (int t = tid; t < nt; t += kFinThreads) { m = fmaxf(m, tn[t]); r = fmaxf(r, tr[t]); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); r = fmaxf(r, __shfl_xor(r, o)); }
            if (lane == 0) { s_red[wave] = m; s_red[NW + wave] = r; }
        }
        __syncthreads();
        float tmax = 0.f, rmax = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) { tmax = fmaxf(tmax, s_red[w2]); rmax = fmaxf(rmax, s_red[NW + w2]); }
        const double sqrt_tmax = sqrt((double)tmax);
        // this wave's share of the train set, in steps of 32 rows
        const int nsteps = (nt + 31) / 32;
        const int st0 = (nsteps * wave) / NW, st1 = (nsteps * (wave + 1)) / NW;
        auto key_of = [](float d, int t) { return (t >= 0 && d < FLT_MAX) ? (((u64)__float_as_uint(d) << 32) | (u64)(uint32_t)t) : ~0ull; };   // FLT_MAX, +inf, NaN: never a neighbour
        for (int c = 0; c < nchunks; ++c) {
#ifdef ESFM_FIN_TRACE2
            const unsigned long long t2a = __builtin_amdgcn_s_memrealtime();
#endif
            if (tid == 0) s_nhit = 0;
            const int nqc = min(64, cnt - c * 64);
            // lane j of either half-wave owns the queries j and 32 + j of the sweep
            int qrow2[2]; float thr2[2]; bf16x8 bq[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int slot = c * 64 + 32 * u + j;
                const bool qok = slot < cnt;
                const int qrow = qok ? ld_coh_i(unc_list + pd.out_off + slot) : 0;
                qrow2[u] = qrow;
                // threshold on the score: s <= U - |q|^2 + E1, rounded up
                float thr = -kBig;
                if (qok) {
                    const double qn = (double)norms[pd.q_row0 + qrow], rq = (double)rho_q[pd.q_row0 + qrow];
                    const double e1 = l2x1_e1(qn, rq, sqrt_tmax, (double)tmax, (double)rmax);
                    const double uu = (double)ld_coh_f(knn_d2 + pd.out_off + qrow);
                    const double x = uu * (1.0 + 1.0 / 1048576.0) - qn + e1;
                    const double xs = x + fabs(x) * (1.0 / 1048576.0);
                    thr = xs < 3.0e38 ? (float)xs : kBig;                 // (NaN compares false: kBig, everything passes -> overflow -> brute force)
                    if (!(xs < 3.0e38)) thr = kBig;
                    if ((double)thr < xs) thr = nextafterf(thr, kBig);
                }
                thr2[u] = thr;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    u32x4 v = qok ? hi_q[((size_t)pd.q_row0 + qrow) * HS + 2 * ks + h] : u32x4{0u, 0u, 0u, 0u};
                    bq[u][ks] = __builtin_bit_cast(bf16x8, v);
                }
            }
            if (tid < 32) { s_qrows[tid] = qrow2[0]; s_qrows[32 + tid] = qrow2[1]; }
            // the running two best of every query of the sweep start from what the re-rank left
            if (tid < nqc) {
                const size_t o = 2 * ((size_t)pd.out_off + ld_coh_i(unc_list + pd.out_off + c * 64 + tid));
                s_best[0][tid] = key_of(ld_coh_f(knn_dist + o), ld_coh_i(knn_idx + o));
                s_best[1][tid] = key_of(ld_coh_f(knn_dist + o + 1), ld_coh_i(knn_idx + o + 1));
            }
            __syncthreads();
            finish_filter_sweep(hi_t + (size_t)pd.t_row0 * HS, tn, st0, st1, nt, j, h, nqc > 32, bq, thr2, &s_nhit, s_h);
            __syncthreads();
            const int nhit = s_nhit;
#ifdef ESFM_FIN_TRACE2
            const unsigned long long t2b = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) { atomicAdd(&counters[4], (int)(t2b - t2a)); atomicAdd(&counters[5], nhit); atomicAdd(&counters[6], 1); atomicMax(&counters[3], nhit); }
#endif
            if (nhit <= CAP) {
                // exact distances of the hits in the oracle's order (a thread per hit), merged as (distance, index) keys: the smallest
                // key of a query by a 64-bit LDS atomic minimum, then the smallest of the others (two different rows never share a key;
                // a hit that IS one of the two rows the re-rank left carries that row's key and changes nothing)
                u64 keyv[CAP / kFinThreads];
#pragma unroll
                for (int x = 0; x < CAP / kFinThreads; ++x) {
                    const int k = tid + kFinThreads * x;
                    keyv[x] = ~0ull;
                    if (k < nhit) {
                        const int hk = s_h[k];
                        const float4 *qp = reinterpret_cast<const float4 *>(desc + ((size_t)pd.q_row0 + s_qrows[hk >> 21]) * 64);
                        const float4 *tp = reinterpret_cast<const float4 *>(desc + ((size_t)pd.t_row0 + (hk & 0x1FFFFF)) * 64);
                        float4 qa[16], tb[16];
#pragma unroll
                        for (int e = 0; e < 16; ++e) { qa[e] = qp[e]; tb[e] = tp[e]; }
                        keyv[x] = key_of(sqrt_rn_f32(l2sqr64_canonical_regs(qa, tb)), hk & 0x1FFFFF);
                    }
                }
                u64 init0 = ~0ull;
                if (tid < nqc) init0 = s_best[0][tid];
                __syncthreads();
#pragma unroll
                for (int x = 0; x < CAP / kFinThreads; ++x) {
                    const int k = tid + kFinThreads * x;
                    if (k < nhit && keyv[x] != ~0ull) atomicMin(&s_best[0][s_h[k] >> 21], keyv[x]);
                }
                __syncthreads();
#pragma unroll
                for (int x = 0; x < CAP / kFinThreads; ++x) {
                    const int k = tid + kFinThreads * x;
                    if (k < nhit && keyv[x] != ~0ull && keyv[x] != s_best[0][s_h[k] >> 21]) atomicMin(&s_best[1][s_h[k] >> 21], keyv[x]);
                }
                if (tid < nqc && init0 != s_best[0][tid]) atomicMin(&s_best[1][tid], init0);      // (the re-rank's best, displaced by a hit)
                __syncthreads();
                if (tid < nqc) {
                    const size_t o = 2 * ((size_t)pd.out_off + s_qrows[tid]);
                    const u64 k0 = s_best[0][tid], k1 = s_best[1][tid];
                    st_coh_i(knn_idx + o, k0 != ~0ull ? (int)(uint32_t)k0 : -1); st_coh_i(knn_idx + o + 1, k1 != ~0ull ? (int)(uint32_t)k1 : -1);
                    st_coh_f(knn_dist + o, k0 != ~0ull ? __uint_as_float((uint32_t)(k0 >> 32)) : FLT_MAX);
                    st_coh_f(knn_dist + o + 1, k1 != ~0ull ? __uint_as_float((uint32_t)(k1 >> 32)) : FLT_MAX);
                }
                __syncthreads();
#ifdef ESFM_FIN_TRACE2
                if (tid == 0) atomicAdd(&counters[7], (int)(__builtin_amdgcn_s_memrealtime() - t2b));
#endif
            } else {
                // too many rows inside the error bound: exact brute force of the sweep's queries, 32 at a time
                if (tid < nqc) {
                    const int sl2 = atomicAdd(&counters[0], 1);
                    if (sl2 < flag_cap) { flagged[2 * sl2] = p; flagged[2 * sl2 + 1] = s_qrows[tid]; }
                }
                __syncthreads();
                if (audit != 1) {
                    finish_bruteforce_chunk(desc, pd, s_qrows, min(32, nqc), s_q, s_keys, knn_idx, knn_dist);
                    if (nqc > 32) finish_bruteforce_chunk(desc, pd, s_qrows + 32, nqc - 32, s_q, s_keys, knn_idx, knn_dist);
                }
                __syncthreads();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the write-through stores above, before the ratio stage reads them back)
        __syncthreads();
    }
#ifdef ESFM_FIN_NORATIO
    if (do_ratio && n_pairs < 0)          // (timing experiments)
#else
    if (do_ratio)
#endif
    {
        if (ratio2m < 1.0e300)            // the screen ran: only its survivors have records
            ratio_compact_pair_sparse<kFinThreads>(pd, surv_list + 3 * (size_t)pd.out_off, min(surv_cnt[p], nq), knn_idx, knn_dist, ratio, query_idx, train_idx,
                                                   distance, n_out + p, reinterpret_cast<uint32_t *>(fin_smem), s_wave, &s_base);
        else
            ratio_compact_pair<kFinThreads, 4096 / kFinThreads, true>(pd, knn_idx, knn_dist, ratio, query_idx, train_idx, distance, n_out + p, s_wave, &s_base);
    }
}

// ---------------------------------------------------------------------------------------------
// Exact brute-force 2-NN for listed queries (flagged != NULL: entries [0, counters[0])) or for
// every query of every pair (flagged == NULL: entries [0, total_queries)).  One workgroup per
// entry, threads stride over the train rows, lexicographic (distance, index) reduction.
template <bool VEC>
__global__ __launch_bounds__(256) void l2_exact_scan_kernel(const float *__restrict__ desc, int dim,
                                                            const PairDesc *__restrict__ pairs, int n_pairs,
                                                            const int32_t *__restrict__ flagged,
                                                            const int32_t *__restrict__ counters, long long total_queries,
                                                            int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist)
{
    __shared__ float s_d[2][256];
    __shared__ int s_i[2][256];
    const int tid = threadIdx.x;
    const long long n_entries = flagged ? (long long)counters[0] : total_queries;
    for (long long e = blockIdx.x; e < n_entries; e += gridDim.x) {
        int pi, qrow;
        if (flagged) { pi = flagged[2 * e]; qrow = flagged[2 * e + 1]; }
        else { pi = find_pair_by_query(pairs, n_pairs, e); qrow = (int)(e - pairs[pi].out_off); }
        const PairDesc pd = pairs[pi];
        const float *q = desc + ((size_t)pd.q_row0 + qrow) * dim;
        const float *T = desc + (size_t)pd.t_row0 * dim;
        Cand b0 = {FLT_MAX, -1, 0.f}, b1 = {FLT_MAX, -1, 0.f};
        for (int t = tid; t < pd.nt; t += 256) {
            const float d2 = l2sqr_canonical<VEC>(q, T + (size_t)t * dim, dim);
            best2_insert(b0, b1, sqrt_rn_f32(d2), t, d2);
        }
        s_d[0][tid] = b0.d; s_i[0][tid] = b0.i; s_d[1][tid] = b1.d; s_i[1][tid] = b1.i;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) {
                Cand a0 = {s_d[0][tid], s_i[0][tid], 0.f}, a1 = {s_d[1][tid], s_i[1][tid], 0.f};
                best2_insert(a0, a1, s_d[0][tid + w], s_i[0][tid + w], 0.f);
                best2_insert(a0, a1, s_d[1][tid + w], s_i[1][tid + w], 0.f);
                s_d[0][tid] = a0.d; s_i[0][tid] = a0.i; s_d[1][tid] = a1.d; s_i[1][tid] = a1.i;
            }
            __syncthreads();
        }
        if (tid == 0) {
            const size_t o = 2 * ((size_t)pd.out_off + qrow);
            knn_idx[o] = s_i[0][0]; knn_idx[o + 1] = s_i[1][0];
            knn_dist[o] = s_i[0][0] >= 0 ? s_d[0][0] : FLT_MAX;
            knn_dist[o + 1] = s_i[1][0] >= 0 ? s_d[1][0] : FLT_MAX;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// (Measured alternative, round 2: one workgroup per PAIR, its uncertified queries -- 2.2 on average -- sharing every train row a
// thread loads: fewer bytes, but eight candidate states and two train rows per thread spill, 0.35 ms against 0.105 ms.)
// The rescan of the queries the certificate rejects, 64-float rows: same result as l2_exact_scan_kernel, but latency-aware --
// the handful of flagged queries (0.06 % on M-SURF-4k) leaves the chip nearly empty, so a thread keeps its query row in
// registers and has the loads of two train rows in flight at a time, and the (distance, index) reduction runs on wave
// shuffles.  l2sqr64_canonical_regs is l2sqr_canonical on register operands: the same 8 chains, the same final order.
__global__ __launch_bounds__(256) void l2_rescan64_kernel(const float *__restrict__ desc, const PairDesc *__restrict__ pairs,
                                                          const int32_t *__restrict__ flagged, const int32_t *__restrict__ counters,
                                                          int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist)
{
    __shared__ float s_d[2][4];
    __shared__ int s_i[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_entries = counters[0];
    for (int e = blockIdx.x; e < n_entries; e += gridDim.x) {
        const int pi = flagged[2 * e], qrow = flagged[2 * e + 1];
        const PairDesc pd = pairs[pi];
        const float4 *qp = reinterpret_cast<const float4 *>(desc + ((size_t)pd.q_row0 + qrow) * 64);
        const float4 *T = reinterpret_cast<const float4 *>(desc + (size_t)pd.t_row0 * 64);
        float4 qv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) qv[c] = qp[c];
        Cand b0 = {FLT_MAX, -1, 0.f}, b1 = {FLT_MAX, -1, 0.f};
        for (int t = tid; t < pd.nt; t += 512) {
            const int t2 = t + 256;
            const bool two = t2 < pd.nt;
            float4 ta[16], tb[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) ta[c] = T[(size_t)t * 16 + c];
#pragma unroll
            for (int c = 0; c < 16; ++c) tb[c] = T[(size_t)(two ? t2 : t) * 16 + c];
            const float da = l2sqr64_canonical_regs(qv, ta), db = l2sqr64_canonical_regs(qv, tb);
            best2_insert(b0, b1, sqrt_rn_f32(da), t, da);
            if (two) best2_insert(b0, b1, sqrt_rn_f32(db), t2, db);
        }
        // (distance, index) is a total order: the merge order does not matter
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float d0 = __shfl_xor(b0.d, o), d1 = __shfl_xor(b1.d, o);
            const int i0 = __shfl_xor(b0.i, o), i1 = __shfl_xor(b1.i, o);
            best2_insert(b0, b1, d0, i0, 0.f);
            best2_insert(b0, b1, d1, i1, 0.f);
        }
        if (lane == 0) { s_d[0][wave] = b0.d; s_i[0][wave] = b0.i; s_d[1][wave] = b1.d; s_i[1][wave] = b1.i; }
        __syncthreads();
        if (tid == 0) {
            Cand a0 = {FLT_MAX, -1, 0.f}, a1 = {FLT_MAX, -1, 0.f};
            for (int w = 0; w < 4; ++w) { best2_insert(a0, a1, s_d[0][w], s_i[0][w], 0.f); best2_insert(a0, a1, s_d[1][w], s_i[1][w], 0.f); }
            const size_t o = 2 * ((size_t)pd.out_off + qrow);
            knn_idx[o] = a0.i; knn_idx[o + 1] = a1.i;
            knn_dist[o] = a0.i >= 0 ? a0.d : FLT_MAX;
            knn_dist[o + 1] = a1.i >= 0 ? a1.d : FLT_MAX;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// The re-scan of l2_knn_bf16_kernel's uncertified queries, pair by pair.  l2_rescan64_kernel above streams a whole train set per
// QUERY (662 MiB through the L2s for the 662 queries of M-SURF-4k, 295 GB for the 148 k of M-SURF-8k) with one row per lane: a
// load instruction touches 64 cache lines, and the L1 looks up one line per clock -- measured, a workgroup's pass over 1 MiB
// took ~50 us whatever else the chip was doing.  Here
//  * a workgroup takes up to `chunk` uncertified queries of ONE pair (the distance pass bins them per pair) and every train row
//    is compared with all of them; workgroup (p, c) of the chunks_per_pair workgroups of pair p takes the chunks c,
//    c + chunks_per_pair, ... of the pair's list, so one pair with thousands of uncertified queries (duplicated descriptors)
//    still spreads over the chip;
//  * train rows come in by LDS-DMA, 16 lanes per 256-B row (4 rows = 1 KiB per wave instruction, every line touched once), XOR
//    swizzled on the source side like the distance pass's tiles; a wave stages exactly the 64 rows its own lanes consume -- lane l
//    reads row l back with 16 conflict-free ds_read_b128 -- so no workgroup barrier is involved, and the next 64 rows are in
//    flight into the same LDS slice while the current ones (now in registers) are compared;
//  * the queries sit in LDS and are read as broadcasts (every lane the same address); through the scalar cache -- no vector
//    registers at all -- the four s_load_dwordx16 of a row came back one after the other into the same SGPRs, ~1 us per query
//    and group; a thread's two best keys per query live in LDS too (a private 16-B slot per query: the query loop is a real
//    loop, NQ x 4 registers indexed by it would go to scratch).
// Same arithmetic as l2_exact_scan_kernel (l2sqr_canonical's 8 chains and final order, sqrtf, (distance, index) order): the
// result is identical.
template <int NQ>
__global__ __launch_bounds__(256) void l2_rescan64_pairs_kernel(const float *__restrict__ desc, const PairDesc *__restrict__ pairs,
                                                                const int32_t *__restrict__ pair_cnt, const int32_t *__restrict__ pair_list,
                                                                int chunks_per_pair, int chunk /* <= NQ */, int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist)
{
    // (distance, train index) as one 64-bit key: distances are >= +0 and not NaN for finite descriptors, so the bit pattern of the
    // float orders like the float and key order is the (distance, index) order of best2_insert; the two smallest keys are kept
    // without a branch.  ~0 is the empty slot (index -1).
    typedef unsigned long long u64;
    constexpr u64 kEmpty = ~0ull;
    auto key_of = [](float d, int t) { return d < FLT_MAX ? (((u64)__float_as_uint(d) << 32) | (u64)(uint32_t)t) : ~0ull; };   // FLT_MAX, +inf, NaN: never a neighbour (oracle: `d < d1`)
    auto insert2 = [](u64 &b0, u64 &b1, u64 k) {
        const u64 hi = k > b0 ? k : b0;
        b0 = k > b0 ? b0 : k;
        b1 = hi < b1 ? hi : b1;
    };
    __shared__ float4 s_q[NQ][16];
    __shared__ int s_qrow[NQ];
    __shared__ u64 s_k[2][NQ][4];
    extern __shared__ __attribute__((aligned(16))) char smem_rescan[];
    float4 *s_rows = reinterpret_cast<float4 *>(smem_rescan);                                 // [4 waves][64 rows][16 slots]
    ulonglong2 *s_state = reinterpret_cast<ulonglong2 *>(smem_rescan + 4 * 64 * 256);         // [NQ][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = blockIdx.x / chunks_per_pair, c0 = blockIdx.x - p * chunks_per_pair;
    const PairDesc pd = pairs[p];
    const int cnt = min(pair_cnt[p], pd.nq);
    if (c0 * chunk >= cnt) return;
    const float *Q = desc + (size_t)pd.q_row0 * 64;
    const u32x4 trsrc = raw_buffer_rsrc(desc + (size_t)pd.t_row0 * 64, (uint32_t)pd.nt * 256u);   // rows past nt read as zeros
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lds_rows = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)s_rows) + (uint32_t)wave_s * (64 * 256);
    const float4 *my_row = s_rows + (size_t)(wave * 64 + lane) * 16;
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * i + (lane >> 4);                       // rows 16 apart share the swizzle
        voff[i] = row * 256 + (((lane & 15) ^ (row & 15)) * 16);
    }
    const int ngroups = (pd.nt + 255) / 256;                       // 256 train rows per step of the workgroup, 64 per wave
    auto dma_rows = [&](int g) {
        // the whole offset travels in the per-lane operand, which is what the descriptor's range check covers
        const int base = (g * 256 + wave_s * 64) * 256;            // byte offset of this wave's 64 rows
#pragma unroll
        for (int i = 0; i < 16; ++i) lds_dma_b128(lds_rows + (uint32_t)i * 1024u, voff[i & 3] + base + (i >> 2) * (16 * 256), trsrc, 0);
    };
    for (int c = c0; c * chunk < cnt; c += chunks_per_pair) {
        const int nqc = min(chunk, cnt - c * chunk);       // workgroup-uniform
        if (tid < nqc * 16) {
            const int k = tid >> 4, qrow = pair_list[pd.out_off + c * chunk + k];
            s_q[k][tid & 15] = reinterpret_cast<const float4 *>(Q)[(size_t)qrow * 16 + (tid & 15)];
            if ((tid & 15) == 0) s_qrow[k] = qrow;
        }
        for (int k = 0; k < nqc; ++k) s_state[k * 256 + tid] = make_ulonglong2(kEmpty, kEmpty);
        if (ngroups > 0) dma_rows(0);
        __syncthreads();
        for (int g = 0; g < ngroups; ++g) {
            const int t = g * 256 + wave * 64 + lane;
            float4 ta[16];
            lds_dma_wait();
#pragma unroll
            for (int j = 0; j < 16; ++j) ta[j] = my_row[j ^ (lane & 15)];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slice is in registers: the next rows may overwrite it
            if (g + 1 < ngroups) dma_rows(g + 1);
            for (int k = 0; k < nqc; ++k) {
                const float4 *qk = s_q[k];       // every lane the same address: LDS broadcast reads
                float2v acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};      // l2sqr64_canonical_regs, packed
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2v av[4] = {{ta[2 * j].x, ta[2 * j].y}, {ta[2 * j].z, ta[2 * j].w}, {ta[2 * j + 1].x, ta[2 * j + 1].y}, {ta[2 * j + 1].z, ta[2 * j + 1].w}};
                    const float4 q0 = qk[2 * j], q1 = qk[2 * j + 1];
                    const float2v qe[4] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}, {q1.z, q1.w}};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float2v d = qe[e] - av[e];
                        acc[e] = acc[e] + d * d;
                    }
                }
                const float2v s01 = acc[0] + acc[2], s23 = acc[1] + acc[3];
                const float da = __fadd_rn(__fadd_rn(__fadd_rn(s01.x, s01.y), s23.x), s23.y);
                ulonglong2 st = s_state[k * 256 + tid];
                insert2(st.x, st.y, t < pd.nt ? key_of(sqrt_rn_f32(da), t) : kEmpty);
                s_state[k * 256 + tid] = st;
            }
        }
        for (int k = 0; k < nqc; ++k) {
            const ulonglong2 st = s_state[k * 256 + tid];
            u64 x0 = st.x, x1 = st.y;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const u64 y0 = __shfl_xor(x0, o), y1 = __shfl_xor(x1, o);
                insert2(x0, x1, y0);
                insert2(x0, x1, y1);
            }
            if (lane == 0) { s_k[0][k][wave] = x0; s_k[1][k][wave] = x1; }
        }
        __syncthreads();
        if (tid < nqc) {
            u64 x0 = kEmpty, x1 = kEmpty;
            for (int w = 0; w < 4; ++w) { insert2(x0, x1, s_k[0][tid][w]); insert2(x0, x1, s_k[1][tid][w]); }
            const size_t o = 2 * ((size_t)pd.out_off + s_qrow[tid]);
            const int i0 = (int)(uint32_t)x0, i1 = (int)(uint32_t)x1;
            knn_idx[o] = i0; knn_idx[o + 1] = i1;
            knn_dist[o] = i0 >= 0 ? __uint_as_float((uint32_t)(x0 >> 32)) : FLT_MAX;
            knn_dist[o + 1] = i1 >= 0 ? __uint_as_float((uint32_t)(x1 >> 32)) : FLT_MAX;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Hamming 2-NN (ORB).  One thread per query row, descriptor words in VGPRs; the train row is
// wave-uniform, so it is fetched through the scalar cache (s_load) and XOR'd against the VGPRs.
// key = distance << 22 | train index: one u32 min orders by (distance, index) exactly.
template <int NW>
__global__ __launch_bounds__(256) void hamming_knn_kernel(const uint32_t *__restrict__ desc, const PairDesc *__restrict__ pairs,
                                                          int n_pairs, int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int pi = find_pair_by_block(pairs, n_pairs, lb);
    const PairDesc pd = pairs[pi];
    const int qrow = (lb - pd.blk_off) * 256 + threadIdx.x;
    const bool qvalid = qrow < pd.nq;
    uint32_t qw[NW];
    {
        const uint32_t *qp = desc + ((size_t)pd.q_row0 + (qvalid ? qrow : 0)) * NW;
#pragma unroll
        for (int w = 0; w < NW; ++w) qw[w] = qp[w];
    }
    const uint32_t *__restrict__ T = desc + (size_t)pd.t_row0 * NW;
    uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
    const int nt = pd.nt;
#pragma unroll 16
    for (int t = 0; t < nt; ++t) {
        const uint32_t *tp = T + (size_t)t * NW;  // wave-uniform address
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) d += __popc(qw[w] ^ tp[w]);
        const uint32_t key = (d << 22) | (uint32_t)t;
        const uint32_t hi = max(k0, key);
        k0 = min(k0, key);
        k1 = min(k1, hi);
    }
    if (qvalid) {
        const size_t o = 2 * ((size_t)pd.out_off + qrow);
        const bool h0 = nt >= 1, h1 = nt >= 2;
        knn_idx[o] = h0 ? (int)(k0 & 0x3FFFFFu) : -1;
        knn_idx[o + 1] = h1 ? (int)(k1 & 0x3FFFFFu) : -1;
        knn_dist[o] = h0 ? (float)(k0 >> 22) : FLT_MAX;
        knn_dist[o + 1] = h1 ? (float)(k1 >> 22) : FLT_MAX;
    }
}

// ---------------------------------------------------------------------------------------------
// Hamming 2-NN for 256-bit descriptors (ORB) on the i8 matrix cores.  Bits are stored as 0/1 bytes; the query operand
// enters the MFMA doubled (0/2) and each train's accumulator starts at 256 - popcount(t), so that
//   acc = 256 - popcount(t) + 2 popcount(t & q) = 256 + popcount(q) - hamming(q, t),
// an exact integer identity: `v_mfma_i32_32x32x32_i8` ranks 32 trains x 32 queries x 32 bits at a time (larger acc =
// closer), and the per-query constant is removed when the two winners are written.  The operand encoding is chosen for
// the matrix pipe's power draw, which is what sets its clock here: on the symmetric +-1 expansion (dot = 256 - 2 ham, half
// the bytes 0xFF) the same kernel is 18 % slower, and bare MFMA loops over this workload's 2.6 POP take 0.68 ms on 0/1
// x 0/1 operands, 0.73 ms on zeros x +-1 and 0.87 ms on +-1 x +-1 -- a "peak" measured on constant operands overstates
// what random descriptors reach, and 0/1 trains against +-1 queries gain nothing: both operands have to be sparse.
// A = train rows (so that a lane's 16 results belong to ONE query, column lane & 31, and 16 different trains), B = query
// columns held in registers for the whole kernel (2 sets of 32 queries per wave: 64 VGPRs), train tiles of 64 rows
// staged through LDS (LDS-DMA, 16-B slots XOR-swizzled with row & 15: conflict-free ds_read_b128) together with their 64
// start values, shared by the 4 waves.  K is contracted in whatever order the hardware pairs the 16 bytes a lane supplies -- A and B are loaded with
// the same lane->byte convention, and the sum does not depend on it.
// Top-2: running (best, second) pairs of keys acc << 21 | (2^21 - 1 - L), largest first, with L = 16 * (32-train group
// number) + accumulator register -- a wave-uniform scalar, so a result costs v_lshl_add + v_max_u32 + v_med3_u32.  Within
// a lane L grows with the train index (row(r) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) is monotonic in r), so key order =
// (distance ascending, train index ascending); four independent pairs per query set give the VALU chain some slack.
// The train index is rebuilt from L at the end, where the slots and the two lane halves are merged.
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

constexpr int kHmTT = 64;          // trains per LDS tile
constexpr int kHmQB = 256;         // queries per workgroup (4 waves x 2 sets x 32)
constexpr uint32_t kHmLMask = 0x1FFFFFu;

// bits -> 0/1 bytes, one 32-bit word (32 output bytes) per thread; the 8 threads of a row also leave 256 - popcount(row)
__global__ __launch_bounds__(256) void hamming_expand_kernel(const uint32_t *__restrict__ desc, long long n_words, uint32_t *__restrict__ out,
                                                             int32_t *__restrict__ start)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const uint32_t w = i < n_words ? desc[i] : 0u;
    int pop = __popc(w);
    pop += __shfl_xor(pop, 1);
    pop += __shfl_xor(pop, 2);
    pop += __shfl_xor(pop, 4);
    if (i >= n_words) return;
    if ((i & 7) == 0) start[i >> 3] = 256 - pop;
    uint32_t o[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const uint32_t x = (w >> (4 * g)) & 0xFu;
        o[g] = (x & 1u) | ((x & 2u) << 7) | ((x & 4u) << 14) | ((x & 8u) << 21);   // one 0/1 byte per bit
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out + i * 8);
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// keeps the two largest keys seen, m1 >= m2
__device__ __forceinline__ void key_insert_max(uint32_t &m1, uint32_t &m2, uint32_t key)
{
    uint32_t med;   // second largest of (m1 >= m2, key); operands are VALU results, no MFMA hazard to pad
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(med) : "v"(m1), "v"(m2), "v"(key));
    m1 = max(m1, key);
    m2 = med;
}

__global__ __launch_bounds__(256, 2) void hamming_knn_mfma_kernel(const unsigned char *__restrict__ ex, const int32_t *__restrict__ start,
                                                               const uint32_t *__restrict__ packed, const PairDesc *__restrict__ pairs, int n_pairs,
                                                               int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][kHmTT * 256];   // 256-B rows, 16-B slots XOR-swizzled with row & 15
    __shared__ __attribute__((aligned(16))) int32_t lds_start[2][kHmTT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int pi = find_pair_by_block(pairs, n_pairs, lb);
    const PairDesc pd = pairs[pi];
    const int nq = pd.nq, nt = pd.nt;
    const unsigned char *__restrict__ Q = ex + (size_t)pd.q_row0 * 256;
    const unsigned char *__restrict__ T = ex + (size_t)pd.t_row0 * 256;
    const int32_t *__restrict__ TS = start + pd.t_row0;
    const int qbase = (lb - pd.blk_off) * kHmQB + wave * 64;

    // B operand: the query rows doubled (0/2 bytes), 8 K-chunks of 32 bytes, this lane's 16
    i32x4 bq[2][8];
    int qpop[2];   // set bits of this lane's query (both lane halves hold the same query)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int qrow = qbase + 32 * s + j;
        const bool ok = qrow < nq;
        const i32x4 *qp = reinterpret_cast<const i32x4 *>(Q + (size_t)(ok ? qrow : 0) * 256 + h * 16);
        int pop = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            i32x4 v = qp[2 * c];
            if (!ok) v = i32x4{0, 0, 0, 0};
            pop += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
            bq[s][c] = i32x4{v.x << 1, v.y << 1, v.z << 1, v.w << 1};
        }
        qpop[s] = pop + __shfl_xor(pop, 32);
    }
    // Two-level top-2, as in l2_knn_bf16_kernel: a lane's 16 results of a 32-train step are four groups of four consecutive
    // train rows (accumulator registers 4g .. 4g+3 = rows 8g + 4h + 0..3); the hot loop keeps the two best GROUPS per lane
    // (key = group maximum << 21 | 2^21 - 1 - (4 step + g): two v_max3, one v_lshl_add, max + med3 = 5 VALU per 4 results instead of
    // 12), and the tail counts the bits of the kept groups' rows exactly.  No certificate is involved: the scores are exact
    // integers, the two nearest rows lie in the two groups with the best maxima of the lane half that holds them (a group that
    // precedes the second nearest row's group in key order contains a row that precedes that row in (distance, index) order, and
    // there is only one such row), and ties between groups go to the lower train index like ties between rows.
    uint32_t m1[2][2], m2[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 2; ++r) { m1[s][r] = 0u; m2[s][r] = 0u; }

    const int n_tiles = (nt + kHmTT - 1) / kHmTT;
    const int n_full = nt / kHmTT;       // tiles with all 64 rows inside the set: the software-pipelined loop
    // Staging is LDS-DMA (buffer_load_dwordx4 ... lds, 4 rows = 1 KiB per wave instruction) with the swizzle applied on the
    // source side, issued from inline asm so that hipcc does not order the tile's LDS reads behind the transfer, and waited for
    // explicitly in front of the barrier -- the scheme of l2_knn_bf16_kernel.  Start values go through a register, loaded
    // before the tile's DMA and stored at the end of the iteration.
    const u32x4 trsrc = raw_buffer_rsrc(T, (uint32_t)nt * 256u);
    const uint32_t lds_addr = (uint32_t)(uintptr_t)&lds[0][0];
    const int wrow0 = __builtin_amdgcn_readfirstlane(wave * 16);             // this wave moves rows [wrow0, wrow0 + 16) of a tile
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wrow0 + 4 * i + (lane >> 4);
        voff[i] = row * 256 + (((lane & 15) ^ (row & 15)) * 16);
    }
    auto dma_tile = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t dst = lds_addr + (uint32_t)(buf * kHmTT * 256 + (wrow0 + 4 * i) * 256);
            const int soff = tile * kHmTT * 256;
            lds_dma_b128(dst, voff[i], trsrc, soff);
        }
    };
    auto start_load = [&](int tile) { return (tid < kHmTT && tile * kHmTT + tid < nt) ? TS[tile * kHmTT + tid] : 0; };
    auto start_store = [&](int buf, int32_t sv) { if (tid < kHmTT) lds_start[buf][tid] = sv; };
    if (n_tiles > 0) {
        const int32_t sv = start_load(0);
        start_store(0, sv);
        dma_tile(0, 0);
    }
    lds_dma_wait();
    __syncthreads();

    // the accumulator start values of a 32-train step, in the C/D register order: rows 8 g + 4 h + (0..3), g = 0..3
    auto load_start = [&](int buf, int sub) {
        const i32x4 *sp = reinterpret_cast<const i32x4 *>(&lds_start[buf][sub * 32 + 4 * h]);
        const i32x4 g0 = sp[0], g1 = sp[2], g2 = sp[4], g3 = sp[6];
        return i32x16{g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w};
    };
    // group g of a step's results p (set s) into the lane's two best groups; pK = 2^21 - 1 - 4 (step number of p)
    auto group_insert = [&](int s, int g, const i32x16 &p, uint32_t pK) {
        const int gm = max(max(p[4 * g], p[4 * g + 1]), max(p[4 * g + 2], p[4 * g + 3]));
        key_insert_max(m1[s][g & 1], m2[s][g & 1], ((uint32_t)gm << 21) + (pK - g));
    };
    // One 32-train step: 16 MFMAs into (c0, c1), with the fold of the PREVIOUS step's results (p0, p1) issued in their shadow --
    // one group insert (5 VALU) behind every second MFMA of a set -- so the matrix pipe and the VALU run concurrently.
    // arow = the lane's train row in LDS; its K-chunk c is the 16-B slot 2 c + h, stored at slot ^ (row & 15) = ^ (j & 15)
    auto step = [&](const unsigned char *arow, const i32x16 &c_init, i32x16 &c0, i32x16 &c1, const i32x16 &p0, const i32x16 &p1, uint32_t pK) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const i32x4 a = *reinterpret_cast<const i32x4 *>(arow + (((2 * c + h) ^ (j & 15)) * 16));
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[0][c], c == 0 ? c_init : c0, 0, 0, 0);
            if (c & 1) group_insert(0, c >> 1, p0, pK);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[1][c], c == 0 ? c_init : c1, 0, 0, 0);
            if (c & 1) group_insert(1, c >> 1, p1, pK);
        }
    };
    // start-up placeholders: acc 0 with pK = 15 gives keys 12..15, below every real key (real 4 step + g < 2^21 - 16)
    i32x16 pa0, pa1, pb0, pb1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { pb0[r] = 0; pb1[r] = 0; }
    uint32_t pbK = 15u;
    for (int tile = 0; tile < n_full; ++tile) {
        const int buf = tile & 1;
        const bool more = tile + 1 < n_tiles;
        int32_t nxt_start = 0;
        if (more) {
            nxt_start = start_load(tile + 1);
            dma_tile(tile + 1, buf ^ 1);                                       // lands under this tile's MFMAs
        }
        const unsigned char *arow = &lds[buf][j * 256];
        step(arow, load_start(buf, 0), pa0, pa1, pb0, pb1, pbK);                                        // sub 0, folding the previous tile's sub 1
        step(arow + 32 * 256, load_start(buf, 1), pb0, pb1, pa0, pa1, kHmLMask - (uint32_t)(tile * 2) * 4u);   // sub 1, folding sub 0
        pbK = kHmLMask - (uint32_t)(tile * 2 + 1) * 4u;
        __builtin_amdgcn_sched_barrier(0);
        if (more) start_store(buf ^ 1, nxt_start);
        lds_dma_wait();                                                        // the DMA issued above has landed
        __syncthreads();
    }
    // drain the pipeline
#pragma unroll
    for (int g = 0; g < 4; ++g) { group_insert(0, g, pb0, pbK); group_insert(1, g, pb1, pbK); }
    // the partial tile at the end of the set; rows past it are zero-filled with start value 0: their score 0 is the worst there
    // is, and the tail skips them by index
    if (n_full < n_tiles) {
        const int tile = n_full, buf = tile & 1;
#pragma unroll 1
        for (int sub = 0; sub < 2; ++sub) {
            const i32x16 c_init = load_start(buf, sub);
            i32x16 acc0 = c_init, acc1 = c_init;
            const unsigned char *arow = &lds[buf][(sub * 32 + j) * 256];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const i32x4 a = *reinterpret_cast<const i32x4 *>(arow + (((2 * c + h) ^ (j & 15)) * 16));
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[0][c], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[1][c], acc1, 0, 0, 0);
            }
            const uint32_t K0 = kHmLMask - (uint32_t)(tile * 2 + sub) * 4u;
#pragma unroll
            for (int g = 0; g < 4; ++g) { group_insert(0, g, acc0, K0); group_insert(1, g, acc1, K0); }
        }
    }
    // ---- tail: the kept groups' rows counted exactly on the packed descriptors (32 B per row) ----
    // The two nearest rows of a query lie in the two best groups of ALL its groups, so the four kept ones (two per lane half) are
    // first merged -- keys rebuilt with the group's first train row in the position field, which orders groups of different lane
    // halves like their rows -- and each lane of the pair counts ONE group: 4 rows, 8 loads.
    // row key = distance << 21 | train index: the smallest two are the (distance, index)-first two.
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    auto key_insert_min = [](uint32_t &k1, uint32_t &k2, uint32_t key) {
        const uint32_t hi = max(k1, key);
        k1 = min(k1, key);
        k2 = min(k2, hi);
    };
    const u32x4 *P = reinterpret_cast<const u32x4 *>(packed);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int qrow = qbase + 32 * s + j;
        const bool qvalid = qrow < nq;
        uint32_t g1 = 0u, g2 = 0u;
#pragma unroll
        for (int r = 0; r < 2; ++r) { key_insert_max(g1, g2, m1[s][r]); key_insert_max(g1, g2, m2[s][r]); }
        // position field: 2^21 - 1 - (first row / 4) = 2^21 - 1 - (8 step + 2 g + h); placeholders (below 16) become 0
        auto global_key = [&](uint32_t k) {
            const uint32_t L = kHmLMask - (k & kHmLMask);
            return k < 16u ? 0u : ((k & ~kHmLMask) | (kHmLMask - (2u * L + (uint32_t)h)));
        };
        g1 = global_key(g1); g2 = global_key(g2);
        const uint32_t p1 = __shfl_xor(g1, 32), p2 = __shfl_xor(g2, 32);
        key_insert_max(g1, g2, p1);
        key_insert_max(g1, g2, p2);                                        // both lanes of the pair now hold the query's two best groups
        const uint32_t mine = h == 0 ? g1 : g2;
        const bool live = mine != 0u && qvalid;
        const int row0 = (int)(kHmLMask - (mine & kHmLMask)) * 4;
        const u32x4 *qp = P + ((size_t)pd.q_row0 + (qvalid ? qrow : 0)) * 2;
        const u32x4 q0 = qp[0], q1 = qp[1];
        u32x4 t0[4], t1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = min(row0 + u, max(nt - 1, 0));
            const u32x4 *tp = P + ((size_t)pd.t_row0 + (live ? t : 0)) * 2;
            t0[u] = tp[0]; t1[u] = tp[1];
        }
        uint32_t k1 = kNone, k2 = kNone;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = row0 + u;
            const int dist = __popc(q0.x ^ t0[u].x) + __popc(q0.y ^ t0[u].y) + __popc(q0.z ^ t0[u].z) + __popc(q0.w ^ t0[u].w) +
                             __popc(q1.x ^ t1[u].x) + __popc(q1.y ^ t1[u].y) + __popc(q1.z ^ t1[u].z) + __popc(q1.w ^ t1[u].w);
            key_insert_min(k1, k2, (live && t < nt) ? (((uint32_t)dist << 21) | (uint32_t)t) : kNone);
        }
        const uint32_t o1 = __shfl_xor(k1, 32), o2 = __shfl_xor(k2, 32);
        key_insert_min(k1, k2, o1);
        key_insert_min(k1, k2, o2);
        if (h == 0 && qvalid) {
            const size_t o = 2 * ((size_t)pd.out_off + qrow);
            const bool h0 = k1 != kNone, h1 = k2 != kNone;
            knn_idx[o] = h0 ? (int)(k1 & kHmLMask) : -1;
            knn_idx[o + 1] = h1 ? (int)(k2 & kHmLMask) : -1;
            knn_dist[o] = h0 ? (float)(k1 >> 21) : FLT_MAX;
            knn_dist[o + 1] = h1 ? (float)(k2 >> 21) : FLT_MAX;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 256-bit Hamming on the FP4 matrix cores (round 4).  hamming(q, t) = pop(q) + pop(t) - 2 q.t, and q.t over 0/1 bits is a dot product
// of 256 NIBBLES: t's bits as e2m1 1.0 (0x2), q's as -2.0 (0xC), accumulated in f32 on top of a start value pop(t) + 512 --
// exact small integers, positive, with 14 zero bits at the low end of the mantissa.  A row of 256 nibbles is 128 B = four K-steps of
// v_mfma_f32_32x32x64_f8f6f4 (cbsz = blgp = 4: FP4 x FP4, 16 B per lane and K-step): byte for byte the shapes of the one-product L2
// pass, so the whole main loop -- LDS-DMA ring of two 256-row tiles, four query sets per wave, fold groups of eight with the
// position in the low mantissa bits -- is that pass's generator with another instruction (hmx1_segment_gfx950.inc).  The FP4
// instruction moves 64 K per 8 passes where v_mfma_i32_32x32x32_i8 moves 32 (measured 7.7 against 4.2 Pop/s,
// scratch/ubench/mfma_fp4.hip, which also checks the products exact), at half the operand bytes of the byte-per-bit form.
// No certificate: the scores are exact, a group key IS the group's smallest score.  With code order = row order inside a lane half,
// the nearest row sits in the half's smallest key's group and the second nearest in one of its two smallest (a group in front of it
// would hold a row in front of it in (distance, index) order, and there is only one such row), and a group whose score exceeds the
// second smallest score of all eight keys holds neither.  The tail counts the bits of those groups' rows exactly on the packed
// descriptors, in (distance, index) order.  Ratio screen as in the L2 pass, exact here: d0 = score(k0) - 512 + pop(q) is the nearest
// distance, the second smallest key bounds the second nearest from above, and (double) d0 >= ratio (double) U1 rejects (marker -2).
typedef int i32x4h __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hamming_expand_fp4_kernel(const uint32_t *__restrict__ desc, long long n_words, u32x4 *__restrict__ img_t,
                                                                 u32x4 *__restrict__ img_q, float *__restrict__ start)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const uint32_t w = i < n_words ? desc[i] : 0u;
    int pop = __popc(w);
    pop += __shfl_xor(pop, 1);
    pop += __shfl_xor(pop, 2);
    pop += __shfl_xor(pop, 4);
    if (i >= n_words) return;
    if ((i & 7) == 0) start[i >> 3] = (float)(pop + 512);
    u32x4 t, q;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t b = (w >> (8 * d)) & 0xFFu, x = 0u;
#pragma unroll
        for (int n = 0; n < 8; ++n) x |= ((b >> n) & 1u) << (4 * n + 1);      // nibble n = bit 8 d + n as 0x2 (e2m1 1.0)
        t[d] = x; q[d] = x * 6u;                                              // 0x2 -> 0xC (-2.0): no carries between nibbles
    }
    img_t[i] = t; img_q[i] = q;
}

#ifdef ESFM_HMX1_TRACE
// timing-only build (scratch/build_variant.sh NAME -DESFM_HMX1_TRACE): per-wave stage times of hamming_fp4_kernel in s_memrealtime ticks
// (10 ns) and the shader clock inside the main loop (s_memtime); scratch/hmx1_trace.py reads them through esfm_debug_hmx1_trace
__device__ int g_hmx1_trace[8];
extern "C" int esfm_debug_hmx1_trace(int *out, int reset)
{
    if (reset) { int z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_hmx1_trace), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hmx1_trace), sizeof(g_hmx1_trace));
}
#endif
__global__ __launch_bounds__(256, 2) void hamming_fp4_kernel(const uint32_t *__restrict__ packed, const u32x4 *__restrict__ img_t,
                                                             const u32x4 *__restrict__ img_q, const float *__restrict__ start,
                                                             const PairDesc *__restrict__ pairs, const int32_t *__restrict__ blk_pair, int n_blocks,
                                                             int32_t *__restrict__ knn_idx, float *__restrict__ knn_dist, double ratio,
                                                             int32_t *__restrict__ done, int n_pairs, int32_t *__restrict__ query_idx,
                                                             int32_t *__restrict__ train_idx, float *__restrict__ distance, int32_t *__restrict__ n_out)
{
    // (a pair without queries has no block: nobody would write its count)
    if (done && blockIdx.x == 0) for (int p = threadIdx.x; p < n_pairs; p += 256) if (pairs[p].nq == 0) n_out[p] = 0;
#ifdef ESFM_HMX1_TRACE
    const uint64_t tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int TT = ESFM_HMX1_TT, NS = ESFM_HMX1_SETS, K = ESFM_HMX1_KEEP, RING = ESFM_HMX1_RING, GRP = ESFM_HMX1_GRP, NG = 16 / GRP;
    constexpr int QB = 128 * NS, HS = 8;
    constexpr int TILE_BYTES = TT * HS * 16;
    static_assert(NS == 4 && (GRP == 8 || GRP == 16) && RING * TT == 512 && K >= 2, "written for the L2 one-product pass's shapes");
    constexpr uint32_t kCodeMask = (1u << ESFM_HMX1_CODE_BITS) - 1u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *lds_tile = reinterpret_cast<u32x4 *>(smem);
    float *lds_norm = reinterpret_cast<float *>(smem + RING * TILE_BYTES);
    int lane = threadIdx.x & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int tid = threadIdx.x, j = lane & 31, h = lane >> 5;
    const uint32_t lds_tile_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_tile);
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int pi = blk_pair[lb];
    const PairDesc pd = pairs[pi];
    const int nq = __builtin_amdgcn_readfirstlane(pd.nq), nt = __builtin_amdgcn_readfirstlane(pd.nt);
    const int q_row0 = __builtin_amdgcn_readfirstlane(pd.q_row0), t_row0 = __builtin_amdgcn_readfirstlane(pd.t_row0);
    const int qblk = lb - __builtin_amdgcn_readfirstlane(pd.blk_off2);
    const int ntiles = (nt + TT - 1) / TT;
    const float *__restrict__ tn = start + t_row0;
    const u32x4 trsrc = raw_buffer_rsrc(img_t + (size_t)t_row0 * HS, (uint32_t)nt * (HS * 16));
    const u32x4 nrsrc = raw_buffer_rsrc(tn, (uint32_t)nt * 4u);
    if (ntiles * TT != nt || ntiles < RING) {
        for (int i = tid; i < RING * TT * HS; i += 256) lds_tile[i] = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < RING; ++b) {
#pragma unroll
        for (int i = 0; i < TT / 32; ++i) {
            const int row = wave_s * (TT / 4) + 8 * i + (lane >> 3);
            const int voff = row * (HS * 16) + (((lane & 7) ^ ((row >> 1) & 7)) * 16);
            lds_dma_b128(lds_tile_addr + (uint32_t)(b * TILE_BYTES + (wave_s * (TT / 4) + 8 * i) * (HS * 16)), voff, trsrc, b * TILE_BYTES);
        }
    }
    u32x4 bq[NS][4];
    {
        const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(img_q + (size_t)q_row0 * HS), 0, nq * (HS * 16), 0x00020000);
        const int qbase0 = qblk * QB + wave_s * 32 * NS;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int voff = (qbase0 + 32 * s + j) * (HS * 16) + h * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bq[s][ks] = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, voff + 32 * ks, 0, 0);
        }
    }
    {
        float big;
        asm volatile("s_mov_b32 %0, 0x7f61b1e6" : "=s"(big));
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int t = tid + 256 * u; lds_norm[t] = t < nt ? tn[t] : big; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef ESFM_HMX1_TRACE
    const uint64_t tr1 = __builtin_amdgcn_s_memrealtime(), clk1 = __builtin_amdgcn_s_memtime();
#endif
    if (ntiles > 0) {
        asm volatile(ESFM_HMX1_SEGMENT_ASM
                     :
                     : "v"(bq[0][0]), "v"(bq[0][1]), "v"(bq[0][2]), "v"(bq[0][3]), "v"(bq[1][0]), "v"(bq[1][1]), "v"(bq[1][2]), "v"(bq[1][3]),
                       "v"(bq[2][0]), "v"(bq[2][1]), "v"(bq[2][2]), "v"(bq[2][3]), "v"(bq[3][0]), "v"(bq[3][1]), "v"(bq[3][2]), "v"(bq[3][3]),
                       "s"(ntiles), "s"(nt), "s"(trsrc), "s"(nrsrc), "s"(lds_tile_addr), "s"(wave_s)
                     : ESFM_HMX1_SEGMENT_CLOBBERS);
    }
#ifdef ESFM_HMX1_TRACE
    const uint64_t tr2 = __builtin_amdgcn_s_memrealtime(), clk2 = __builtin_amdgcn_s_memtime();
#endif
    {   // (nothing thread-dependent lives across the block: see l2_knn_bf16x1_kernel)
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        lane = l; tid = wave_s * 64 + l; j = l & 31; h = l >> 5;
    }
    float key0[NS], key1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        key0[s] = reinterpret_cast<const float *>(smem)[(K * s + 0) * 256 + tid];
        key1[s] = reinterpret_cast<const float *>(smem)[(K * s + 1) * 256 + tid];
    }
    if (ntiles == 0) {
        float big;
        asm volatile("s_mov_b32 %0, 0x7f61b1e6" : "=s"(big));
#pragma unroll
        for (int s = 0; s < NS; ++s) { key0[s] = big; key1[s] = big; }
    }
    // ---- tail: exact (distance, index)-first two rows of every query that the screen lets through
    constexpr uint32_t kNone = 0xFFFFFFFFu, kIdxMask = 0x1FFFFFu;
    auto key_insert_min = [](uint32_t &k1, uint32_t &k2, uint32_t key) {
        const uint32_t hi = max(k1, key);
        k1 = min(k1, key);
        k2 = min(k2, hi);
    };
    const u32x4 *P = reinterpret_cast<const u32x4 *>(packed);
    const int qbase = qblk * QB + wave_s * 32 * NS;
    float fltmax; int minus2;
    asm volatile("s_mov_b32 %0, 0x7f7fffff" : "=s"(fltmax));
    asm volatile("s_mov_b32 %0, -2" : "=s"(minus2));
#pragma unroll 1                         // (unrolled by 2 / 4 -- the four sets' row loads in flight together -- measured: 0.442 / 0.441 ms against 0.443)
    for (int s = 0; s < NS; ++s) {
        const int qrow = qbase + 32 * s + j;
        const bool qvalid = qrow < nq;
        const float v0 = s == 0 ? key0[0] : (s == 1 ? key0[1] : (s == 2 ? key0[2] : key0[3]));
        const float v1 = s == 0 ? key1[0] : (s == 1 ? key1[1] : (s == 2 ? key1[2] : key1[3]));
        const float p0 = other_half(v0, h != 0), p1 = other_half(v1, h != 0);
        const float k0 = fminf(v0, p0), kb = fminf(fmaxf(v0, p0), fminf(v1, p1));           // the two smallest of the eight keys
        const float thr = __uint_as_float(__float_as_uint(kb) & ~kCodeMask);                   // ... the second one's score
        const float qpop = start[q_row0 + (qvalid ? qrow : 0)] - 512.f;
        // ratio screen (exact): d0 and an upper bound of d1
        const double d0 = (double)(__uint_as_float(__float_as_uint(k0) & ~kCodeMask) - 512.f + qpop), U1 = (double)(thr - 512.f + qpop);
        const bool rej = qvalid && kb < 1.0e38f && d0 >= ratio * U1;                           // (+inf ratio: never; one row only: re-rank)
        uint32_t k1 = kNone, k2 = kNone;
        const u32x4 *qp = P + ((size_t)q_row0 + (qvalid ? qrow : 0)) * 2;
        const u32x4 q0 = qp[0], q1 = qp[1];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float key = i == 0 ? v0 : v1;
            const float score = __uint_as_float(__float_as_uint(key) & ~kCodeMask);
            const bool need = qvalid && !rej && key < 1.0e38f && score <= thr;
            if (need) {
                const int code = (int)(__float_as_uint(key) & kCodeMask);
                const int row0 = (code / NG) * 32 + (32 / NG) * (code % NG) + 4 * h;
#pragma unroll
                for (int hb = 0; hb < GRP / 8; ++hb) {         // eight rows at a time: rows {0..3, 8..11} (+ 16 hb) of the step, + 4 h
                    u32x4 t0[8], t1[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = min(row0 + 16 * hb + (u & 3) + 8 * (u >> 2), max(nt - 1, 0));
                        const u32x4 *tp = P + ((size_t)t_row0 + t) * 2;
                        t0[u] = tp[0]; t1[u] = tp[1];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = row0 + 16 * hb + (u & 3) + 8 * (u >> 2);
                        const int dist = __popc(q0[0] ^ t0[u][0]) + __popc(q0[1] ^ t0[u][1]) + __popc(q0[2] ^ t0[u][2]) + __popc(q0[3] ^ t0[u][3]) +
                                         __popc(q1[0] ^ t1[u][0]) + __popc(q1[1] ^ t1[u][1]) + __popc(q1[2] ^ t1[u][2]) + __popc(q1[3] ^ t1[u][3]);
                        key_insert_min(k1, k2, t < nt ? (((uint32_t)dist << 21) | (uint32_t)t) : kNone);
                    }
                }
            }
        }
        const uint32_t o1 = __float_as_uint(other_half(__uint_as_float(k1), h != 0)), o2 = __float_as_uint(other_half(__uint_as_float(k2), h != 0));
        key_insert_min(k1, k2, o1);
        key_insert_min(k1, k2, o2);
        if (h == 0 && qvalid) {
            const size_t o = 2 * ((size_t)pd.out_off + qrow);
            const bool h0 = !rej && k1 != kNone, h1 = !rej && k2 != kNone;
            const int i0 = rej ? minus2 : (h0 ? (int)(k1 & kIdxMask) : -1), i1 = rej ? minus2 : (h1 ? (int)(k2 & kIdxMask) : -1);
            const float f0 = h0 ? (float)(k1 >> 21) : fltmax, f1 = h1 ? (float)(k2 >> 21) : fltmax;
            if (done) {       // another workgroup of this launch reads the records (the ratio stage below): write-through stores
                st_coh_i(knn_idx + o, i0); st_coh_i(knn_idx + o + 1, i1); st_coh_f(knn_dist + o, f0); st_coh_f(knn_dist + o + 1, f1);
            } else {
                *reinterpret_cast<int2 *>(knn_idx + o) = make_int2(i0, i1);
                *reinterpret_cast<float2 *>(knn_dist + o) = make_float2(f0, f1);
            }
        }
    }
#ifdef ESFM_HMX1_TRACE
    if (lane == 0) {
        const uint64_t tr3 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&g_hmx1_trace[0], (int)(tr1 - tr0)); atomicAdd(&g_hmx1_trace[1], (int)(tr2 - tr1)); atomicAdd(&g_hmx1_trace[2], (int)(tr3 - tr2));
        atomicAdd(&g_hmx1_trace[3], 1); atomicAdd(&g_hmx1_trace[4], (int)((clk2 - clk1) >> 8));
    }
#endif
    // ---- the match entry points: ratio test + ordered compaction of the pair by the workgroup that brings its last block (the
    // protocol of l2_finish_kernel: stores acknowledged, barrier, one relaxed agent-scope arrival; `done` reads 0 again afterwards)
    if (done) {
        __shared__ int s_last, s_wave[4], s_base;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nblk = (nq + QB - 1) / QB;
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&done[pi], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
        __syncthreads();
        if (!s_last) return;
        if (threadIdx.x == 0) __hip_atomic_store(&done[pi], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ratio_compact_pair<256, 16, true>(pd, knn_idx, knn_dist, ratio, query_idx, train_idx, distance, n_out + pi, s_wave, &s_base);
    }
}

// ---------------------------------------------------------------------------------------------
// The ratio test + compaction as a launch of its own (ratio_compact_pair above): one workgroup per pair.  The 64-float L2 path does
// it inside l2_finish_kernel; this serves Hamming and the other L2 passes.
constexpr int kRatioThreads = 1024;     // 4096 queries per sweep of the workgroup: one round of loads for a 4096-row set
__global__ __launch_bounds__(kRatioThreads) void ratio_compact_kernel(const PairDesc *__restrict__ pairs, const int32_t *__restrict__ knn_idx,
                                                                      const float *__restrict__ knn_dist, double ratio,
                                                                      int32_t *__restrict__ query_idx, int32_t *__restrict__ train_idx,
                                                                      float *__restrict__ distance, int32_t *__restrict__ n_out)
{
    __shared__ int s_wave[kRatioThreads / 64];
    __shared__ int s_base;
    const PairDesc pd = pairs[blockIdx.x];
    ratio_compact_pair<kRatioThreads, 4>(pd, knn_idx, knn_dist, ratio, query_idx, train_idx, distance, n_out + blockIdx.x, s_wave, &s_base);
}

// ---------------------------------------------------------------------------------------------
// launchers

static inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

int launch_l2_norms(hipStream_t st, const float *desc, int dim, long long n_rows, float *norms)
{
    if (n_rows <= 0) return ESFM_OK;
    hipLaunchKernelGGL(l2_row_norms_kernel, dim3(div_up(n_rows, 256)), dim3(256), 0, st, desc, dim, n_rows, norms);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

bool l2_mfma_supported(int dim) { return dim == 64 || dim == 128; }

// 64-float descriptors take the split-bf16 pass (256 queries per workgroup); ESFM_L2_PASS=f32 in the environment keeps them on
// the f32-MFMA kernel (measurement only: bench.py reports both)
bool l2_bf16_pass(int dim)
{
    static const bool forced_f32 = [] { const char *e = getenv("ESFM_L2_PASS"); return e && strcmp(e, "f32") == 0; }();
    return dim == 64 && !forced_f32;
}
constexpr int kL2RescanQueries = 8;   // uncertified queries of one pair that share a pass over the train set (l2_rescan64_pairs_kernel)
constexpr int kL2BfSets = 2;     // query sets of 32 per wave in l2_knn_bf16_kernel (1: 3 waves per SIMD, measured 7-15 % slower)
int l2_query_block(int dim) { return l2_bf16_pass(dim) ? 128 * kL2BfSets : 128; }
size_t l2_split_bytes(int dim, long long total_rows) { return l2_bf16_pass(dim) ? (size_t)512 * (size_t)std::max(total_rows, 1LL) : 0; }

// one-product pass scratch: [bf16(t) image: 128 B/row][bf16(-2 q) image: 128 B/row][rho_t: 4 B/row][rho_q: 4 B/row]
// bf16(t) image, bf16(-2 q) image (128 B per row each), rho_t, rho_q (4 B per row each), per 256-row block max |row|^2 and max rho_t
size_t l2_hi_bytes(long long total_rows) { const size_t n = (size_t)std::max(total_rows, 1LL); return (128 + 128 + 4 + 4) * n + 8 * ((n + 255) / 256); }
static inline char *l2_hi_part(void *hi, long long total_rows, int part)
{
    const size_t n = (size_t)std::max(total_rows, 1LL);
    const size_t off[5] = {0, 128 * n, 256 * n, 260 * n, 264 * n};
    return static_cast<char *>(hi) + off[part];
}
bool l2_one_product_pass()
{
    static const bool off = [] { const char *e = getenv("ESFM_L2_PASS"); return e && strcmp(e, "bf16x3") == 0; }();
    return !off;
}

int launch_l2_split_bf16(hipStream_t st, const float *desc, long long total_rows, void *split, float *norms, int32_t *counters,
                         int32_t *pair_cnt, int n_pairs, void *hi, int32_t *pair_cnt2)
{
    // `split` holds two images of 256 B per row: the train operand, then the query operand (-2 x); NULL when only the one-product
    // pass and its refine pass follow (they read the dense hi images in `hi`): 52 MB less to write per 25 x 4096 rows
    const long long n_pieces = std::max(total_rows * 16, (long long)std::max(n_pairs, 16));     // the launch also zeroes counters / pair_cnt
    hipLaunchKernelGGL(l2_split_bf16_kernel, dim3((unsigned)((n_pieces + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4 *>(desc), total_rows * 16, reinterpret_cast<u32x4 *>(split),
                       split ? reinterpret_cast<u32x4 *>(split) + (size_t)std::max(total_rows, 1LL) * 16 : nullptr, norms, counters, pair_cnt, n_pairs,
                       hi ? reinterpret_cast<u32x4 *>(l2_hi_part(hi, total_rows, 0)) : nullptr,
                       hi ? reinterpret_cast<u32x4 *>(l2_hi_part(hi, total_rows, 1)) : nullptr,
                       hi ? reinterpret_cast<float *>(l2_hi_part(hi, total_rows, 2)) : nullptr,
                       hi ? reinterpret_cast<float *>(l2_hi_part(hi, total_rows, 3)) : nullptr, pair_cnt2);
    ESFM_HIP_TRY(hipGetLastError());
    if (hi && total_rows > 0) {
        hipLaunchKernelGGL(l2_blockmax_kernel, dim3((unsigned)((total_rows + 255) / 256)), dim3(256), 0, st, norms,
                           reinterpret_cast<const float *>(l2_hi_part(hi, total_rows, 2)), total_rows, reinterpret_cast<float2 *>(l2_hi_part(hi, total_rows, 4)));
        ESFM_HIP_TRY(hipGetLastError());
    }
    return ESFM_OK;
}

int launch_l2_knn_bf16(hipStream_t st, const float *desc, const void *split, long long total_rows, const float *norms, const PairDesc *pairs,
                       int n_pairs, int n_blocks, int32_t *knn_idx, float *knn_dist, int32_t *flagged, int32_t *counters, int flag_cap,
                       int32_t *pair_cnt, int32_t *pair_list)
{
    if (n_blocks <= 0) return ESFM_OK;
    constexpr int TT = 128;   // train rows per LDS tile: one barrier per 96 MFMAs per wave
    constexpr size_t lds = 2 * TT * 16 * 16 + 2 * TT * 4 + 16 + kL2BfSets * 6 * 256 * 4;   // two tiles, their norms, the master top-3
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    const u32x4 *sp = reinterpret_cast<const u32x4 *>(split);
    const u32x4 *sq = sp + (size_t)std::max(total_rows, 1LL) * 16;
    hipLaunchKernelGGL(l2_knn_bf16_kernel, dim3(n_blocks), dim3(256), lds, st, desc, sp, sq, norms, pairs,
                       n_pairs, knn_idx, knn_dist, flagged, counters, flag_cap, pair_cnt, pair_list);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

// the ratio screen's constant: ratio^2 (1 + 2^-20); a ratio that is not a finite number >= 0 switches the screen off
static inline double l2_ratio2m(double ratio) { return (ratio >= 0.0 && ratio < 1.0e150) ? ratio * ratio * (1.0 + 1.0 / 1048576.0) : (double)INFINITY; }

int l2_x1_query_block() { return l2x1_query_block_c; }
bool l2_x1_supported(int max_nt) { return max_nt <= (1 << (ESFM_L2X1_CODE_BITS - (ESFM_L2X1_GRP == 4 ? 2 : 1))) * 32; }   // the position code names a 32-row step and one of its 16 / GRP groups

int launch_l2_knn_bf16x1(hipStream_t st, int num_cu, const float *desc, const void *hi, long long total_rows, const float *norms, const PairDesc *pairs,
                         const int32_t *blk_pair, int n_blocks, int32_t *knn_idx, float *knn_dist, int32_t *counters, int flag_cap,
                         int32_t *surv_cnt, void *surv_list, double ratio, bool markers, int32_t *rejected,
                         int32_t *zero_a, int32_t *zero_b, int zero_n, int32_t *zero_counters)
{
    if (n_blocks <= 0) return ESFM_OK;
    // ring of bf16 tiles, their norms, two sets of reductions, two sets of query norms and residual norms
    constexpr size_t lds = 4 * 128 * 128 + 4 * 128 * 4 + 64 + 4 * l2x1_query_block_c * 4;
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    // (set on every launch, like the other large-LDS kernels: the attribute belongs to the current device)
    ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&l2_knn_bf16x1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void *h = const_cast<void *>(hi);
    // one workgroup per block by default (see the kernel); ESFM_X1_GRID = persistent workgroups (a multiple of 8, e.g. 2 x CUs)
    static const int forced = [] { const char *e = getenv("ESFM_X1_GRID"); return e ? atoi(e) : 0; }();     // (measurement)
    (void)num_cu;
    const int grid = forced > 0 && forced < n_blocks ? std::max(8, forced / 8 * 8) : n_blocks;
    hipLaunchKernelGGL(l2_knn_bf16x1_kernel, dim3(grid), dim3(256), lds, st, desc,
                       reinterpret_cast<const u32x4 *>(l2_hi_part(h, total_rows, 0)), reinterpret_cast<const u32x4 *>(l2_hi_part(h, total_rows, 1)), norms,
                       reinterpret_cast<const float *>(l2_hi_part(h, total_rows, 2)), reinterpret_cast<const float *>(l2_hi_part(h, total_rows, 3)),
                       reinterpret_cast<const float2 *>(l2_hi_part(h, total_rows, 4)),
                       pairs, blk_pair, n_blocks, knn_idx, knn_dist, counters, flag_cap, surv_cnt, reinterpret_cast<float4 *>(surv_list), l2_ratio2m(ratio), markers ? 1 : 0, rejected,
                       zero_a, zero_b, zero_n, zero_counters);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int l2_finish_slices(int n_pairs)
{
    static const int forced = [] { const char *e = getenv("ESFM_FIN_SLICES"); return e ? atoi(e) : 0; }();     // (measurement)
    if (forced > 0) return forced;
    return std::max(1, std::min(8, 2304 / std::max(n_pairs, 1)));      // (300 pairs, three workgroups per CU, survivors dealt evenly: 7 slices 60 us, 4: 70, 5: 65, 6: 66, 8: 63)
}

int launch_l2_finish(hipStream_t st, const float *desc, const void *hi, long long total_rows, const float *norms, const PairDesc *pairs,
                     const int32_t *pair_order, int n_pairs, const int32_t *surv_cnt, const void *surv_list, int32_t *unc_cnt, int32_t *unc_list, float *knn_d2,
                     int32_t *knn_idx, float *knn_dist, int32_t *counters, int32_t *flagged, int flag_cap, int32_t *done,
                     int audit, bool do_ratio, double ratio, int32_t *query_idx, int32_t *train_idx, float *distance, int32_t *n_out)
{
    if (n_pairs <= 0) return ESFM_OK;
    const int S = l2_finish_slices(n_pairs);
    ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&l2_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFinLdsBytes));
    void *h = const_cast<void *>(hi);
    hipLaunchKernelGGL(l2_finish_kernel, dim3((unsigned)n_pairs * (unsigned)S), dim3(kFinThreads), kFinLdsBytes, st, desc,
                       reinterpret_cast<const u32x4 *>(l2_hi_part(h, total_rows, 0)), reinterpret_cast<const u32x4 *>(l2_hi_part(h, total_rows, 1)), norms,
                       reinterpret_cast<const float *>(l2_hi_part(h, total_rows, 2)), reinterpret_cast<const float *>(l2_hi_part(h, total_rows, 3)),
                       pairs, pair_order, n_pairs, S, surv_cnt, reinterpret_cast<const float4 *>(surv_list), unc_cnt, unc_list, knn_d2, knn_idx, knn_dist, counters, flagged,
                       flag_cap, done, audit, do_ratio ? 1 : 0, ratio, l2_ratio2m(ratio), query_idx, train_idx, distance, n_out);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}
size_t l2_survivor_entry_bytes() { return 48; }

int launch_l2_rescan64_pairs(hipStream_t st, const float *desc, const PairDesc *pairs, int n_pairs, const int32_t *pair_cnt,
                             const int32_t *pair_list, int32_t *knn_idx, float *knn_dist)
{
    if (n_pairs <= 0) return ESFM_OK;
    // about 4096 workgroups whatever the pair count: a workgroup without work leaves after one load.  Few pairs: the launch is as
    // long as its longest workgroup (a 16-step latency chain per 4096 train rows), so the chunks are small -- more workgroups, two
    // per CU; many pairs: throughput counts, the chunks are as large as the kernel's LDS allows (M-SURF-8k-like launch of 2415
    // pairs: 1.85 ms with chunks of 2, 1.33 with 3, 1.28 with 8).
    const int chunks_per_pair = std::max(1, std::min(512, 4096 / n_pairs));
    ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&l2_rescan64_pairs_kernel<kL2RescanQueries>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 256 + kL2RescanQueries * 256 * 16));
    const int chunk = n_pairs >= 2048 ? kL2RescanQueries : 3;     // 3: 64 + 12 + 3 KB of LDS, still two workgroups per CU (measured 1: 69, 2: 65, 3: 60, 4: 92 us)
    hipLaunchKernelGGL(l2_rescan64_pairs_kernel<kL2RescanQueries>, dim3((unsigned)n_pairs * (unsigned)chunks_per_pair), dim3(256),
                       (size_t)4 * 64 * 256 + (size_t)chunk * 256 * 16, st, desc, pairs,
                       pair_cnt, pair_list, chunks_per_pair, chunk, knn_idx, knn_dist);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_l2_knn_mfma(hipStream_t st, int dim, const float *desc, const float *norms, const PairDesc *pairs, int n_pairs,
                       int n_blocks, int32_t *knn_idx, float *knn_dist, int32_t *flagged, int32_t *counters, int flag_cap)
{
    if (n_blocks <= 0) return ESFM_OK;
    // train tile = 128 rows (one barrier per 128 MFMAs per wave); LDS = 2 x TT x DIM x 4 B + norms
    if (dim == 64) {
        constexpr int TT = 128;
        constexpr size_t lds = 2 * TT * 16 * 16 + 2 * TT * 4 + 16;
        hipLaunchKernelGGL((l2_knn_mfma_kernel<64, TT>), dim3(n_blocks), dim3(256), lds, st, desc, norms, pairs, n_pairs, knn_idx,
                           knn_dist, flagged, counters, flag_cap);
    } else if (dim == 128) {
        constexpr int TT = 64;
        constexpr size_t lds = 2 * TT * 32 * 16 + 2 * TT * 4 + 16;
        hipLaunchKernelGGL((l2_knn_mfma_kernel<128, TT>), dim3(n_blocks), dim3(256), lds, st, desc, norms, pairs, n_pairs, knn_idx,
                           knn_dist, flagged, counters, flag_cap);
    } else {
        set_error("l2 MFMA kernel is built for dim 64 and 128 only (got %d)", dim);
        return ESFM_ERR_UNSUPPORTED;
    }
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_l2_exact_scan(hipStream_t st, int dim, const float *desc, const PairDesc *pairs, int n_pairs,
                         const int32_t *flagged, const int32_t *counters, long long total_queries, int grid,
                         int32_t *knn_idx, float *knn_dist)
{
    if (grid <= 0) return ESFM_OK;
    if (dim == 64 && flagged)
        hipLaunchKernelGGL(l2_rescan64_kernel, dim3(grid), dim3(256), 0, st, desc, pairs, flagged, counters, knn_idx, knn_dist);
    else if (dim % 4 == 0)
        hipLaunchKernelGGL(l2_exact_scan_kernel<true>, dim3(grid), dim3(256), 0, st, desc, dim, pairs, n_pairs, flagged, counters,
                           total_queries, knn_idx, knn_dist);
    else
        hipLaunchKernelGGL(l2_exact_scan_kernel<false>, dim3(grid), dim3(256), 0, st, desc, dim, pairs, n_pairs, flagged, counters,
                           total_queries, knn_idx, knn_dist);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

bool hamming_supported(int nbytes) { return nbytes == 16 || nbytes == 32 || nbytes == 64; }

// queries per workgroup of the kernel that serves this descriptor width (the pair plan's block count depends on it)
int hamming_query_block(int nbytes) { return 256; }

// the 0/1 byte image of every descriptor followed by one start value (256 - popcount) per row
size_t hamming_expanded_bytes(int nbytes, long long total_rows) { return nbytes == 32 ? (size_t)(256 + 4) * (size_t)std::max(total_rows, 1LL) : 0; }

int launch_hamming_expand(hipStream_t st, int nbytes, const void *desc, long long total_rows, void *exp_scratch)
{
    if (nbytes != 32 || !exp_scratch || total_rows <= 0) return ESFM_OK;
    const long long n_words = total_rows * 8;
    int32_t *start = reinterpret_cast<int32_t *>(static_cast<unsigned char *>(exp_scratch) + (size_t)256 * (size_t)std::max(total_rows, 1LL));
    hipLaunchKernelGGL(hamming_expand_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint32_t *>(desc), n_words,
                       reinterpret_cast<uint32_t *>(exp_scratch), start);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_hamming_knn(hipStream_t st, int nbytes, const void *desc, long long total_rows, void *exp_scratch, const PairDesc *pairs,
                       int n_pairs, int n_blocks, int32_t *knn_idx, float *knn_dist, bool expanded)
{
    if (n_blocks <= 0) return ESFM_OK;
    const uint32_t *d = reinterpret_cast<const uint32_t *>(desc);
    if (nbytes == 32 && exp_scratch) {
        int32_t *start = reinterpret_cast<int32_t *>(static_cast<unsigned char *>(exp_scratch) + (size_t)256 * (size_t)std::max(total_rows, 1LL));
        if (!expanded)
            if (int rc = launch_hamming_expand(st, nbytes, desc, total_rows, exp_scratch)) return rc;
        hipLaunchKernelGGL(hamming_knn_mfma_kernel, dim3(n_blocks), dim3(256), 0, st, reinterpret_cast<const unsigned char *>(exp_scratch),
                           start, d, pairs, n_pairs, knn_idx, knn_dist);
    } else if (nbytes == 32)
        hipLaunchKernelGGL(hamming_knn_kernel<8>, dim3(n_blocks), dim3(256), 0, st, d, pairs, n_pairs, knn_idx, knn_dist);
    else if (nbytes == 64)
        hipLaunchKernelGGL(hamming_knn_kernel<16>, dim3(n_blocks), dim3(256), 0, st, d, pairs, n_pairs, knn_idx, knn_dist);
    else if (nbytes == 16)
        hipLaunchKernelGGL(hamming_knn_kernel<4>, dim3(n_blocks), dim3(256), 0, st, d, pairs, n_pairs, knn_idx, knn_dist);
    else {
        set_error("hamming kernel is built for 16/32/64-byte descriptors (got %d)", nbytes);
        return ESFM_ERR_UNSUPPORTED;
    }
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

bool hamming_fp4_pass()
{
    static const bool off = [] { const char *e = getenv("ESFM_HM_PASS"); return e && strcmp(e, "i8") == 0; }();
    return !off;
}
bool hamming_fp4_supported(int nbytes, int max_nt) { return nbytes == 32 && hamming_fp4_pass() && max_nt <= (1 << (ESFM_HMX1_CODE_BITS - 1)) * 32 && max_nt < (1 << 21); }

// the FP4 form's operands: nibble images of every row in both roles (128 B each) and pop(row) + 512 as a float -- the same 260 B per
// row as the byte image + start value of the i8 form (hamming_expanded_bytes)
int launch_hamming_expand_fp4(hipStream_t st, const void *desc, long long total_rows, void *exp_scratch)
{
    if (!exp_scratch || total_rows <= 0) return ESFM_OK;
    const long long n_words = total_rows * 8;
    unsigned char *base = static_cast<unsigned char *>(exp_scratch);
    const size_t n = (size_t)std::max(total_rows, 1LL);
    hipLaunchKernelGGL(hamming_expand_fp4_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint32_t *>(desc), n_words,
                       reinterpret_cast<u32x4 *>(base), reinterpret_cast<u32x4 *>(base + 128 * n), reinterpret_cast<float *>(base + 256 * n));
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_hamming_fp4(hipStream_t st, const void *desc, long long total_rows, void *exp_scratch, const PairDesc *pairs, const int32_t *blk_pair,
                       int n_blocks, int32_t *knn_idx, float *knn_dist, double ratio, bool expanded, int32_t *done, int n_pairs, int32_t *query_idx,
                       int32_t *train_idx, float *distance, int32_t *n_out)
{
    if (n_blocks <= 0) return ESFM_OK;
    if (!expanded)
        if (int rc = launch_hamming_expand_fp4(st, desc, total_rows, exp_scratch)) return rc;
    constexpr size_t lds = 4 * 128 * 128 + 4 * 128 * 4 + 64;      // ring of nibble tiles, their start values (the keys leave through the ring)
    ESFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&hamming_fp4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned char *base = static_cast<unsigned char *>(exp_scratch);
    const size_t n = (size_t)std::max(total_rows, 1LL);
    // (the ratio test's own compare is `(double) d0 < ratio * (double) d1`: a NaN or negative ratio rejects nothing here)
    const double r = (ratio >= 0.0 && ratio < 1.0e150) ? ratio : (double)INFINITY;
    hipLaunchKernelGGL(hamming_fp4_kernel, dim3(n_blocks), dim3(256), lds, st, reinterpret_cast<const uint32_t *>(desc), reinterpret_cast<const u32x4 *>(base),
                       reinterpret_cast<const u32x4 *>(base + 128 * n), reinterpret_cast<const float *>(base + 256 * n), pairs, blk_pair, n_blocks, knn_idx,
                       knn_dist, done ? ratio : r, done, n_pairs, query_idx, train_idx, distance, n_out);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

// ---------------------------------------------------------------------------------------------
// Fingerprint of a resident descriptor buffer (esfm_ctx_set_prepared_check): a position-keyed 64-bit sum over its 4-byte words --
// integer addition, so the order in which the waves arrive does not matter.  One word of `out` is added to (zeroed by the caller).
__global__ __launch_bounds__(256) void buffer_checksum_kernel(const uint32_t *__restrict__ p, long long n_words, unsigned long long *__restrict__ out)
{
    unsigned long long h = 0ull;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (long long)gridDim.x * 256) {
        unsigned long long x = ((unsigned long long)p[i] << 32 | (unsigned long long)(uint32_t)i) ^ ((unsigned long long)(i >> 32) << 17);
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;      // (murmur3's finaliser)
        h += x;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) h += __shfl_xor(h, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

// esfm_match_pairs (host pointers): the pairs' match lists, each at its own offset in three sum(nq)-long arrays, packed back to
// back so that the read-back moves the matches and not the gaps.  tab: per pair {source offset, packed offset} (int64) and count.
__global__ __launch_bounds__(256) void pack_match_lists_kernel(const long long *__restrict__ tab, const int32_t *__restrict__ n_out, int n_pairs,
                                                               const int32_t *__restrict__ sq, const int32_t *__restrict__ stn, const float *__restrict__ sd,
                                                               int32_t *__restrict__ dq, int32_t *__restrict__ dtn, float *__restrict__ dd)
{
    for (int p = blockIdx.x; p < n_pairs; p += gridDim.x) {
        const long long so = tab[2 * (size_t)p], dof = tab[2 * (size_t)p + 1];
        const int n = n_out[p];
        for (int e = threadIdx.x; e < n; e += 256) { dq[dof + e] = sq[so + e]; dtn[dof + e] = stn[so + e]; dd[dof + e] = sd[so + e]; }
    }
}

int launch_pack_match_lists(hipStream_t st, const long long *tab, const int32_t *n_out, int n_pairs, const int32_t *sq, const int32_t *stn, const float *sd,
                            int32_t *dq, int32_t *dtn, float *dd)
{
    if (n_pairs <= 0) return ESFM_OK;
    hipLaunchKernelGGL(pack_match_lists_kernel, dim3(std::min(n_pairs, 4096)), dim3(256), 0, st, tab, n_out, n_pairs, sq, stn, sd, dq, dtn, dd);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_buffer_checksum(hipStream_t st, const void *buf, size_t bytes, unsigned long long *out)
{
    ESFM_HIP_TRY(hipMemsetAsync(out, 0, sizeof(unsigned long long), st));
    const long long n_words = (long long)(bytes / 4);
    if (n_words <= 0) return ESFM_OK;
    const int grid = (int)std::min<long long>((n_words + 255) / 256, 2048);
    hipLaunchKernelGGL(buffer_checksum_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint32_t *>(buf), n_words, out);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_ratio_compact(hipStream_t st, const PairDesc *pairs, int n_pairs, const int32_t *knn_idx, const float *knn_dist,
                         double ratio, int32_t *query_idx, int32_t *train_idx, float *distance, int32_t *n_out)
{
    if (n_pairs <= 0) return ESFM_OK;
    hipLaunchKernelGGL(ratio_compact_kernel, dim3(n_pairs), dim3(kRatioThreads), 0, st, pairs, knn_idx, knn_dist, ratio, query_idx,
                       train_idx, distance, n_out);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm
