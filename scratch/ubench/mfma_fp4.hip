// v_mfma_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands against v_mfma_i32_32x32x32_i8 and v_mfma_f32_32x32x16_bf16: issue rate, and that
// 0/1 x 0/-2 nibble products accumulate exactly (the byte-per-bit Hamming matcher's arithmetic at a quarter of the operand bytes).
// hipcc -O3 --offload-arch=gfx950 mfma_fp4.hip -o mfma_fp4 && ./mfma_fp4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx16 __attribute__((ext_vector_type(16)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void rate(float *out, int iters)
{
    const int lane = threadIdx.x & 63;
    intx8 a8 = {lane, lane * 3, 7, 9, 0, 0, 0, 0}, b8 = {lane * 5, 11, lane, 13, 0, 0, 0, 0};
    floatx16 f[4] = {};
    intx16 ii[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) f[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f[u], 4, 4, 0, 0, 0, 0);
            if (MODE == 1) ii[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(intx4{a8[0], a8[1], a8[2], a8[3]}, intx4{b8[0], b8[1], b8[2], b8[3]}, ii[u], 0, 0, 0);
            if (MODE == 2) f[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, intx4{a8[0], a8[1], a8[2], a8[3]}), __builtin_bit_cast(bf16x8, intx4{b8[0], b8[1], b8[2], b8[3]}), f[u], 0, 0, 0);
            if (MODE == 3) f[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f[u], 0, 0, 0, 0, 0, 0);      // fp8 x fp8
        }
    }
    float s = 0.f;
    for (int u = 0; u < 4; ++u) for (int e = 0; e < 16; ++e) s += f[u][e] + (float)ii[u][e];
    if (s == 123.f) out[0] = s;
}

// exactness: row r of A = bits of a random 256-bit string as nibbles 0x2 (1.0), column c of B = bits as 0xC (-2.0): acc = -2 a.b
__global__ void exact(const uint32_t *abits, const uint32_t *bbits, float *out)
{
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    floatx16 acc = {};
    for (int ks = 0; ks < 4; ++ks) {          // K-step ks covers bits 64 ks .. 64 ks + 63; this lane's 32 of them: 64 ks + 32 h ..
        intx8 a = {}, b = {};
        for (int w = 0; w < 4; ++w) {         // 8 nibbles per dword
            uint32_t av = 0, bv = 0;
            for (int n = 0; n < 8; ++n) {
                const int bit = 64 * ks + 32 * h + 8 * w + n;
                av |= (((abits[j * 8 + bit / 32] >> (bit % 32)) & 1u) ? 0x2u : 0u) << (4 * n);
                bv |= (((bbits[j * 8 + bit / 32] >> (bit % 32)) & 1u) ? 0xCu : 0u) << (4 * n);
            }
            a[w] = (int)av; b[w] = (int)bv;
        }
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) out[lane * 16 + e] = acc[e];
}

int main()
{
    float *out; hipMalloc(&out, 64 * 16 * 4 + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * 2;
    const char *names[4] = {"fp4 32x32x64", "i8 32x32x32", "bf16 32x32x16", "fp8 32x32x64"};
    const double ops[4] = {2.0 * 32 * 32 * 64, 2.0 * 32 * 32 * 32, 2.0 * 32 * 32 * 16, 2.0 * 32 * 32 * 64};
    for (int m = 0; m < 4; ++m) {
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(rate<0>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (m == 1) hipLaunchKernelGGL(rate<1>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (m == 2) hipLaunchKernelGGL(rate<2>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (m == 3) hipLaunchKernelGGL(rate<3>, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)grid * 4 * iters * 4;
        printf("%-14s %8.3f ms  %.2f Pop/s  (%.1f ns per MFMA per SIMD at 2 waves)\n", names[m], ms, n * ops[m] / ms * 1e-12, ms * 1e6 / (iters * 4.0 * 2));
    }
    // exactness
    std::vector<uint32_t> ha(32 * 8), hb(32 * 8);
    uint32_t s = 12345u;
    for (auto &v : ha) { s = s * 1664525u + 1013904223u; v = s; }
    for (auto &v : hb) { s = s * 1664525u + 1013904223u; v = s; }
    uint32_t *da, *db; hipMalloc(&da, 1024); hipMalloc(&db, 1024);
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact, dim3(1), dim3(64), 0, 0, da, db, out);
    std::vector<float> ho(64 * 16);
    hipMemcpy(ho.data(), out, 64 * 16 * 4, hipMemcpyDeviceToHost);
    // D[i][j]: lane = j + 32 * ((i / 4) % 2), element = (i % 4) + 4 * (i / 8)   (32x32 f32 layout: rows i = 8 (e / 4) + 4 h + e % 4)
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 16; ++e) {
        const int col = lane & 31, row = 8 * (e / 4) + 4 * (lane >> 5) + (e % 4);
        int dot = 0;
        for (int w = 0; w < 8; ++w) dot += __builtin_popcount(ha[row * 8 + w] & hb[col * 8 + w]);
        if (ho[lane * 16 + e] != -2.0f * dot) { if (bad < 5) printf("mismatch row %d col %d: %g vs %d\n", row, col, ho[lane * 16 + e], -2 * dot); ++bad; }
    }
    printf("exactness: %d mismatches of 1024 (A rows x B columns, 256-bit strings)\n", bad);
    return 0;
}
