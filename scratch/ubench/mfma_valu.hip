// microbench: how much VALU of which kind hides under v_mfma_f32_32x32x2_f32 on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    floatx16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = seed * r; acc1[r] = seed + r; }
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v0 = seed, v1 = seed + 1, v2 = seed + 2; int c0 = 0, c1 = 1, c2 = 2;
    float x = seed * 3 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (KIND == 0) { x = fmaf(x, 1.0001f, 0.5f); }                                 // dependent fma chain
                if (KIND == 1) { v2 = __builtin_amdgcn_fmed3f(v1, v2, x); v1 = __builtin_amdgcn_fmed3f(v0, v1, x); x += 1.0f; }   // med3 (2 ops + add)
                if (KIND == 2) { bool l = x < v0; c0 = l ? c1 : c0; v0 = l ? x : v0; x += 1.0f; }   // cmp + 2 cndmask + add
                if (KIND == 3) { unsigned u = (__float_as_uint(x) & 0xFFFFF800u) | (unsigned)(it + q); v0 = __builtin_amdgcn_fmed3f(v0, __uint_as_float(u), -1e30f); x += 1.0f; }
            }
        }
        asm volatile("" : "+v"(x), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(c0));
    }
    float s = x + v0 + v1 + v2 + c0 + c1 + c2;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NV> void run(const char *name, float *d, int blocks_per_cu)
{
    const int iters = 200, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(grid), dim3(256), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks_per_cu waves, each iters*64 MFMAs
    double cyc_per_mfma = ms * 1e-3 * 2.4e9 / (double(iters) * 64 * blocks_per_cu);
    printf("%-28s NV=%d waves/SIMD=%d : %.3f ms, %.1f cycles per MFMA per SIMD (64 = MFMA-bound)\n", name, NV, blocks_per_cu, ms, cyc_per_mfma);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>("mfma only", d, w);
        run<0, 4>("fma chain", d, w); run<0, 8>("fma chain", d, w); run<0, 16>("fma chain", d, w);
        run<1, 2>("med3 x2 + add", d, w); run<1, 4>("med3 x2 + add", d, w);
        run<2, 1>("cmp + 2 cndmask + add", d, w); run<2, 2>("cmp + 2 cndmask + add", d, w); run<2, 4>("cmp + 2 cndmask + add", d, w);
        run<3, 2>("and_or + med3 + add", d, w); run<3, 4>("and_or + med3 + add", d, w);
    }
    return 0;
}
