// microbench: how much of the top-3 fold (v_and_or + 3 v_med3 per element) hides under v_mfma_f32_32x32x16_bf16 on gfx950,
// at 1 and 2 waves per SIMD, with constant and with random operands (the matrix pipe's clock is power-managed)?
// Reports shader cycles per MFMA per SIMD assuming 2.4 GHz (32 = matrix-pipe bound) and the wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NE>   // NE = elements folded per MFMA (4 VALU each)
__global__ __launch_bounds__(256) void k(const u32x4 *in, float *out, int iters)
{
    u32x4 av[4], bv[4];
    for (int i = 0; i < 4; ++i) { av[i] = in[(threadIdx.x * 8 + i) & 4095]; bv[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
    floatx16 acc0, acc1, p0, p1;
    for (int r = 0; r < 16; ++r) { acc0[r] = r; acc1[r] = r + 1; p0[r] = 1e30f; p1[r] = 1e30f; }
    float k0 = 3e38f, k1 = 3e38f, k2 = 3e38f, j0 = 3e38f, j1 = 3e38f, j2 = 3e38f;
    unsigned kmask = 0xFFFFFF00u;
    asm volatile("" : "+v"(kmask));
    for (int it = 0; it < iters; ++it) {
        const int code = __builtin_amdgcn_readfirstlane((it & 15) * 16);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[m & 3]), __builtin_bit_cast(bf16x8, bv[(m + 1) & 3]), acc0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float key = __uint_as_float((__float_as_uint(p0[(m * NE + e) & 15]) & kmask) | (unsigned)(code + ((m * NE + e) & 15)));
                k2 = __builtin_amdgcn_fmed3f(k1, k2, key); k1 = __builtin_amdgcn_fmed3f(k0, k1, key); k0 = __builtin_amdgcn_fmed3f(k0, key, -3e38f);
            }
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[(m + 2) & 3]), __builtin_bit_cast(bf16x8, bv[m & 3]), acc1, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float key = __uint_as_float((__float_as_uint(p1[(m * NE + e) & 15]) & kmask) | (unsigned)(code + ((m * NE + e) & 15)));
                j2 = __builtin_amdgcn_fmed3f(j1, j2, key); j1 = __builtin_amdgcn_fmed3f(j0, j1, key); j0 = __builtin_amdgcn_fmed3f(j0, key, -3e38f);
            }
        }
        // swap roles like the real kernel: the results just produced are folded next
        floatx16 t0 = p0, t1 = p1; p0 = acc0; p1 = acc1; acc0 = t0; acc1 = t1;
    }
    float s = k0 + k1 + k2 + j0 + j1 + j2;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + p0[r] + p1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// phased variant: all 24 MFMAs of a step back to back, then the whole fold of the previous step's results
template <int NE>
__global__ __launch_bounds__(256) void kphase(const u32x4 *in, float *out, int iters)
{
    u32x4 av[4], bv[4];
    for (int i = 0; i < 4; ++i) { av[i] = in[(threadIdx.x * 8 + i) & 4095]; bv[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
    floatx16 acc0, acc1, p0, p1;
    for (int r = 0; r < 16; ++r) { acc0[r] = r; acc1[r] = r + 1; p0[r] = 1e30f; p1[r] = 1e30f; }
    float k0 = 3e38f, k1 = 3e38f, k2 = 3e38f, j0 = 3e38f, j1 = 3e38f, j2 = 3e38f;
    unsigned kmask = 0xFFFFFF00u;
    asm volatile("" : "+v"(kmask));
    for (int it = 0; it < iters; ++it) {
        const int code = __builtin_amdgcn_readfirstlane((it & 15) * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[m & 3]), __builtin_bit_cast(bf16x8, bv[(m + 1) & 3]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[(m + 2) & 3]), __builtin_bit_cast(bf16x8, bv[m & 3]), acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 12 * NE; ++e) {
            const float key = __uint_as_float((__float_as_uint(p0[e & 15]) & kmask) | (unsigned)(code + (e & 15)));
            k2 = __builtin_amdgcn_fmed3f(k1, k2, key); k1 = __builtin_amdgcn_fmed3f(k0, k1, key); k0 = __builtin_amdgcn_fmed3f(k0, key, -3e38f);
            const float key1 = __uint_as_float((__float_as_uint(p1[e & 15]) & kmask) | (unsigned)(code + (e & 15)));
            j2 = __builtin_amdgcn_fmed3f(j1, j2, key1); j1 = __builtin_amdgcn_fmed3f(j0, j1, key1); j0 = __builtin_amdgcn_fmed3f(j0, key1, -3e38f);
        }
        __builtin_amdgcn_sched_barrier(0);
        floatx16 t0 = p0, t1 = p1; p0 = acc0; p1 = acc1; acc0 = t0; acc1 = t1;
    }
    float s = k0 + k1 + k2 + j0 + j1 + j2;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + p0[r] + p1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NE> void runphase(const u32x4 *in, float *d, int waves_per_simd, const char *data)
{
    const int iters = 2000, grid = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kphase<NE>), dim3(grid), dim3(256), 0, 0, in, d, 500);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kphase<NE>), dim3(grid), dim3(256), 0, 0, in, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / (double(iters) * 24 * waves_per_simd);
    printf("%-8s PHASED: 24 MFMAs then fold of %d elements per MFMA (VALU per MFMA %2d)  waves/SIMD %d : %7.3f ms  %6.1f cycles@2.4GHz per MFMA per SIMD\n", data, NE, 4 * NE, waves_per_simd, ms, cyc);
}

template <int NACC>
__global__ __launch_bounds__(256) void kacc(const u32x4 *in, float *out, int iters)
{
    u32x4 av[4], bv[4];
    for (int i = 0; i < 4; ++i) { av[i] = in[(threadIdx.x * 8 + i) & 4095]; bv[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = r + a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m)
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[m & 3]), __builtin_bit_cast(bf16x8, bv[(m + 1) & 3]), acc[m % NACC], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void runacc(const u32x4 *in, float *d, int waves_per_simd, const char *data)
{
    const int iters = 2000, grid = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kacc<NACC>), dim3(grid), dim3(256), 0, 0, in, d, 2000);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kacc<NACC>), dim3(grid), dim3(256), 0, 0, in, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / (double(iters) * 24 * waves_per_simd);
    const double tf = double(iters) * 24 * waves_per_simd * 1024 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-8s MFMA only, %d independent accumulators, waves/SIMD %d : %7.3f ms  %6.1f cycles@2.4GHz per MFMA  %7.1f TFLOP/s\n", data, NACC, waves_per_simd, ms, cyc, tf);
}

template <int NE> void run(const u32x4 *in, float *d, int waves_per_simd, const char *data)
{
    const int iters = 2000, grid = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NE>), dim3(grid), dim3(256), 0, 0, in, d, 500);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NE>), dim3(grid), dim3(256), 0, 0, in, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / (double(iters) * 24 * waves_per_simd);
    printf("%-8s fold elements per MFMA %d (VALU per MFMA %2d)  waves/SIMD %d : %7.3f ms  %6.1f cycles@2.4GHz per MFMA per SIMD\n", data, NE, 4 * NE, waves_per_simd, ms, cyc);
}

int main()
{
    std::vector<unsigned> h(4096 * 4);
    u32x4 *in; float *d;
    hipMalloc(&in, h.size() * 4); hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int pass = 0; pass < 2; ++pass) {
        srand(1);
        for (auto &v : h) {
            if (pass == 0) v = 0x3F803F80u;                                        // bf16 1.0, 1.0
            else { unsigned a = 0x3F00u + (rand() & 0xFF) + ((rand() & 1) << 15), b = 0x3F00u + (rand() & 0xFF) + ((rand() & 1) << 15); v = a | (b << 16); }
        }
        hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const char *name = pass ? "random" : "const";
        for (int w = 1; w <= 2; ++w) { run<0>(in, d, w, name); run<1>(in, d, w, name); run<2>(in, d, w, name); }
        run<1>(in, d, 3, name); run<1>(in, d, 4, name);
        for (int w = 1; w <= 4; w *= 2) { runphase<1>(in, d, w, name); runphase<2>(in, d, w, name); }
        for (int w = 1; w <= 4; w *= 2) { runacc<1>(in, d, w, name); runacc<2>(in, d, w, name); runacc<4>(in, d, w, name); }
    }
    return 0;
}
