// issue cost of the candidates for potrf16's (pivot, column) update, one wave, dependent only through the accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
template <int MODE> __global__ void k(double *out, long long *cyc, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 * 2, a2 = a0 * 3, a3 = a0 * 4, a4 = a0 * 5, a5 = a0 * 6, a6 = a0 * 7, a7 = a0 * 8;
    double src = seed * 0.5 + threadIdx.x, m = 1e-9 * seed;
    long long t0 = clock64();
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {           // 8 x v_fmac_f64_dpp (independent accumulators)
            asm volatile(REP16(
                "v_fmac_f64_dpp %0, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %2, -%8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %3, -%8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %4, -%8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %5, -%8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %6, -%8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %7, -%8, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(src), "v"(m));
        } else if (MODE == 1) {    // 8 x plain v_fma_f64
            asm volatile(REP16(
                "v_fma_f64 %0, -%8, %9, %0\n\t" "v_fma_f64 %1, -%8, %9, %1\n\t" "v_fma_f64 %2, -%8, %9, %2\n\t" "v_fma_f64 %3, -%8, %9, %3\n\t"
                "v_fma_f64 %4, -%8, %9, %4\n\t" "v_fma_f64 %5, -%8, %9, %5\n\t" "v_fma_f64 %6, -%8, %9, %6\n\t" "v_fma_f64 %7, -%8, %9, %7\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(src), "v"(m));
        } else if (MODE == 2) {    // 8 x (2 v_readlane_b32 + v_fma_f64 with the SGPR pair)
            asm volatile(REP16(
                "v_readlane_b32 s20, %8, 3\n\t v_readlane_b32 s21, %9, 3\n\t v_fma_f64 %0, -s[20:21], %10, %0\n\t"
                "v_readlane_b32 s22, %8, 4\n\t v_readlane_b32 s23, %9, 4\n\t v_fma_f64 %1, -s[22:23], %10, %1\n\t"
                "v_readlane_b32 s24, %8, 5\n\t v_readlane_b32 s25, %9, 5\n\t v_fma_f64 %2, -s[24:25], %10, %2\n\t"
                "v_readlane_b32 s26, %8, 6\n\t v_readlane_b32 s27, %9, 6\n\t v_fma_f64 %3, -s[26:27], %10, %3\n\t"
                "v_readlane_b32 s20, %8, 7\n\t v_readlane_b32 s21, %9, 7\n\t v_fma_f64 %4, -s[20:21], %10, %4\n\t"
                "v_readlane_b32 s22, %8, 8\n\t v_readlane_b32 s23, %9, 8\n\t v_fma_f64 %5, -s[22:23], %10, %5\n\t"
                "v_readlane_b32 s24, %8, 9\n\t v_readlane_b32 s25, %9, 9\n\t v_fma_f64 %6, -s[24:25], %10, %6\n\t"
                "v_readlane_b32 s26, %8, 10\n\t v_readlane_b32 s27, %9, 10\n\t v_fma_f64 %7, -s[26:27], %10, %7\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(__double2loint(src)), "v"(__double2hiint(src)), "v"(m) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (MODE == 3) {    // 8 x v_mov_b64_dpp
            asm volatile(REP16(
                "v_mov_b64_dpp %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_mov_b64_dpp %1, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b64_dpp %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_mov_b64_dpp %3, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b64_dpp %4, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_mov_b64_dpp %5, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b64_dpp %6, %8 row_newbcast:9 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_mov_b64_dpp %7, %8 row_newbcast:10 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(src), "v"(m));
        } else if (MODE == 4) {    // ONE dependent chain of v_fma_f64 (latency)
            asm volatile(REP16(
                "v_fma_f64 %0, -%8, %9, %0\n\t" "v_fma_f64 %0, -%0, %9, %0\n\t" "v_fma_f64 %0, -%0, %9, %0\n\t" "v_fma_f64 %0, -%0, %9, %0\n\t"
                "v_fma_f64 %0, -%0, %9, %0\n\t" "v_fma_f64 %0, -%0, %9, %0\n\t" "v_fma_f64 %0, -%0, %9, %0\n\t" "v_fma_f64 %0, -%0, %9, %0\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(src), "v"(m));
        } else if (MODE == 5) {    // dependent chain of v_rsq_f64
            asm volatile(REP16(
                "v_rsq_f64 %0, %0\n\t s_nop 0\n\t" "v_rsq_f64 %0, %0\n\t s_nop 0\n\t" "v_rsq_f64 %0, %0\n\t s_nop 0\n\t" "v_rsq_f64 %0, %0\n\t s_nop 0\n\t"
                "v_rsq_f64 %0, %0\n\t s_nop 0\n\t" "v_rsq_f64 %0, %0\n\t s_nop 0\n\t" "v_rsq_f64 %0, %0\n\t s_nop 0\n\t" "v_rsq_f64 %0, %0\n\t s_nop 0\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(src), "v"(m));
        } else if (MODE == 6) {    // dependent chain: fmac_dpp -> fmac_dpp (broadcast of the value just written: 2 wait states by hand)
            asm volatile(REP16(
                "v_fmac_f64_dpp %0, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t" "v_fmac_f64_dpp %1, -%0, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t"
                "v_fmac_f64_dpp %0, -%1, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t" "v_fmac_f64_dpp %1, -%0, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t"
                "v_fmac_f64_dpp %0, -%1, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t" "v_fmac_f64_dpp %1, -%0, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t"
                "v_fmac_f64_dpp %0, -%1, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t" "v_fmac_f64_dpp %1, -%0, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t")
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(src), "v"(m));
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double *out; long long *cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const char *names[] = {"v_fmac_f64_dpp (indep)", "v_fma_f64 (indep)", "2 v_readlane + v_fma_f64 sgpr (indep)", "v_mov_b64_dpp (indep)", "v_fma_f64 dependent chain", "v_rsq_f64 dependent chain (+s_nop)", "v_fmac_f64_dpp dependent chain (+s_nop 1)"};
    for (int mode = 0; mode < 7; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, out, cyc, 1.25); break; case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, out, cyc, 1.25); break;
            case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, out, cyc, 1.25); break; case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, out, cyc, 1.25); break;
            case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, out, cyc, 1.25); break; case 5: hipLaunchKernelGGL(k<5>, 1, 64, 0, 0, out, cyc, 1.25); break;
            case 6: hipLaunchKernelGGL(k<6>, 1, 64, 0, 0, out, cyc, 1.25); break;
            }
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double groups = 64.0 * 16 * 8;
        printf("%-44s %8.2f clock64 ticks per group (s_memtime runs at 100 MHz: x %.0f = shader cycles at 2.1 GHz)\n", names[mode], c / groups, 21.0);
    }
    return 0;
}
