// What does the layout of the stored Jacobians cost?  BA-512's kernels stream 18 + 2 arrays of doubles (SoA, one observation per lane:
// 8 B per lane and load) at 2.4 TB/s.  Same bytes, same one-observation-per-lane mapping, three layouts:
//   A  18 arrays of double        J[a * n + k]
//   B   9 arrays of double2       J2[a * n + k]                  (16 B per lane and load)
//   C   tiles of 64 observations: T[(k / 64) * 18 * 64 + a * 64 + (k % 64)]   (a wave's 18 loads fall into ONE 9 KiB stretch)
//   D  tiles of 64 observations x double2: T2[(k / 64) * 9 * 64 + a * 64 + (k % 64)]
// read (sum into one double per lane, stored) and write (the same values back out) legs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, bool WRITE>
__global__ __launch_bounds__(256) void stream_kernel(const double *__restrict__ in, double *__restrict__ out, double *__restrict__ sink, size_t n)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double v[18];
    if (MODE == 0) {
#pragma unroll
        for (int a = 0; a < 18; ++a) v[a] = in[a * n + k];
    } else if (MODE == 1) {
        const double2 *in2 = reinterpret_cast<const double2 *>(in);
#pragma unroll
        for (int a = 0; a < 9; ++a) { const double2 t = in2[a * n + k]; v[2 * a] = t.x; v[2 * a + 1] = t.y; }
    } else if (MODE == 2) {
        const double *t = in + (k >> 6) * (18 * 64) + (k & 63);
#pragma unroll
        for (int a = 0; a < 18; ++a) v[a] = t[a * 64];
    } else {
        const double2 *t = reinterpret_cast<const double2 *>(in) + (k >> 6) * (9 * 64) + (k & 63);
#pragma unroll
        for (int a = 0; a < 9; ++a) { const double2 u = t[a * 64]; v[2 * a] = u.x; v[2 * a + 1] = u.y; }
    }
    if (WRITE) {
        if (MODE == 0) {
#pragma unroll
            for (int a = 0; a < 18; ++a) out[a * n + k] = v[a] * 1.5;
        } else if (MODE == 1) {
            double2 *o2 = reinterpret_cast<double2 *>(out);
#pragma unroll
            for (int a = 0; a < 9; ++a) o2[a * n + k] = make_double2(v[2 * a] * 1.5, v[2 * a + 1] * 1.5);
        } else if (MODE == 2) {
            double *t = out + (k >> 6) * (18 * 64) + (k & 63);
#pragma unroll
            for (int a = 0; a < 18; ++a) t[a * 64] = v[a] * 1.5;
        } else {
            double2 *t = reinterpret_cast<double2 *>(out) + (k >> 6) * (9 * 64) + (k & 63);
#pragma unroll
            for (int a = 0; a < 9; ++a) t[a * 64] = make_double2(v[2 * a] * 1.5, v[2 * a + 1] * 1.5);
        }
    } else {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 18; ++a) s += v[a];
        sink[k] = s;
    }
}

template <int MODE, bool WRITE>
static void run(const char *name, const double *in, double *out, double *sink, size_t n)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = (int)((n + 255) / 256);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((stream_kernel<MODE, WRITE>), dim3(grid), dim3(256), 0, 0, in, out, sink, n);
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((stream_kernel<MODE, WRITE>), dim3(grid), dim3(256), 0, 0, in, out, sink, n);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    const double bytes = (double)n * 8.0 * (WRITE ? 36.0 : 19.0);
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, us, bytes / us * 1e-6);
}

int main()
{
    const size_t n = 3000000 / 64 * 64;
    double *in, *out, *sink;
    CHECK(hipMalloc(&in, n * 18 * 8)); CHECK(hipMalloc(&out, n * 18 * 8)); CHECK(hipMalloc(&sink, n * 8));
    CHECK(hipMemset(in, 0, n * 18 * 8));
    printf("n = %zu observations, 18 doubles each (%.0f MB)\n", n, n * 144e-6);
    run<0, false>("A read  18 x double SoA", in, out, sink, n);
    run<1, false>("B read   9 x double2 SoA", in, out, sink, n);
    run<2, false>("C read  tiles of 64 x 18 double", in, out, sink, n);
    run<3, false>("D read  tiles of 64 x 9 double2", in, out, sink, n);
    run<0, true>("A copy  18 x double SoA", in, out, sink, n);
    run<1, true>("B copy   9 x double2 SoA", in, out, sink, n);
    run<2, true>("C copy  tiles of 64 x 18 double", in, out, sink, n);
    run<3, true>("D copy  tiles of 64 x 9 double2", in, out, sink, n);
    return 0;
}
