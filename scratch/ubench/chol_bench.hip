// Standalone harness for ba_chol_solve_kernel: random SPD reduced system of 25 cameras, per-phase cycle counts.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -DESFM_CHOL_PROFILE -I easysfm_amd/csrc \
//   scratch/ubench/chol_bench.hip easysfm_amd/csrc/ctx.cpp -o gpurun_out/chol_bench
#include "../../easysfm_amd/csrc/ba_kernels.hip"
#include "../../easysfm_amd/csrc/ba_chol_large.hip"
#include <random>
#include <vector>
using namespace esfm;
int main(int argc, char **argv)
{
    const int nc = argc > 1 ? atoi(argv[1]) : 25;
    const int n = 6 * nc;
    BADev d; d.n_cam = nc; d.n_real_cam = nc;
    std::mt19937 rng(1);
    std::normal_distribution<double> g(0, 1);
    std::vector<double> B((size_t)n * n), S((size_t)n * n + n), cam(42 * (size_t)nc, 0.0);
    for (auto &x : B) x = g(rng);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += B[(size_t)i * n + k] * B[(size_t)j * n + k]; S[(size_t)i * n + j] = s + (i == j ? n : 0); }
    for (int i = 0; i < n; ++i) S[(size_t)n * n + i] = g(rng);
    for (int c = 0; c < nc; ++c) for (int a = 0; a < 6; ++a) cam[36 * (size_t)c + 7 * a] = 1.0;
    hipMalloc(&d.red, sizeof(double) * S.size()); hipMemcpy(d.red, S.data(), sizeof(double) * S.size(), hipMemcpyHostToDevice);
    hipMalloc(&d.camacc, sizeof(double) * cam.size()); hipMemcpy(d.camacc, cam.data(), sizeof(double) * cam.size(), hipMemcpyHostToDevice);
    hipMalloc(&d.y_c, sizeof(double) * n); hipMalloc(&d.scal, sizeof(double) * SC_COUNT); hipMemset(d.scal, 0, sizeof(double) * SC_COUNT);
    hipMalloc(&d.chol, sizeof(double) * ((size_t)(n + 1) * (n + 2) / 2 + 1024)); hipMemset(d.chol, 0, sizeof(double) * 1024);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) ba_solve_reduced(st, d, 1e4, 1e-6, 1e32);
    hipStreamSynchronize(st);
    hipMemset(d.chol, 0, sizeof(double) * 1024);
    hipEventRecord(e0, st);
    const int reps = 20;
    for (int it = 0; it < reps; ++it) ba_solve_reduced(st, d, 1e4, 1e-6, 1e32);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> y(n), prof(16);
    hipMemcpy(y.data(), d.y_c, sizeof(double) * n, hipMemcpyDeviceToHost);
    hipMemcpy(prof.data(), d.chol, sizeof(double) * 16, hipMemcpyDeviceToHost);
    // residual check
    double worst = 0;
    for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += (S[(size_t)i * n + j] + (i == j ? 1.0 + 1.0 / 1e4 : 0.0)) * y[j]; worst = std::max(worst, std::fabs(s - S[(size_t)n * n + i])); }
    printf("n=%d  %.2f us per solve (back-to-back)  residual %.2e\n", n, ms * 1e3 / reps, worst);
    const char *names[] = {"assemble", "B1", "B2", "C", "backward", "store"};
    for (int i = 0; i < 6; ++i) printf("  %-9s %8.0f cycles/solve (%.1f us @100MHz memtime)\n", names[i], prof[i] / reps, prof[i] / reps / 100.0);
    return 0;
}
