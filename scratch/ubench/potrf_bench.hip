// times tile_potrf64 / strip_trsm64 of ba_chol_large.hip in isolation: shader cycles (clock64) per phase and wall time,
// one workgroup alone on the chip and 256 of them at once (is the lone workgroup slow because the chip idles at a low clock?)
#include "../../easysfm_amd/csrc/ba_chol_large.hip"
#include <vector>
#include <cstdio>
namespace esfm { void set_error(const char *, ...) {} }
using namespace esfm;

__global__ __launch_bounds__(256) void bench_p16(const double *A, long long *cyc, int reps, int variant)
{
    __shared__ double T[CB * ULD];
    __shared__ double Vi[4 * SB * VLD];
    __shared__ double rd[CB];
    __shared__ double pbuf[2 * SB];
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long tot = 0, first = 0;
    for (int it = 0; it < reps; ++it) {
        for (int e = tid; e < CB * CB; e += 256) { const int r = e / CB, c = e % CB; T[r * ULD + c] = A[r * CB + c]; }
        __syncthreads();
        long long t0 = clock64();
        if (wave == 0) potrf16_inv(T, ULD, Vi, rd, &fail, lane);
        long long t1 = clock64();
        __syncthreads();
        if (it == 0) first = t1 - t0; else tot += t1 - t0;
    }
    if (tid == 0) { cyc[0] = first; cyc[1] = reps > 1 ? tot / (reps - 1) : 0; }
}

__global__ __launch_bounds__(256) void bench_potrf(const double *A, double *out, long long *cyc, int reps)
{
    __shared__ double T[CB * ULD];
    __shared__ double Vi[4 * SB * VLD];
    __shared__ double rd[CB];
    __shared__ double pbuf[2 * SB];
    __shared__ int fail;
    const int tid = threadIdx.x;
    long long t_load = 0, t_fact = 0;
    for (int it = 0; it < reps; ++it) {
        long long t0 = clock64();
        for (int e = tid; e < CB * CB; e += 256) { const int r = e / CB, c = e % CB; T[r * ULD + c] = (c <= r) ? A[r * CB + c] : 0.0; }
        if (tid == 0) fail = 0;
        __syncthreads();
        long long t1 = clock64();
        {   // tile_potrf64 with a clock per phase
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll 1
            for (int b = 0; b < 4; ++b) {
                long long c0 = clock64();
                if (wave == 0) potrf16_inv(T + (SB * b) * ULD + SB * b, ULD, Vi + b * SB * VLD, rd + SB * b, &fail, lane);
                long long c1 = clock64();
                __syncthreads();
                if (wave > b) {
                    const int i = wave;
                    doublex4 acc = pqt16(doublex4{0.0, 0.0, 0.0, 0.0}, T + (SB * i) * ULD + SB * b, ULD, Vi + b * SB * VLD, VLD, 1.0, lane);
                    __builtin_amdgcn_wave_barrier();
                    store_d16(T + (SB * i) * ULD + SB * b, ULD, acc, lane);
                }
                __syncthreads();
                long long c2 = clock64();
                int idx = 0;
                for (int i = b + 1; i < 4; ++i)
                    for (int j = b + 1; j <= i; ++j, ++idx)
                        if ((idx & 3) == wave) {
                            doublex4 acc = load_d16(T + (SB * i) * ULD + SB * j, ULD, lane);
                            acc = pqt16(acc, T + (SB * i) * ULD + SB * b, ULD, T + (SB * j) * ULD + SB * b, ULD, -1.0, lane);
                            store_d16(T + (SB * i) * ULD + SB * j, ULD, acc, lane);
                        }
                __syncthreads();
                long long c3 = clock64();
                if (tid == 0 && it == reps - 1) { cyc[8 + 3 * b] = c1 - c0; cyc[9 + 3 * b] = c2 - c1; cyc[10 + 3 * b] = c3 - c2; }
            }
        }
        long long t2 = clock64();
        t_load += t1 - t0; t_fact += t2 - t1;
    }
    if (tid == 0) { cyc[2 * blockIdx.x] = t_load / reps; cyc[2 * blockIdx.x + 1] = t_fact / reps; }
    for (int e = tid; e < CB * CB; e += 256) out[(size_t)blockIdx.x * CB * CB + e] = T[(e / CB) * ULD + (e % CB)];
}

int main()
{
    std::vector<double> h(CB * CB), M(CB * CB);
    srand(3);
    for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < CB; ++i) for (int j = 0; j < CB; ++j) { double s = (i == j) ? 4.0 : 0.0; for (int k = 0; k < CB; ++k) s += M[i * CB + k] * M[j * CB + k]; h[i * CB + j] = s; }
    double *A, *out; long long *cyc;
    hipMalloc(&A, h.size() * 8); hipMalloc(&out, 256 * CB * CB * 8); hipMalloc(&cyc, 512 * 8);
    hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int grid : {1, 256}) for (int reps : {1, 20}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        bench_potrf<<<grid, 256>>>(A, out, cyc, reps); hipDeviceSynchronize();
        hipEventRecord(e0);
        bench_potrf<<<grid, 256>>>(A, out, cyc, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c[20]; hipMemcpy(c, cyc, 160, hipMemcpyDeviceToHost);
        printf("grid %3d reps %2d: wall %8.2f us per rep; clock64 ticks per rep: load %lld, tile_potrf64 %lld\n", grid, reps, ms * 1e3 / reps, c[0], c[1]);
        if (grid == 1) for (int b = 0; b < 4; ++b) printf("    sub-block %d: potrf16_inv %lld, panel %lld, trailing %lld\n", b, c[8 + 3 * b], c[9 + 3 * b], c[10 + 3 * b]);
    }
    for (int variant : {0}) {
        bench_p16<<<1, 256>>>(A, cyc, 20, variant); hipDeviceSynchronize();
        long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
        printf("potrf16_inv %s: first call %lld cycles, warm %lld cycles\n", "(registers + v_readlane)", c[0], c[1]);
    }
    // correctness: L L' == A
    std::vector<double> L(CB * CB);
    hipMemcpy(L.data(), out, CB * CB * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < CB; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += L[i * CB + k] * L[j * CB + k]; err = fmax(err, fabs(s - h[i * CB + j])); }
    printf("max |L L' - A| = %.3e\n", err);
    return 0;
}
