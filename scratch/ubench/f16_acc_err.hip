// How does v_mfma_f32_32x32x16_bf16 accumulate?  Measures, for the exact MFMA sequence of l2_knn_bf16_kernel (|t|^2 in C,
// 12 MFMAs over the hi/lo split of 64-float rows), the error of the result against an f64 evaluation of the SAME split
// products (accumulation error only) and against the true |t|^2 - 2 q.t (total error), in units of
// u * (|t|^2 + 2 sum |q_i t_i|), u = 2^-24.  Build: hipcc -O3 --offload-arch=gfx950 bf16_acc_err.hip -o bf16_acc_err
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint32_t bf16_rne_bits(float a) { const _Float16 h = (_Float16)a; return (uint32_t)__builtin_bit_cast(unsigned short, h); }   // f16 here
__host__ __device__ inline float bf16_to_f(uint32_t b) { return (float)__builtin_bit_cast(_Float16, (unsigned short)b); }

// one wave: A = 32 train rows, B = 32 query rows (already scaled by -2), C = norms; out[t][q]
__global__ void tile_kernel(const float *T, const float *Q, const float *norms, float *out)
{
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = norms[(r & 3) + 8 * (r >> 2) + 4 * h];
    for (int ks = 0; ks < 4; ++ks) {
        uint32_t ah[4], al[4], bh[4], bl[4];
        for (int e = 0; e < 4; ++e) {
            float a0 = T[j * 64 + ks * 16 + 8 * h + 2 * e], a1 = T[j * 64 + ks * 16 + 8 * h + 2 * e + 1];
            float b0 = Q[j * 64 + ks * 16 + 8 * h + 2 * e], b1 = Q[j * 64 + ks * 16 + 8 * h + 2 * e + 1];
            uint32_t x0 = bf16_rne_bits(a0), x1 = bf16_rne_bits(a1), y0 = bf16_rne_bits(b0), y1 = bf16_rne_bits(b1);
            ah[e] = x0 | (x1 << 16); bh[e] = y0 | (y1 << 16);
            al[e] = bf16_rne_bits(a0 - bf16_to_f(x0)) | (bf16_rne_bits(a1 - bf16_to_f(x1)) << 16);
            bl[e] = bf16_rne_bits(b0 - bf16_to_f(y0)) | (bf16_rne_bits(b1 - bf16_to_f(y1)) << 16);
        }
        const f16x8 ahi = __builtin_bit_cast(f16x8, (u32x4{ah[0], ah[1], ah[2], ah[3]})), alo = __builtin_bit_cast(f16x8, (u32x4{al[0], al[1], al[2], al[3]}));
        const f16x8 bhi = __builtin_bit_cast(f16x8, (u32x4{bh[0], bh[1], bh[2], bh[3]}));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}

int main()
{
    std::mt19937_64 rng(7);
    std::normal_distribution<float> N(0.f, 1.f);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    float *dT, *dQ, *dN, *dO;
    hipMalloc(&dT, 32 * 64 * 4); hipMalloc(&dQ, 32 * 64 * 4); hipMalloc(&dN, 32 * 4); hipMalloc(&dO, 32 * 32 * 4);
    const char *names[] = {"unit random", "all positive", "wide dynamic range", "near-duplicate rows (cancellation)", "large norms 1e3", "sparse rows"};
    for (int mode = 0; mode < 6; ++mode) {
        double worst_acc = 0, worst_tot = 0, worst_split = 0;
        for (int rep = 0; rep < 400; ++rep) {
            std::vector<float> T(32 * 64), Q(32 * 64), Nn(32), O(32 * 32);
            for (int r = 0; r < 32; ++r) {
                double nt = 0, nq = 0;
                for (int k = 0; k < 64; ++k) {
                    float a = N(rng), b = N(rng);
                    if (mode == 1) { a = std::fabs(a); b = -std::fabs(b); }
                    if (mode == 2) { a *= std::exp(6 * N(rng)); b *= std::exp(6 * N(rng)); }
                    if (mode == 5) { if (U(rng) < 0.8f) a = 0; if (U(rng) < 0.8f) b = 0; }
                    T[r * 64 + k] = a; Q[r * 64 + k] = b; nt += (double)a * a; nq += (double)b * b;
                }
                const float st = mode == 4 ? 1e3f : 1.f;
                for (int k = 0; k < 64; ++k) { T[r * 64 + k] *= st / (float)std::sqrt(nt); Q[r * 64 + k] *= st / (float)std::sqrt(nq); }
            }
            if (mode == 3) for (int r = 0; r < 32; ++r) for (int k = 0; k < 64; ++k) Q[r * 64 + k] = T[((r * 7) % 32) * 64 + k] * (1.f + 1e-3f * N(rng));
            for (int r = 0; r < 32; ++r) { float s = 0; for (int k = 0; k < 64; ++k) s = fmaf(T[r * 64 + k], T[r * 64 + k], s); Nn[r] = s; }
            std::vector<float> Q2(Q); for (auto &v : Q2) v *= -2.f;
            hipMemcpy(dT, T.data(), T.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dQ, Q2.data(), Q2.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dN, Nn.data(), 32 * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(tile_kernel, dim3(1), dim3(64), 0, 0, dT, dQ, dN, dO);
            hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
            for (int t = 0; t < 32; ++t) for (int q = 0; q < 32; ++q) {
                double split = Nn[t], truth = Nn[t], mag = std::fabs((double)Nn[t]);
                for (int k = 0; k < 64; ++k) {
                    const float a = T[t * 64 + k], b = Q2[q * 64 + k];
                    const double ah = bf16_to_f(bf16_rne_bits(a)), al = bf16_to_f(bf16_rne_bits(a - (float)ah));
                    const double bh = bf16_to_f(bf16_rne_bits(b)), bl = bf16_to_f(bf16_rne_bits(b - (float)bh));
                    split += al * bh + ah * bh; truth += (double)a * b; mag += std::fabs((double)a * b);
                }
                const double u = std::ldexp(1.0, -24);
                worst_acc = std::max(worst_acc, std::fabs(O[t * 32 + q] - split) / (u * mag));
                worst_split = std::max(worst_split, std::fabs(split - truth) / (u * mag));
                worst_tot = std::max(worst_tot, std::fabs(O[t * 32 + q] - truth) / (u * mag));
            }
        }
        std::printf("%-36s accumulation err %.2f u*M   split err %.2f u*M   total %.2f u*M   (bound used: 2^-13 of |q|^2+|t|^2 ~ %.0f u*M for unit rows)\n",
                    names[mode], worst_acc, worst_split, worst_tot, 2048.0 * 2 / 3);
    }
    return 0;
}
