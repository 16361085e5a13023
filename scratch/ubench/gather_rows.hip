// Random 256-byte row gather from a 26-MB table (the re-rank's access pattern in l2_finish_kernel): achieved bytes/s of
//   (a) buffer_load_dwordx4 ... lds (LDS-DMA: 14 instructions of 1 KiB per wave and round, as the kernel does),
//   (b) global_load_dwordx4 into registers (14 per lane in flight),
// at 2 workgroups of 4 waves per CU, one round = 56 rows per wave, wait, next round.
// hipcc -O3 --offload-arch=gfx950 gather_rows.hip -o gather_rows && ./gather_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 rsrc_of(const void *base, uint32_t bytes)
{
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b); r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xFFFFu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes); r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE, int NI, bool STORE = false>
__global__ __launch_bounds__(256) void gather(const float *tab, uint32_t n_rows, int rounds, float *out, uint32_t span_rows)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem) + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 16384u;
    const u32x4 rs = rsrc_of(tab, n_rows * 256u);
    // a workgroup gathers from a window of span_rows rows (one train set: 4096), windows differ per workgroup
    const uint32_t win = (hash32(blockIdx.x) % (n_rows / span_rows)) * span_rows;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t row = win + hash32((blockIdx.x * 4u + wave) * 7919u + r * 131u + i * 4u + (lane >> 4)) % span_rows;
                const int voff = (int)(row * 256u + (lane & 15) * 16u);
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "s"(lds_base + (uint32_t)i * 1024u), "v"(voff), "s"(rs) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += reinterpret_cast<const float *>(smem)[wave * 4096 + lane];
            if (STORE) {      // a round's results: write-through stores (relaxed agent-scope atomics), 4 dwords for 7 lanes of the wave
                if ((lane & 7) == 0 && lane < 56) {
                    int *o = reinterpret_cast<int *>(out) + 16 + (((size_t)blockIdx.x * 4 + wave) * 16 + r % 16) * 64 + lane;
#pragma unroll
                    for (int e = 0; e < 4; ++e) __hip_atomic_store(o + e, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else {
            float4 v[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t row = win + hash32((blockIdx.x * 4u + wave) * 7919u + r * 131u + i * 4u + (lane >> 4)) % span_rows;
                v[i] = *reinterpret_cast<const float4 *>(tab + (size_t)row * 64 + (lane & 15) * 4);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) acc += v[i].x;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main()
{
    const uint32_t n_rows = 25 * 4096;
    float *tab, *out;
    hipMalloc(&tab, (size_t)n_rows * 256); hipMalloc(&out, 64 + (size_t)2048 * 4 * 16 * 64 * 4 + 4096);
    hipMemset(tab, 0, (size_t)n_rows * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int rounds = 16;
    auto run = [&](const char *name, auto kern, int ni, int grid, size_t lds, uint32_t span) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, tab, n_rows, rounds, out, span);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)grid * 4 * rounds * ni * 1024.0;
        printf("%-44s grid %5d span %6u: %.3f ms  %.2f TB/s  (%.1f us per round)\n", name, grid, span, ms, bytes / ms * 1e-9, ms * 1e3 / rounds / (grid / 512.0 > 1 ? grid / 512.0 : 1));
    };
    for (uint32_t span : {4096u, n_rows}) {
        run("lds-dma 14 x 1 KiB per wave-round", gather<0, 14>, 14, 512, 65536, span);
        run("lds-dma 14 x 1 KiB per wave-round", gather<0, 14>, 14, 2048, 65536, span);
        run("lds-dma  7 x 1 KiB per wave-round", gather<0, 7>, 7, 512, 65536, span);
        run("lds-dma 14 x 1 KiB + 28 coherent stores", gather<0, 14, true>, 14, 2048, 65536, span);
        run("global_load_dwordx4 x 14 per lane", gather<1, 14>, 14, 512, 65536, span);
        run("global_load_dwordx4 x 14 per lane", gather<1, 14>, 14, 2048, 65536, span);
        run("global_load_dwordx4 x 14, no LDS (8 wg/CU)", gather<1, 14>, 14, 2048, 0, span);
        run("global_load_dwordx4 x 28, no LDS", gather<1, 28>, 28, 2048, 0, span);
    }
    return 0;
}
