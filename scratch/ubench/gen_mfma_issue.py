#!/usr/bin/env python3
"""Generates scratch/ubench/mfma_issue.hip: exact instruction streams (one asm block per kernel, hard-coded registers) that
answer what the round-1 microbenchmark could not (its loop carried 64 v_mov per 24 MFMAs from a register-role swap):

  * how many VALU / SALU / LDS instructions of which kind hide in the gap behind a v_mfma_f32_32x32x16_bf16,
  * whether it matters that the accumulators live in VGPRs (what -amdgpu-mfma-vgpr-form gives the matcher) or in AGPRs,
  * what the matcher's real fold (v_and_or + 3 v_med3 per element) and the two-level folds (group minimum first) cost.

Stream shape = l2_knn_bf16_kernel's: a phase is 24 MFMAs alternating between two accumulators of the CURRENT set while the
fillers read the PREVIOUS set; the sets swap roles every phase (by unrolling, not by copies).
"""
import sys

VARIANTS = []


def acc(form, s, i):
    base = {"v": 32, "a": 0}[form] + 32 * s + 16 * i
    return f"{form}[{base}:{base + 15}]", base


def gaps_for(kind, form, prev_set, F):
    """24 lists of filler instructions for one phase; fillers read the accumulators of prev_set."""
    g = [[] for _ in range(24)]
    _, p0 = acc(form, prev_set, 0)
    _, p1 = acc(form, prev_set, 1)
    elems = [(0, r) for r in range(16)] + [(1, r) for r in range(16)]     # (chain, reg)
    # order: interleave chains so that the two fold chains alternate like the kernel's (s = 0, 1)
    elems = [e for pair in zip(elems[:16], elems[16:]) for e in pair]
    R = lambda c, r: (p0 if c == 0 else p1) + r

    def spread(instrs_per_unit, units):
        """distribute `units` (lists of instructions) over the 24 gaps as evenly as possible, in order"""
        flat = [i for u in units for i in u]
        n = len(flat)
        k = 0
        for gi in range(24):
            take = (n * (gi + 1)) // 24 - (n * gi) // 24
            g[gi] = flat[k:k + take]
            k += take

    def fold_key(c, src):
        st = 24 + 3 * c
        return [f"v_med3_f32 v{st + 2}, v{st + 1}, v{st + 2}, {src}", f"v_med3_f32 v{st + 1}, v{st}, v{st + 1}, {src}",
                f"v_med3_f32 v{st}, v{st}, {src}, v31"]

    if kind == "none":
        pass
    elif kind == "min_indep":
        for gi in range(24):
            g[gi] = [f"v_min_f32 v{96 + (gi * F + i) % 8}, v{96 + (gi * F + i) % 8}, v{104 + (gi * F + i) % 8}" for i in range(F)]
    elif kind == "med3_indep":
        for gi in range(24):
            g[gi] = [f"v_med3_f32 v{96 + (gi * F + i) % 8}, v{96 + (gi * F + i) % 8}, v{104 + (gi * F + i) % 8}, v{112 + (gi * F + i) % 8}" for i in range(F)]
    elif kind == "min_acc":       # 2-source VALU reading the previous accumulators (register-file port pressure?)
        for gi in range(24):
            g[gi] = [f"v_min_f32 v{96 + (gi * F + i) % 8}, v{96 + (gi * F + i) % 8}, v{R((gi * F + i) & 1, ((gi * F + i) >> 1) % 16)}" for i in range(F)]
    elif kind == "salu":
        for gi in range(24):
            g[gi] = [f"s_or_b32 s{24 + (gi * F + i) % 8}, s{24 + (gi * F + i) % 8}, s{32 + (gi * F + i) % 8}" for i in range(F)]
    elif kind == "nop":
        for gi in range(24):
            g[gi] = ["s_nop 0"] * F
    elif kind == "fold4":          # the shipped fold: 4 VALU per element
        units = []
        for (c, r) in elems:
            if form == "a":
                units.append([f"v_accvgpr_read_b32 v{120 + c}, a{R(c, r)}", f"v_and_or_b32 v{122 + c}, v{120 + c}, v30, s22"] + fold_key(c, f"v{122 + c}"))
            else:
                units.append([f"v_and_or_b32 v{122 + c}, v{R(c, r)}, v30, s22"] + fold_key(c, f"v{122 + c}"))
        spread(None, units)
    elif kind in ("fold_g2", "fold_g4", "fold_g8", "fold_g16"):
        G = int(kind[6:])
        units = []
        for c in (0, 1):
            regs = [R(c, r) for r in range(16)]
            for gidx in range(16 // G):
                grp = regs[gidx * G:(gidx + 1) * G]
                u = []
                if form == "a":
                    for k, rr in enumerate(grp):
                        u.append(f"v_accvgpr_read_b32 v{100 + 16 * c + gidx * G % 16 + k}, a{rr}")
                    grp = [100 + 16 * c + gidx * G % 16 + k for k in range(G)]
                t = f"v{120 + c}"
                rest = list(grp)
                # min3 chain: first instruction takes 3 (or 2) fresh values, every further one 2 more
                if len(rest) >= 3:
                    u.append(f"v_min3_f32 {t}, v{rest[0]}, v{rest[1]}, v{rest[2]}"); rest = rest[3:]
                else:
                    u.append(f"v_min_f32 {t}, v{rest[0]}, v{rest[1]}"); rest = rest[2:]
                while len(rest) >= 2:
                    u.append(f"v_min3_f32 {t}, {t}, v{rest[0]}, v{rest[1]}"); rest = rest[2:]
                if rest:
                    u.append(f"v_min_f32 {t}, {t}, v{rest[0]}")
                u.append(f"v_and_or_b32 v{122 + c}, {t}, v30, s22")
                u += fold_key(c, f"v{122 + c}")
                units.append(u)
        # interleave the two chains' groups
        half = len(units) // 2
        units = [x for pair in zip(units[:half], units[half:]) for x in pair]
        spread(None, units)
    elif kind == "ldsread":        # F ds_read_b128 per 6 gaps (the kernel: 12 per 24) plus nothing else
        for gi in range(24):
            if gi % 2 == 0 and (gi // 2) < F:
                g[gi] = [f"ds_read_b128 v[{100 + 4 * ((gi // 2) % 4)}:{103 + 4 * ((gi // 2) % 4)}], v99 offset:{(gi // 2) * 1024}"]
    else:
        raise ValueError(kind)
    return g


def phase(form, cur, kind, F):
    a0, _ = acc(form, cur, 0)
    a1, _ = acc(form, cur, 1)
    g = gaps_for(kind, form, 1 - cur, F)
    out = []
    for m in range(24):
        A = "v[0:3]" if (m // 2) % 2 == 0 else "v[4:7]"
        B = f"v[{8 + 4 * (m % 4)}:{11 + 4 * (m % 4)}]"
        d = a0 if m % 2 == 0 else a1
        out.append(f"v_mfma_f32_32x32x16_bf16 {d}, {A}, {B}, {d}")
        out += g[m]
    if kind == "ldsread":
        out.append("s_waitcnt lgkmcnt(0)")
    return out


def kernel(name, form, kind, F):
    body = []
    for i in range(24):
        body.append(f"v_mov_b32 v{i}, %{2 + i % 8}")
    for i in range(24, 32):
        body.append(f"v_mov_b32 v{i}, 0x7f000000")
    body.append("v_mov_b32 v30, 0xffffff00")
    body.append("v_mov_b32 v31, 0xff000000")
    for i in range(32, 128):
        body.append(f"v_mov_b32 v{i}, %{2 + i % 8}")
    if form == "a":
        for i in range(64):
            body.append(f"v_accvgpr_write_b32 a{i}, v{32 + i}")
    body.append("v_lshlrev_b32 v99, 4, %10")
    body.append("s_mov_b32 s22, 0x35")
    for i in range(24, 40):
        body.append(f"s_mov_b32 s{i}, {i}")
    body.append("s_mov_b32 s20, %11")
    body.append("s_memtime s[40:41]")
    body.append("s_waitcnt lgkmcnt(0)")
    body.append("L_%=:")
    body += phase(form, 0, kind, F)
    body += phase(form, 1, kind, F)
    body.append("s_sub_u32 s20, s20, 1")
    body.append("s_cmp_lg_u32 s20, 0")
    body.append("s_cbranch_scc1 L_%=")
    body.append("s_nop 15"); body.append("s_nop 15")
    body.append("s_memtime s[42:43]")
    body.append("s_waitcnt lgkmcnt(0)")
    body.append("s_sub_u32 s42, s42, s40")
    body.append("s_subb_u32 s43, s43, s41")
    body.append("v_mov_b32 %0, s42")
    # fold every state register into the output so that nothing is dead (the asm is volatile anyway)
    if form == "a":
        body.append("v_accvgpr_read_b32 v32, a0")
    body.append("v_add_f32 v24, v24, v32")
    body.append("v_add_f32 v24, v24, v96")
    body.append("v_mov_b32 %1, v24")
    clob = [f"v{i}" for i in range(128)] + ([f"a{i}" for i in range(64)] if form == "a" else []) + [f"s{i}" for i in range(20, 44)] + ["scc", "memory"]
    asm = "\n".join(f'        "{l}\\n"' for l in body)
    cl = ", ".join(f'"{c}"' for c in clob)
    return f'''
__global__ __launch_bounds__(256) void {name}(const unsigned *in, unsigned *out, float *sink, int iters)
{{
    __shared__ unsigned lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = in[i & 4095];
    __syncthreads();
    unsigned x[8];
    for (int i = 0; i < 8; ++i) x[i] = in[(threadIdx.x * 8 + i) & 4095];
    unsigned cyc; float res;
    asm volatile(
{asm}
        : "=&v"(cyc), "=&v"(res)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(threadIdx.x & 63), "s"(iters)
        : {cl});
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = cyc;
    sink[blockIdx.x * 256 + threadIdx.x] = res + (float)lds[threadIdx.x];
}}
'''


def main():
    V = []
    for form in ("v", "a"):
        V.append((f"k_{form}_none", form, "none", 0))
        for F in (2, 4, 6, 8):
            V.append((f"k_{form}_min{F}", form, "min_indep", F))
        V.append((f"k_{form}_fold4", form, "fold4", 0))
        V.append((f"k_{form}_foldg4", form, "fold_g4", 0))
    for F in (4, 6):
        V.append((f"k_v_med3i{F}", "v", "med3_indep", F))
        V.append((f"k_v_minacc{F}", "v", "min_acc", F))
    V.append(("k_v_foldg2", "v", "fold_g2", 0))
    V.append(("k_v_foldg8", "v", "fold_g8", 0))
    V.append(("k_v_foldg16", "v", "fold_g16", 0))
    V.append(("k_v_salu2", "v", "salu", 2))
    V.append(("k_v_nop2", "v", "nop", 2))
    V.append(("k_v_lds12", "v", "ldsread", 12))
    src = ['// GENERATED by scratch/ubench/gen_mfma_issue.py -- do not edit', '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>', '#include <vector>', '#include <algorithm>']
    for (n, form, kind, F) in V:
        src.append(kernel(n, form, kind, F))
    src.append('''
typedef void (*kern_t)(const unsigned *, unsigned *, float *, int);
struct Var { const char *name; kern_t fn; int fillers; };
static void run(const Var &v, const unsigned *in, unsigned *out, float *sink, int wps, const char *data)
{
    const int iters = 1500, grid = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, in, out, sink, 1500);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s FAILED\\n", v.name); fflush(stdout); exit(1); }
    hipEventRecord(e0);
    hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, in, out, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(grid * 4);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double ticks = h[h.size() / 2];
    const double mf = double(iters) * 48;
    printf("%-7s %-14s fillers/MFMA %5.2f  waves/SIMD %d : %8.3f ms  wall-cycles@2.4GHz/MFMA/SIMD %6.1f   s_memtime ticks per MFMA of a wave %7.2f  (x waves %7.2f)  -> %7.1f TFLOP/s\\n",
           data, v.name, v.fillers / 48.0, wps, ms, ms * 1e-3 * 2.4e9 / (mf * wps), ticks / mf, ticks / mf / wps,
           mf * wps * 1024.0 * 32768.0 / (ms * 1e-3) / 1e12);
    fflush(stdout);
}
int main()
{
    std::vector<unsigned> h(4096);
    unsigned *in, *out; float *sink;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 8 * 4 * 4); hipMalloc(&sink, 256 * 8 * 256 * 4);
    Var vars[] = {''')
    for (n, form, kind, F) in V:
        # count fillers per two phases
        cnt = sum(len(x) for x in gaps_for(kind, form, 0, F)) * 2
        src.append(f'        {{"{n}", {n}, {cnt}}},')
    src.append('''    };
    for (int pass = 1; pass >= 0; --pass) {
        srand(1);
        for (auto &v : h) {
            if (pass == 0) v = 0x3F803F80u;
            else { unsigned a = 0x3F00u + (rand() & 0xFF) + ((rand() & 1) << 15), b = 0x3F00u + (rand() & 0xFF) + ((rand() & 1) << 15); v = a | (b << 16); }
        }
        hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (const Var &v : vars)
            for (int w = 1; w <= 2; ++w) run(v, in, out, sink, w, pass ? "random" : "const");
    }
    return 0;
}''')
    open(sys.argv[1] if len(sys.argv) > 1 else "mfma_issue.hip", "w").write("\n".join(src))


if __name__ == "__main__":
    main()
