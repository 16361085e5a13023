// Dense solve of the reduced camera system when it does not fit one workgroup's LDS (n = 6 n_cam > ~185;
// BASELINE config 5: 512 cameras, n = 3072).  Blocked right-looking Cholesky on 64 x 64 f64 tiles across the
// whole chip, the right-hand side carried as an extra block row (forward substitution for free), then a
// blocked backward substitution.  This is DENSE_SCHUR's factorisation step (reference cpp_code/src/ba.cpp:201,
// Ceres' Eigen LLT [upstream]) for large camera counts.
//
//   chol_assemble_kernel   W = F'F + D_c^2 + S_schur (lower), rhs row = F'r + rhs_corr, identity padding
//   chol_panel_kernel(k)   every workgroup factors the diagonal tile (k,k) redundantly in LDS, then solves its
//                          own 64-row tile (i,k) against it; workgroup 0 stores L_kk
//   chol_update_kernel(k)  trailing update C_ij -= A_ik A_jk' for k < j <= i (rhs block row included)
//   chol_back_kernel(k)    y_k = L_kk^-T z_k (redundantly per workgroup), z_b -= L_kb' y_k for b < k
#include "ba_kernels.hpp"

namespace esfm {

constexpr int CB = 64;          // tile edge
constexpr int CLD = CB + 1;     // LDS leading dimension (f64, odd: conflict-free column access)

__global__ __launch_bounds__(256) void chol_assemble_kernel(BADev d, double *__restrict__ W, int ld, int nb, double radius,
                                                            double min_diag, double max_diag)
{
    const int n = 6 * d.n_cam;
    const long long rows = (long long)(nb + 1) * CB;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * ld) return;
    const int i = (int)(e / ld), j = (int)(e % ld);
    double v = 0.0;
    if (i < n) {
        if (j <= i) {
            v = d.red[(size_t)i * n + j];
            if (i / 6 == j / 6) {
                const int c = i / 6;
                v += d.camacc[36 * (size_t)c + 6 * (i % 6) + (j % 6)];
                if (i == j) v += fmin(fmax(d.camacc[36 * (size_t)c + 7 * (i % 6)], min_diag), max_diag) / radius;
            }
        }
    } else if (i < nb * CB) {
        v = (i == j) ? 1.0 : 0.0;  // padding rows: identity
    } else if (i == nb * CB) {
        v = (j < n) ? d.camacc[36 * (size_t)d.n_cam + j] + d.red[(size_t)n * n + j] : 0.0;  // rhs row
    }
    W[(size_t)i * ld + j] = v;
}

// In-LDS Cholesky of a 64 x 64 tile (lower), 256 threads, blocked by 8 columns (2 barriers + 1 per panel
// instead of 3 per column): (A) all threads subtract the already-factored columns from the panel by dot
// products, (B1) wave 0 factors the 8 x 8 diagonal block in registers with __shfl, (B2) the rows below solve
// against it.  rd[] receives the reciprocal diagonal.  *fail is raised on a non-positive pivot.
constexpr int FB = 8;
// value of lane `l` (wave-uniform index): two v_readlane_b32 instead of the LDS round trip of __shfl
__device__ __forceinline__ double lane_value_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// rows: CB for the diagonal tile alone, 2*CB when a panel tile is stacked below it (rows CB..2CB-1 then come out as
// X = A L^-T, the triangular solve, at no extra barriers).
__device__ __forceinline__ void factor_tile_lds(double *L, double *rd, volatile int *fail, int rows)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    for (int j0 = 0; j0 < CB; j0 += FB) {
        if (j0 > 0) {
            for (int e = tid; e < (rows - j0) * FB; e += 256) {
                const int i = j0 + e / FB, col = j0 + e % FB;
                if (col > i) continue;
                const double *Li = L + i * CLD, *Lc = L + col * CLD;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;   // four chains: the dot product is latency-bound
                for (int k = 0; k < j0; k += 4) {                // j0 is a multiple of 8
                    s0 += Li[k] * Lc[k]; s1 += Li[k + 1] * Lc[k + 1]; s2 += Li[k + 2] * Lc[k + 2]; s3 += Li[k + 3] * Lc[k + 3];
                }
                L[i * CLD + col] -= (s0 + s1) + (s2 + s3);
            }
        }
        __syncthreads();
        if (wave == 0) {
            const int r = lane;
            double a[FB];
#pragma unroll
            for (int c = 0; c < FB; ++c) a[c] = (r < FB && c <= r) ? L[(j0 + r) * CLD + j0 + c] : 0.0;
#pragma unroll
            for (int c = 0; c < FB; ++c) {
                const double piv = lane_value_f64(a[c], c);
                if (!(piv > 0.0) || !isfinite(piv)) { if (lane == 0) *fail = 1; }
                const double rinv = rsqrt(piv > 0.0 ? piv : 1.0);
                a[c] = (r == c) ? piv * rinv : a[c] * rinv;
                if (lane == c) rd[j0 + c] = rinv;
#pragma unroll
                for (int c2 = c + 1; c2 < FB; ++c2) {
                    const double l2 = lane_value_f64(a[c], c2);
                    if (r >= c2) a[c2] -= a[c] * l2;
                }
            }
#pragma unroll
            for (int c = 0; c < FB; ++c)
                if (r < FB && c <= r) L[(j0 + r) * CLD + j0 + c] = a[c];
        }
        __syncthreads();
        for (int i = j0 + FB + tid; i < rows; i += 256) {
            double *Li = L + i * CLD;
            double x[FB];
#pragma unroll
            for (int c = 0; c < FB; ++c) x[c] = Li[j0 + c];
#pragma unroll
            for (int c = 0; c < FB; ++c) {
                const double *Lc = L + (j0 + c) * CLD + j0;
                double v = x[c];
#pragma unroll
                for (int c1 = 0; c1 < c; ++c1) v -= x[c1] * Lc[c1];
                x[c] = v * rd[j0 + c];
                Li[j0 + c] = x[c];
            }
        }
        __syncthreads();
    }
}

// Ldiag: factored diagonal tiles, kept OUT of W: other workgroups of the same launch still read the unfactored (k,k) tile
__global__ __launch_bounds__(256) void chol_panel_kernel(double *__restrict__ W, double *__restrict__ Ldiag, int ld, int nb, int k,
                                                         double *__restrict__ scal)
{
    __shared__ double T[2 * CB * CLD];   // rows 0..63: diagonal tile (k,k); rows 64..127: this workgroup's tile (i,k)
    __shared__ double rd[CB];
    __shared__ int fail;
    const int tid = threadIdx.x;
    const int bi = k + blockIdx.x;  // block row handled by this workgroup (k .. nb, nb = rhs block)
    if (tid == 0) fail = 0;
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c = e % CB;
        T[r * CLD + c] = (c <= r) ? W[(size_t)(k * CB + r) * ld + k * CB + c] : 0.0;
        if (bi != k) T[(CB + r) * CLD + c] = W[(size_t)(bi * CB + r) * ld + k * CB + c];
    }
    factor_tile_lds(T, rd, &fail, bi == k ? CB : 2 * CB);
    if (bi == k) {
        for (int e = tid; e < CB * CB; e += 256) {
            const int r = e / CB, c = e % CB;
            Ldiag[(size_t)k * (CB * CB + CB) + e] = (c <= r) ? T[r * CLD + c] : 0.0;
        }
        if (tid < CB) Ldiag[(size_t)k * (CB * CB + CB) + CB * CB + tid] = rd[tid];   // reciprocal diagonal for the back-substitution
        if (tid == 0 && fail) scal[SC_CHOL_FAIL] = 1.0;
        return;
    }
    for (int e = tid; e < CB * CB; e += 256) W[(size_t)(bi * CB + e / CB) * ld + k * CB + (e % CB)] = T[(CB + e / CB) * CLD + (e % CB)];
}

// C_ij -= A_ik A_jk'  for the tiles k < j <= i <= nb (j <= nb-1).  256 threads, 4 x 4 outputs per thread.
__global__ __launch_bounds__(256) void chol_update_kernel(double *__restrict__ W, int ld, int nb, int k)
{
    __shared__ double Ai[CB * CLD];
    __shared__ double Aj[CB * CLD];
    // linear tile id -> (i, j): tiles of block row i (k+1 .. nb) are j = k+1 .. min(i, nb-1)
    const int m = nb - k - 1;  // square trailing block rows
    int t = blockIdx.x, i, j;
    const int tri = m * (m + 1) / 2;
    if (t < tri) {
        int ri = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((ri + 1) * (ri + 2) / 2 <= t) ++ri;
        while (ri * (ri + 1) / 2 > t) --ri;
        i = k + 1 + ri; j = k + 1 + (t - ri * (ri + 1) / 2);
    } else {
        i = nb; j = k + 1 + (t - tri);
    }
    const int tid = threadIdx.x;
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c = e % CB;
        Ai[r * CLD + c] = W[(size_t)(i * CB + r) * ld + k * CB + c];
        Aj[r * CLD + c] = W[(size_t)(j * CB + r) * ld + k * CB + c];
    }
    __syncthreads();
    const int tr = (tid / 16) * 4, tc = (tid % 16) * 4;
    double acc[4][4] = {{0}};
    for (int kk = 0; kk < CB; ++kk) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = Ai[(tr + u) * CLD + kk]; b[u] = Aj[(tc + u) * CLD + kk]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = tr + u, c = tc + v;
            if (i != j || c <= r) W[(size_t)(i * CB + r) * ld + j * CB + c] -= acc[u][v];
        }
}

// Backward substitution step k: z = rhs row (row nb*CB of W).  Every workgroup solves L_kk' y_k = z_k (one wave),
// workgroup b < k then applies z_b -= L_kb' y_k; workgroup k stores y_k.
__global__ __launch_bounds__(256) void chol_back_kernel(double *__restrict__ W, const double *__restrict__ Ldiag, int ld, int nb, int k)
{
    __shared__ double Lkk[CB * CLD];
    __shared__ double y[CB];
    __shared__ double rd[CB];
    const int tid = threadIdx.x;
    double *z = W + (size_t)nb * CB * ld;
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c = e % CB;
        Lkk[r * CLD + c] = Ldiag[(size_t)k * (CB * CB + CB) + e];
    }
    if (tid < CB) rd[tid] = Ldiag[(size_t)k * (CB * CB + CB) + CB * CB + tid];
    __syncthreads();
    if (tid < CB) {  // one wave; lane = row index, its y value lives in a register
        double yl = z[k * CB + tid];
        for (int c = CB - 1; c >= 0; --c) {
            const double yc = __shfl(yl, c) * rd[c];
            if (tid == c) yl = yc;
            if (tid < c) yl -= Lkk[c * CLD + tid] * yc;
        }
        y[tid] = yl;
    }
    __syncthreads();
    const int b = blockIdx.x;
    if (b == k) { if (tid < CB) z[k * CB + tid] = y[tid]; return; }
    // z_b[t] -= sum_r L[k*CB + r][b*CB + t] * y[r]; 4 row-groups per column, reduced through LDS
    __shared__ double part[4][CB];
    const int t = tid & 63, g = tid >> 6;
    double s = 0.0;
    for (int r = g; r < CB; r += 4) s += W[(size_t)(k * CB + r) * ld + b * CB + t] * y[r];
    part[g][t] = s;
    __syncthreads();
    if (tid < CB) z[b * CB + tid] -= part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
}

__global__ void chol_extract_kernel(BADev d, const double *__restrict__ W, int ld, int nb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = 6 * d.n_cam;
    if (i >= n) return;
    const bool fail = d.scal[SC_CHOL_FAIL] != 0.0;
    d.y_c[i] = fail ? 0.0 : W[(size_t)nb * CB * ld + i];
}

size_t ba_chol_large_doubles(int n_cam)
{
    const int n = 6 * n_cam, nb = (n + CB - 1) / CB;
    return (size_t)(nb + 1) * CB * (size_t)(nb * CB) + (size_t)nb * (CB * CB + CB);
}

int ba_solve_reduced_large(hipStream_t st, const BADev &d, double radius, double min_diag, double max_diag)
{
    const int n = 6 * d.n_cam, nb = (n + CB - 1) / CB, ld = nb * CB;
    double *W = d.chol;
    double *Ldiag = W + (size_t)(nb + 1) * CB * ld;
    const long long tot = (long long)(nb + 1) * CB * ld;
    hipLaunchKernelGGL(chol_assemble_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d, W, ld, nb, radius, min_diag, max_diag);
    ESFM_HIP_TRY(hipGetLastError());
    for (int k = 0; k < nb; ++k) {
        hipLaunchKernelGGL(chol_panel_kernel, dim3(nb - k + 1), dim3(256), 0, st, W, Ldiag, ld, nb, k, d.scal);
        const int m = nb - k - 1;
        const int tiles = m * (m + 1) / 2 + m;
        if (tiles > 0) hipLaunchKernelGGL(chol_update_kernel, dim3(tiles), dim3(256), 0, st, W, ld, nb, k);
    }
    ESFM_HIP_TRY(hipGetLastError());
    for (int k = nb - 1; k >= 0; --k) hipLaunchKernelGGL(chol_back_kernel, dim3(k + 1), dim3(256), 0, st, W, Ldiag, ld, nb, k);
    hipLaunchKernelGGL(chol_extract_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d, W, ld, nb);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm
