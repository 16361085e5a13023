// SURF detector + 64-float descriptor for gfx950 (MI355X): the device side of the replacement for
// cv::xfeatures2d::SURF::create(minHessian)->detect + SURF::create()->compute as FeatureMatching::detectFeaturesSURF calls
// them (reference cpp_code/src/feature_matching.cpp:43-58).  SURVEY.md section 8 row f-2 (SURF half).  The arithmetic follows
// OpenCV's surf.cpp step by step (oracle/surf_ref.c documents the restated rules); every float expression keeps OpenCV's
// operation order (no FMA contraction), sums that OpenCV forms sequentially are formed sequentially by one lane, and the two
// Gaussian tables come from the host's exp(), so the result is bit-identical to the CPU restatement.
//
//   surf_gray_kernel          BGR -> gray, 14-bit fixed point
//   surf_integral_*_kernel    32-bit integral image, one row / column larger than the image (row scan, then column scan)
//   surf_det_trace_kernel     box-filter Hessian determinant and trace of every pyramid layer (one thread per sample)
//   surf_maxima_kernel        3 x 3 x 3 non-maximum suppression + quadratic refinement on the middle layers
//   surf_orient_kernel        one workgroup per keypoint: dominant orientation, start of every row of the rotated window
//   surf_window_kernel        rotated window, one thread per chunk of a window row (all keypoints in one grid)
//   surf_rowsum_kernel        horizontal pass of the area shrink to 21 x 21, one thread per (window row, output column)
//   surf_vector_kernel        one workgroup per keypoint: vertical pass, weighted gradients, 4 x 4 x 4 sums, normalisation
#include "surf_kernels.hpp"

#include <float.h>
#include <math.h>

namespace esfm {

__device__ __forceinline__ int cv_round_f(float v) { return (int)rintf(v); }

__device__ __forceinline__ float calc_haar(const int32_t *__restrict__ origin, const SurfHF *f, int n)
{
    double d = 0.0;
    for (int k = 0; k < n; ++k) d += (origin[f[k].p0] + origin[f[k].p3] - origin[f[k].p1] - origin[f[k].p2]) * f[k].w;
    return (float)d;
}

// cv::fastAtan2 [upstream core/mathfuncs_core]: 7th-order odd polynomial, degrees, 0.3 degree accuracy
__device__ __forceinline__ float fast_atan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846), p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846),
                p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846), p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__global__ __launch_bounds__(256) void surf_gray_kernel(const uint8_t *__restrict__ bgr, int n, uint8_t *__restrict__ gray)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) gray[i] = (uint8_t)((bgr[3 * (size_t)i] * 1868 + bgr[3 * (size_t)i + 1] * 9617 + bgr[3 * (size_t)i + 2] * 4899 + 8192) >> 14);
}

// row pass: sum[(y+1)][x+1] = prefix of row y; one wave per row, 64 pixels per step
__global__ __launch_bounds__(64) void surf_integral_rows_kernel(const uint8_t *__restrict__ gray, int rows, int cols, int32_t *__restrict__ sum)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    const int sc = cols + 1;
    int carry = 0;
    if (lane == 0) sum[(size_t)(y + 1) * sc] = 0;
    for (int x0 = 0; x0 < cols; x0 += 64) {
        const int x = x0 + lane;
        int v = x < cols ? gray[(size_t)y * cols + x] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
        if (x < cols) sum[(size_t)(y + 1) * sc + x + 1] = carry + v;
        carry += __shfl(v, 63);
    }
}

// column pass: running sum down each column (row 0 of the integral image is zero)
__global__ __launch_bounds__(64) void surf_integral_cols_kernel(int rows, int cols, int32_t *__restrict__ sum)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int sc = cols + 1;
    if (x > cols) return;
    int acc = 0;
    sum[x] = 0;
    // the loads do not depend on the running sum: 16 rows are fetched at a time so that their latencies overlap
    int y = 1;
    for (; y + 15 <= rows; y += 16) {
        int v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = sum[(size_t)(y + k) * sc + x];
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc += v[k]; sum[(size_t)(y + k) * sc + x] = acc; }
    }
    for (; y <= rows; ++y) { acc += sum[(size_t)y * sc + x]; sum[(size_t)y * sc + x] = acc; }
}

// grid.y = layer; one thread per written sample of the layer (calcLayerDetAndTrace)
__global__ __launch_bounds__(256) void surf_det_trace_kernel(const SurfParams *__restrict__ P, const int32_t *__restrict__ sum, float *__restrict__ det,
                                                             float *__restrict__ trace)
{
    const SurfLayer &L = P->layer[blockIdx.y];
    if (!L.valid) return;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= L.samples_i * L.samples_j) return;
    const int i = g / L.samples_j, j = g % L.samples_j;
    const int32_t *sp = sum + (size_t)(i * L.step) * (P->cols + 1) + (size_t)j * L.step;
    const float dx = calc_haar(sp, L.dx, 3), dy = calc_haar(sp, L.dy, 3), dxy = calc_haar(sp, L.dxy, 4);
    const size_t o = (size_t)L.offset + (size_t)(i + L.margin) * L.cols + (size_t)(j + L.margin);
    det[o] = dx * dy - 0.81f * dxy * dxy;
    trace[o] = dx + dy;
}

// Matx33f::solve(b, DECOMP_LU): LU with partial pivoting in float
__device__ bool solve3f(float A[3][3], float b[3], float x[3])
{
    for (int i = 0; i < 3; ++i) {
        int k = i;
        for (int j = i + 1; j < 3; ++j) if (fabsf(A[j][i]) > fabsf(A[k][i])) k = j;
        if (fabsf(A[k][i]) < FLT_EPSILON) return false;
        if (k != i) { for (int j = i; j < 3; ++j) { const float t = A[i][j]; A[i][j] = A[k][j]; A[k][j] = t; } const float t = b[i]; b[i] = b[k]; b[k] = t; }
        const float d = -1 / A[i][i];
        for (int j = i + 1; j < 3; ++j) {
            const float alpha = A[j][i] * d;
            for (int c = i + 1; c < 3; ++c) A[j][c] += alpha * A[i][c];
            b[j] += alpha * b[i];
        }
    }
    for (int i = 2; i >= 0; --i) {
        float s = b[i];
        for (int k = i + 1; k < 3; ++k) s -= A[i][k] * x[k];
        x[i] = s / A[i][i];
    }
    return true;
}

// grid.y = middle layer index (octave * 3 + layer - 1); one thread per interior sample (findMaximaInLayer + interpolateKeypoint)
__global__ __launch_bounds__(256) void surf_maxima_kernel(const SurfParams *__restrict__ P, const float *__restrict__ det, const float *__restrict__ trace,
                                                          SurfKeypoint *__restrict__ cand, int32_t *__restrict__ n_cand)
{
    const int oct = blockIdx.y / kSurfOctaveLayers, layer = blockIdx.y % kSurfOctaveLayers + 1;
    const int li = oct * (kSurfOctaveLayers + 2) + layer;
    const SurfLayer &L = P->layer[li], &Lb = P->layer[li - 1], &Lt = P->layer[li + 1];
    const int size = L.size, st = L.step;
    const int layer_rows = P->rows / st, layer_cols = P->cols / st;
    const int margin = (Lt.size / 2) / st + 1;
    const int wi = layer_rows - 2 * margin, wj = layer_cols - 2 * margin;
    if (wi <= 0 || wj <= 0) return;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= wi * wj) return;
    const int i = margin + g / wj, j = margin + g % wj;
    const int stp = L.cols;
    const float val0 = det[(size_t)L.offset + (size_t)i * stp + j];
    if (!(val0 > P->hessian_threshold)) return;
    float N9[3][9];
    const int offs[3] = {Lb.offset, L.offset, Lt.offset};
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const float *d = det + (size_t)offs[l] + (size_t)i * stp + j;
        N9[l][0] = d[-stp - 1]; N9[l][1] = d[-stp]; N9[l][2] = d[-stp + 1]; N9[l][3] = d[-1]; N9[l][4] = d[0]; N9[l][5] = d[1];
        N9[l][6] = d[stp - 1]; N9[l][7] = d[stp]; N9[l][8] = d[stp + 1];
    }
    bool is_max = true;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int q = 0; q < 9; ++q) if (!(l == 1 && q == 4)) is_max = is_max && (val0 > N9[l][q]);
    if (!is_max) return;
    const int sum_i = st * (i - (size / 2) / st), sum_j = st * (j - (size / 2) / st);
    SurfKeypoint kp;
    kp.y = sum_i + (size - 1) * 0.5f; kp.x = sum_j + (size - 1) * 0.5f;
    kp.size = (float)size; kp.angle = -1.f; kp.response = val0; kp.octave = oct; kp.valid = 1;
    const float tr = trace[(size_t)L.offset + (size_t)i * stp + j];
    kp.class_id = (tr > 0) - (tr < 0);
    const int ds = size - Lb.size;
    float b[3] = {-(N9[1][5] - N9[1][3]) / 2, -(N9[1][7] - N9[1][1]) / 2, -(N9[2][4] - N9[0][4]) / 2};
    float A[3][3] = {{N9[1][3] - 2 * N9[1][4] + N9[1][5], (N9[1][8] - N9[1][6] - N9[1][2] + N9[1][0]) / 4, (N9[2][5] - N9[2][3] - N9[0][5] + N9[0][3]) / 4},
                     {(N9[1][8] - N9[1][6] - N9[1][2] + N9[1][0]) / 4, N9[1][1] - 2 * N9[1][4] + N9[1][7], (N9[2][7] - N9[2][1] - N9[0][7] + N9[0][1]) / 4},
                     {(N9[2][5] - N9[2][3] - N9[0][5] + N9[0][3]) / 4, (N9[2][7] - N9[2][1] - N9[0][7] + N9[0][1]) / 4, N9[0][4] - 2 * N9[1][4] + N9[2][4]}};
    float x[3] = {0.f, 0.f, 0.f};
    if (!solve3f(A, b, x)) return;
    const bool ok = (x[0] != 0 || x[1] != 0 || x[2] != 0) && fabsf(x[0]) <= 1 && fabsf(x[1]) <= 1 && fabsf(x[2]) <= 1;
    if (!ok) return;
    kp.x += x[0] * st; kp.y += x[1] * st; kp.size = (float)cv_round_f(kp.size + x[2] * ds);
    const int slot = atomicAdd(n_cand, 1);
    if (slot < P->max_candidates) cand[slot] = kp;
}

// ---- descriptors (SURFInvoker::operator()) in four launches.  A keypoint's window holds up to 633 x 633 samples and the widest
// windows of an image hold most of its samples: with one workgroup per keypoint (rounds 2 - 5) a launch lasted as long as ONE
// compute unit needed for the widest window (fountain image: 607 x 607 samples x ~80 instructions = 0.4 of the 0.53 ms), with the
// rest of the chip idle.  Now the per-sample stages are spread over the chip by ITEMS (a chunk of a window row; a row sum), looked
// up through host-built block tables; the per-keypoint stages (orientation; 21 x 21 patch -> 64 floats) are one narrow
// workgroup each.  Every position / sum that OpenCV forms sequentially is still formed sequentially by one thread, on the same
// operands in the same order: the split changes who computes what, not a single bit.
//   per-keypoint scratch (win_offset[k]):  window bytes | 21 row sums per window row (float) | (x, y) start of every window row (float)
__device__ __forceinline__ float *surf_rowsum_ptr(uint8_t *win, int win_size) { return reinterpret_cast<float *>(win + surf_align16((size_t)win_size * win_size)); }
__device__ __forceinline__ float *surf_start_ptr(uint8_t *win, int win_size)
{
    return reinterpret_cast<float *>(win + surf_align16((size_t)win_size * win_size) + surf_align16(sizeof(float) * (kSurfPatch + 1) * (size_t)win_size));
}

// (1) dominant orientation + the window rows' start positions: one workgroup of 128 per keypoint
constexpr int kSurfOriThreads = 128;
__global__ __launch_bounds__(kSurfOriThreads) void surf_orient_kernel(const SurfParams *__restrict__ P, const SurfDescTables *__restrict__ T,
                                                                      const int32_t *__restrict__ sum, SurfKeypoint *__restrict__ kps,
                                                                      const int64_t *__restrict__ win_offset, uint8_t *__restrict__ win_scratch)
{
    __shared__ float sX[kSurfOriSamples], sY[kSurfOriSamples], sAng[kSurfOriSamples];
    __shared__ float sMod[72], sSumX[72], sSumY[72];
    __shared__ float sDir;
    __shared__ int sN;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SurfKeypoint kp = kps[k];
    const int rows = P->rows, cols = P->cols, sr = rows + 1, sc = cols + 1;
    const float s = kp.size * 1.2f / 9.0f;
    const int grad_wav_size = 2 * cv_round_f(2 * s);
    if (sr < grad_wav_size || sc < grad_wav_size) { if (tid == 0) kps[k].valid = 0; return; }
    SurfHF dx_t[2], dy_t[2];
    {
        const int gx[2][5] = {{0, 0, 2, 4, -1}, {2, 0, 4, 4, 1}}, gy[2][5] = {{0, 0, 4, 2, 1}, {0, 2, 4, 4, -1}};
        const float ratio = (float)grad_wav_size / 4;
        for (int q = 0; q < 2; ++q) {
            int x1 = cv_round_f(ratio * gx[q][0]), y1 = cv_round_f(ratio * gx[q][1]), x2 = cv_round_f(ratio * gx[q][2]), y2 = cv_round_f(ratio * gx[q][3]);
            dx_t[q].p0 = y1 * sc + x1; dx_t[q].p1 = y2 * sc + x1; dx_t[q].p2 = y1 * sc + x2; dx_t[q].p3 = y2 * sc + x2;
            dx_t[q].w = gx[q][4] / ((float)(x2 - x1) * (y2 - y1));
            x1 = cv_round_f(ratio * gy[q][0]); y1 = cv_round_f(ratio * gy[q][1]); x2 = cv_round_f(ratio * gy[q][2]); y2 = cv_round_f(ratio * gy[q][3]);
            dy_t[q].p0 = y1 * sc + x1; dy_t[q].p1 = y2 * sc + x1; dy_t[q].p2 = y1 * sc + x2; dy_t[q].p3 = y2 * sc + x2;
            dy_t[q].w = gy[q][4] / ((float)(x2 - x1) * (y2 - y1));
        }
    }
    int nangle = 0;
    if (wave == 0) {   // one wave: the ballot compaction keeps the sample order, in which the window sums below add
        for (int base = 0; base < T->n_ori; base += 64) {
            const int kk = base + lane;
            bool ok = false;
            float vx = 0.f, vy = 0.f;
            if (kk < T->n_ori) {
                const int x = cv_round_f(kp.x + T->aptx[kk] * s - (float)(grad_wav_size - 1) / 2), y = cv_round_f(kp.y + T->apty[kk] * s - (float)(grad_wav_size - 1) / 2);
                ok = !(y < 0 || y >= sr - grad_wav_size || x < 0 || x >= sc - grad_wav_size);
                if (ok) {
                    const int32_t *ptr = sum + (size_t)y * sc + x;
                    vx = calc_haar(ptr, dx_t, 2) * T->aptw[kk]; vy = calc_haar(ptr, dy_t, 2) * T->aptw[kk];
                }
            }
            const unsigned long long m = __ballot(ok);
            const int pos = nangle + __popcll(m & ((1ull << lane) - 1ull));
            if (ok) { sX[pos] = vx; sY[pos] = vy; sAng[pos] = fast_atan2(vy, vx); }
            nangle += __popcll(m);
        }
        if (lane == 0) sN = nangle;
    }
    __syncthreads();
    nangle = sN;
    if (nangle == 0) { if (tid == 0) kps[k].valid = 0; return; }
    for (int w = tid; w < 72; w += kSurfOriThreads) {
        const int i = 5 * w;
        float sumx = 0.f, sumy = 0.f;
        for (int j = 0; j < nangle; ++j) {
            int d = cv_round_f(sAng[j]) - i; d = d < 0 ? -d : d;
            if (d < 30 || d > 330) { sumx += sX[j]; sumy += sY[j]; }
        }
        sSumX[w] = sumx; sSumY[w] = sumy; sMod[w] = sumx * sumx + sumy * sumy;
    }
    __syncthreads();
    if (tid == 0) {
        float bestx = 0.f, besty = 0.f, best = 0.f;
        for (int w = 0; w < 72; ++w) if (sMod[w] > best) { best = sMod[w]; bestx = sSumX[w]; besty = sSumY[w]; }
        sDir = fast_atan2(-besty, bestx);
        kps[k].angle = sDir;
    }
    __syncthreads();
    // start of every row of the rotated window: OpenCV walks them with float accumulators (start_x += sin, start_y += cos) -- four
    // threads share the walk only in the sense that each replays it up to its own first row
    const int win_size = (int)((kSurfPatch + 1) * s);
    float *start = surf_start_ptr(win_scratch + win_offset[k], win_size);
    const float ddir = sDir * (float)(3.14159265358979323846 / 180);
    const float sin_dir = -(float)sin((double)ddir), cos_dir = (float)cos((double)ddir);
    {
        const int per = (win_size + kSurfOriThreads - 1) / kSurfOriThreads, i0 = tid * per, i1 = i0 + per < win_size ? i0 + per : win_size;
        if (i0 < win_size) {
            const float win_off = -(float)(win_size - 1) / 2;
            float sx = kp.x + win_off * cos_dir + win_off * sin_dir, sy = kp.y - win_off * sin_dir + win_off * cos_dir;
            for (int i = 0; i < i0; ++i) { sx += sin_dir; sy += cos_dir; }
            for (int i = i0; i < i1; ++i, sx += sin_dir, sy += cos_dir) { start[2 * i] = sx; start[2 * i + 1] = sy; }
        }
    }
}

// (2) rotated window of (int)(21 s) pixels, bilinear.  OpenCV walks each row with double accumulators (pixel_x += cos, pixel_y
// -= sin); a row is cut into chunks of about kSurfWinChunk samples and the thread of a chunk first replays the additions up to its
// start, so every position is the same sum of the same terms in the same order.  One thread per chunk, chunks of all keypoints
// in one grid (block table: 256 consecutive chunks of one keypoint).
__global__ __launch_bounds__(256) void surf_window_kernel(const SurfParams *__restrict__ P, const uint8_t *__restrict__ gray,
                                                          const SurfKeypoint *__restrict__ kps, const int64_t *__restrict__ win_offset,
                                                          const SurfBlk *__restrict__ blks, uint8_t *__restrict__ win_scratch)
{
    const SurfBlk bk = blks[blockIdx.x];
    const int k = bk.k, item = bk.first + (int)threadIdx.x;
    const SurfKeypoint kp = kps[k];
    if (!kp.valid) return;
    const int rows = P->rows, cols = P->cols;
    const float s = kp.size * 1.2f / 9.0f;
    const int win_size = (int)((kSurfPatch + 1) * s);
    const int nch = surf_window_chunks(win_size), clen = (win_size + nch - 1) / nch;
    if (item >= win_size * nch) return;
    uint8_t *win = win_scratch + win_offset[k];
    const float *start = surf_start_ptr(win, win_size);
    const float ddir = kp.angle * (float)(3.14159265358979323846 / 180);
    const float sin_dir = -(float)sin((double)ddir), cos_dir = (float)cos((double)ddir);
    const int ncols1 = cols - 1, nrows1 = rows - 1;
    const int i = item / nch, j0 = (item % nch) * clen, j1 = j0 + clen < win_size ? j0 + clen : win_size;
    double pixel_x = start[2 * i], pixel_y = start[2 * i + 1];
    for (int j = 0; j < j0; ++j) { pixel_x += cos_dir; pixel_y -= sin_dir; }
    uint8_t *wrow = win + (size_t)i * win_size;
    // The lanes of a wave sit a chunk apart: every memory instruction touches 64 cache lines, and the kernel is bound by how many of
    // them (and how much f64 position arithmetic) it issues -- so a sample's four taps are two 16-bit loads (any alignment) and
    // four samples leave in one 32-bit store.  Measured and not kept: eight positions per trip with their taps requested together
    // (59 -> 82 us: more instructions, and latency was not what it waited for); chunk-major items, equal replay lengths in a wave
    // (59 -> 107 us: the lanes' stores and taps then lie a whole window row apart).
    auto sample = [&](double px, double py) -> uint32_t {
        const int ix = (int)floor(px), iy = (int)floor(py);
        if ((unsigned)ix < (unsigned)ncols1 && (unsigned)iy < (unsigned)nrows1) {
            const float a = (float)(px - ix), b = (float)(py - iy);
            const uint8_t *p = gray + (size_t)iy * cols + ix;
            uint16_t u0, u1;
            __builtin_memcpy(&u0, p, 2); __builtin_memcpy(&u1, p + cols, 2);
            const int p00 = u0 & 255, p01 = u0 >> 8, p10 = u1 & 255, p11 = u1 >> 8;
            return (uint32_t)(uint8_t)cv_round_f(p00 * (1.f - a) * (1.f - b) + p01 * a * (1.f - b) + p10 * (1.f - a) * b + p11 * a * b);
        }
        int x = (int)rint(px), y = (int)rint(py);
        x = x < 0 ? 0 : (x > ncols1 ? ncols1 : x); y = y < 0 ? 0 : (y > nrows1 ? nrows1 : y);
        return gray[(size_t)y * cols + x];
    };
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
        uint32_t v = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u, pixel_x += cos_dir, pixel_y -= sin_dir) v |= sample(pixel_x, pixel_y) << (8 * u);
        __builtin_memcpy(wrow + j, &v, 4);
    }
    for (; j < j1; ++j, pixel_x += cos_dir, pixel_y -= sin_dir) wrow[j] = (uint8_t)sample(pixel_x, pixel_y);
}

// INTER_AREA shrink of the window to 21 x 21: out = sum_y beta_y (sum_x alpha_x S[y][x]).  cell d of an axis covers the source
// interval [d scale, (d + 1) scale): a fractional first sample, whole samples s1 .. s2 - 1, a fractional last one
struct SurfCell { int s1, s2; float a_first, a_mid, a_last; bool has_first, has_last; };
__device__ __forceinline__ SurfCell surf_area_cell(int dd, int win_size)
{
    const int D = kSurfPatch + 1;
    const double scale = (double)win_size / D;
    const double f1 = dd * scale, f2 = f1 + scale, cell = fmin(scale, win_size - f1);
    SurfCell c;
    c.s1 = (int)ceil(f1); c.s2 = (int)floor(f2);
    c.s2 = c.s2 < win_size - 1 ? c.s2 : win_size - 1; c.s1 = c.s1 < c.s2 ? c.s1 : c.s2;
    c.has_first = c.s1 - f1 > 1e-3; c.has_last = f2 - c.s2 > 1e-3;
    c.a_first = (float)((c.s1 - f1) / cell); c.a_mid = (float)(1.0 / cell); c.a_last = (float)(fmin(fmin(f2 - c.s2, 1.), cell) / cell);
    return c;
}

// (3) the shrink's horizontal pass: one thread per (window row, output column) forms that row's weighted sum sequentially, as
// OpenCV's row pass does (21 of them per window row, whichever output rows use them)
__global__ __launch_bounds__(256) void surf_rowsum_kernel(const SurfKeypoint *__restrict__ kps, const int64_t *__restrict__ win_offset,
                                                          const SurfBlk *__restrict__ blks, uint8_t *__restrict__ win_scratch)
{
    const int D = kSurfPatch + 1;
    const SurfBlk bk = blks[blockIdx.x];
    const int k = bk.k, item = bk.first + (int)threadIdx.x;
    const SurfKeypoint kp = kps[k];
    if (!kp.valid) return;
    const float s = kp.size * 1.2f / 9.0f;
    const int win_size = (int)((kSurfPatch + 1) * s);
    if (item >= win_size * D) return;
    uint8_t *win = win_scratch + win_offset[k];
    const int sy = item / D, dx = item % D;
    const SurfCell c = surf_area_cell(dx, win_size);
    const uint8_t *row = win + (size_t)sy * win_size;
    float buf = 0.f;
    if (c.has_first) buf += row[c.s1 - 1] * c.a_first;
    for (int sx = c.s1; sx < c.s2; ++sx) buf += row[sx] * c.a_mid;
    if (c.has_last) buf += row[c.s2] * c.a_last;
    surf_rowsum_ptr(win, win_size)[item] = buf;
}

// (4) vertical pass -> 21 x 21 patch, weighted gradients, 4 x 4 x 4 sums, normalisation: one workgroup per keypoint
constexpr int kSurfVecThreads = 512;     // 441 output pixels, each a sequential sum over its window rows
__global__ __launch_bounds__(kSurfVecThreads) void surf_vector_kernel(const SurfDescTables *__restrict__ T, const SurfKeypoint *__restrict__ kps,
                                                                      const int64_t *__restrict__ win_offset, uint8_t *__restrict__ win_scratch,
                                                                      float *__restrict__ desc)
{
    __shared__ uint8_t sPatch[kSurfPatch + 1][kSurfPatch + 1];
    __shared__ float sDX[kSurfPatch][kSurfPatch], sDY[kSurfPatch][kSurfPatch];
    __shared__ float sVec[64];
    __shared__ float sInv;
    const int D = kSurfPatch + 1;
    const int k = blockIdx.x, tid = threadIdx.x;
    const SurfKeypoint kp = kps[k];
    if (!kp.valid) return;
    const float s = kp.size * 1.2f / 9.0f;
    const int win_size = (int)((kSurfPatch + 1) * s);
    const float *rsum = surf_rowsum_ptr(win_scratch + win_offset[k], win_size);
    for (int o = tid; o < D * D; o += kSurfVecThreads) {
        const int dy = o / D, dx = o % D;
        const SurfCell c = surf_area_cell(dy, win_size);
        float acc = 0.f;
        if (c.has_first) acc += rsum[(c.s1 - 1) * D + dx] * c.a_first;
        for (int sy = c.s1; sy < c.s2; ++sy) acc += rsum[sy * D + dx] * c.a_mid;
        if (c.has_last) acc += rsum[c.s2 * D + dx] * c.a_last;
        const int v = cv_round_f(acc);
        sPatch[dy][dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    __syncthreads();
    for (int o = tid; o < kSurfPatch * kSurfPatch; o += kSurfVecThreads) {
        const int i = o / kSurfPatch, j = o % kSurfPatch;
        const float dw = T->DW[o];
        sDX[i][j] = (sPatch[i][j + 1] - sPatch[i][j] + sPatch[i + 1][j + 1] - sPatch[i + 1][j]) * dw;
        sDY[i][j] = (sPatch[i + 1][j] - sPatch[i][j] + sPatch[i + 1][j + 1] - sPatch[i][j + 1]) * dw;
    }
    __syncthreads();
    if (tid < 16) {
        const int i = tid / 4, j = tid % 4;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        for (int y = i * 5; y < i * 5 + 5; ++y)
            for (int x = j * 5; x < j * 5 + 5; ++x) { const float tx = sDX[y][x], ty = sDY[y][x]; v0 += tx; v1 += ty; v2 += fabsf(tx); v3 += fabsf(ty); }
        sVec[4 * tid] = v0; sVec[4 * tid + 1] = v1; sVec[4 * tid + 2] = v2; sVec[4 * tid + 3] = v3;
    }
    __syncthreads();
    if (tid == 0) {
        double square_mag = 0.0;
        for (int q = 0; q < 64; ++q) square_mag += sVec[q] * sVec[q];
        sInv = (float)(1. / (sqrt(square_mag) + DBL_EPSILON));
    }
    __syncthreads();
    if (tid < 64) desc[64 * (size_t)k + tid] = sVec[tid] * sInv;
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
#define LAUNCH_OK() ESFM_HIP_TRY(hipGetLastError())

int launch_surf_gray(hipStream_t st, const uint8_t *bgr, int n_pixels, uint8_t *gray)
{
    if (n_pixels <= 0) return ESFM_OK;
    hipLaunchKernelGGL(surf_gray_kernel, dim3((n_pixels + 255) / 256), dim3(256), 0, st, bgr, n_pixels, gray);
    LAUNCH_OK();
    return ESFM_OK;
}

int launch_surf_integral(hipStream_t st, const uint8_t *gray, int rows, int cols, int32_t *sum)
{
    hipLaunchKernelGGL(surf_integral_rows_kernel, dim3(rows), dim3(64), 0, st, gray, rows, cols, sum);
    LAUNCH_OK();
    hipLaunchKernelGGL(surf_integral_cols_kernel, dim3((cols + 1 + 63) / 64), dim3(64), 0, st, rows, cols, sum);   // one wave per CU: more of them stream
    LAUNCH_OK();
    return ESFM_OK;
}

int launch_surf_det_trace(hipStream_t st, const SurfParams *params_dev, const SurfParams &ph, const int32_t *sum, float *det, float *trace,
                          esfm_ctx *timing_ctx)
{
    int max_samples = 1;
    for (int l = 0; l < kSurfLayers; ++l) if (ph.layer[l].valid) max_samples = std::max(max_samples, ph.layer[l].samples_i * ph.layer[l].samples_j);
    KernelTimer tm(timing_ctx, ESFM_K_SURF_DET);
    hipLaunchKernelGGL(surf_det_trace_kernel, dim3((max_samples + 255) / 256, kSurfLayers), dim3(256), 0, st, params_dev, sum, det, trace);
    LAUNCH_OK();
    return ESFM_OK;
}

int launch_surf_maxima(hipStream_t st, const SurfParams *params_dev, const SurfParams &ph, const float *det, const float *trace, SurfKeypoint *cand,
                       int32_t *n_cand)
{
    const int n = std::max(1, ph.rows * ph.cols);
    hipLaunchKernelGGL(surf_maxima_kernel, dim3((n + 255) / 256, kSurfOctaves * kSurfOctaveLayers), dim3(256), 0, st, params_dev, det, trace, cand, n_cand);
    LAUNCH_OK();
    return ESFM_OK;
}

int launch_surf_describe(hipStream_t st, const SurfParams *params_dev, const SurfDescTables *tables_dev, const uint8_t *gray, const int32_t *sum,
                         SurfKeypoint *kps, int n_kp, const int64_t *win_offset, const SurfBlk *win_blocks, int n_win_blocks,
                         const SurfBlk *row_blocks, int n_row_blocks, uint8_t *win_scratch, float *desc, esfm_ctx *timing_ctx)
{
    if (n_kp <= 0) return ESFM_OK;
    KernelTimer tm(timing_ctx, ESFM_K_SURF_DESC);     // (the four launches together)
    hipLaunchKernelGGL(surf_orient_kernel, dim3(n_kp), dim3(kSurfOriThreads), 0, st, params_dev, tables_dev, sum, kps, win_offset, win_scratch);
    if (n_win_blocks > 0)
        hipLaunchKernelGGL(surf_window_kernel, dim3(n_win_blocks), dim3(256), 0, st, params_dev, gray, kps, win_offset, win_blocks, win_scratch);
    if (n_row_blocks > 0)
        hipLaunchKernelGGL(surf_rowsum_kernel, dim3(n_row_blocks), dim3(256), 0, st, kps, win_offset, row_blocks, win_scratch);
    hipLaunchKernelGGL(surf_vector_kernel, dim3(n_kp), dim3(kSurfVecThreads), 0, st, tables_dev, kps, win_offset, win_scratch, desc);
    LAUNCH_OK();
    return ESFM_OK;
}

}  // namespace esfm
