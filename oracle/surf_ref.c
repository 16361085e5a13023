/*
 * oracle/surf_ref.c -- CPU restatement of the SURF detector + 64-float descriptor the reference extracts for every image
 * (cv::xfeatures2d::SURF::create(minHessian)->detect + SURF::create()->compute, reference
 * cpp_code/src/feature_matching.cpp:43-58).  SURVEY.md section 8 row f-2 (SURF half).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link, call or execute this file
 * (see oracle/match_ref.c header).
 *
 * PARITY UNPINNED: opencv_contrib's xfeatures2d (non-free SURF) is absent here and the reference holds no fixture.  Restated
 * from memory of OpenCV 3.4 [upstream opencv_contrib/modules/xfeatures2d/src/surf.cpp, modules/imgproc color / resize /
 * smooth, modules/core mathfuncs], defaults nOctaves = 4, nOctaveLayers = 3, extended = false, upright = false:
 *   gray        (B*1868 + G*9617 + R*4899 + 8192) >> 14   (cvtColor BGR2GRAY, 14-bit fixed point)
 *   integral    32-bit sums, one row and column larger than the image
 *   layers      size = (9 + 6 layer) << octave, sample step = 1 << octave, 5 layers per octave; box filters dx, dy (3 boxes,
 *               weights 1 -2 1) and dxy (4 boxes, 1 -1 -1 1) of the 9 x 9 template scaled by cvRound(size / 9 * coord), each box
 *               normalised by its area; det = dx dy - 0.81 dxy^2, trace = dx + dy (float)
 *   maxima      middle layers only; det > threshold and strictly greater than its 26 neighbours; 3-D quadratic refinement
 *               (A x = b by LU with partial pivoting in float), kept iff x != 0 and |x_i| <= 1; pt += x * step,
 *               size = cvRound(size + x_s * (size - size of the layer below)); response = det, class_id = sign(trace);
 *               keypoints sorted by (response, size, octave, y) descending, then x ascending
 *   orientation s = size * 1.2 / 9; Haar responses of side 2 cvRound(2 s) at the 109 grid points of a radius-6 disc scaled by
 *               s, Gaussian weights (sigma 2.5); angles by cv::fastAtan2 (the 7th-order polynomial, degrees); a 60 degree
 *               window slid in 5 degree steps over the rounded angles, the window with the largest summed vector wins;
 *               angle = fastAtan2(-sum_y, sum_x)
 *   descriptor  a (int)(21 s) square window rotated by the angle, sampled bilinearly (cvRound to uchar; border: nearest
 *               clamped pixel), shrunk to 21 x 21 by area averaging (INTER_AREA, general weighted form), 20 x 20 gradients
 *               (2 x 2 differences) weighted by a Gaussian (sigma 3.3), 4 x 4 cells of 5 x 5 samples with sum dx, sum dy,
 *               sum |dx|, sum |dy|, normalised to unit length.  Keypoints whose orientation cannot be sampled are dropped.
 * Deviation: INTER_AREA's integer fast path (window side an exact multiple of 21) is not special-cased.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

void esfm_ref_bgr2gray(const uint8_t *bgr, int n_pixels, uint8_t *gray)
{
    for (int i = 0; i < n_pixels; ++i) gray[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + 8192) >> 14);
}

static float fast_atan2(float y, float x)
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

typedef struct { int p0, p1, p2, p3; float w; } surf_hf;

static void resize_haar(const int src[][5], surf_hf *dst, int n, int old_size, int new_size, int width_step)
{
    const float ratio = (float)new_size / old_size;
    for (int k = 0; k < n; ++k) {
        const int dx1 = cv_round_f(ratio * src[k][0]), dy1 = cv_round_f(ratio * src[k][1]);
        const int dx2 = cv_round_f(ratio * src[k][2]), dy2 = cv_round_f(ratio * src[k][3]);
        dst[k].p0 = dy1 * width_step + dx1; dst[k].p1 = dy2 * width_step + dx1;
        dst[k].p2 = dy1 * width_step + dx2; dst[k].p3 = dy2 * width_step + dx2;
        dst[k].w = src[k][4] / ((float)(dx2 - dx1) * (dy2 - dy1));
    }
}

static inline float calc_haar(const int *origin, const surf_hf *f, int n)
{
    double d = 0;
    for (int k = 0; k < n; ++k) d += (origin[f[k].p0] + origin[f[k].p3] - origin[f[k].p1] - origin[f[k].p2]) * f[k].w;
    return (float)d;
}

typedef struct { float x, y, size, angle, response; int octave, class_id; } surf_kp;

static int kp_greater(const void *pa, const void *pb)
{
    const surf_kp *a = (const surf_kp *)pa, *b = (const surf_kp *)pb;
    if (a->response != b->response) return a->response > b->response ? -1 : 1;
    if (a->size != b->size) return a->size > b->size ? -1 : 1;
    if (a->octave != b->octave) return a->octave > b->octave ? -1 : 1;
    if (a->y != b->y) return a->y > b->y ? -1 : 1;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    return 0;
}

/* 3 x 3 solve by LU with partial pivoting in float (Matx33f::solve(b, DECOMP_LU)); returns 0 when singular */
static int solve3f(float A[3][3], float b[3], float x[3])
{
    for (int i = 0; i < 3; ++i) {
        int k = i;
        for (int j = i + 1; j < 3; ++j) if (fabsf(A[j][i]) > fabsf(A[k][i])) k = j;
        if (fabsf(A[k][i]) < FLT_EPSILON) return 0;
        if (k != i) { for (int j = i; j < 3; ++j) { float t = A[i][j]; A[i][j] = A[k][j]; A[k][j] = t; } float t = b[i]; b[i] = b[k]; b[k] = t; }
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
    return 1;
}

#define N_OCT 4
#define N_LAYERS 3
#define ORI_RADIUS 6
#define PATCH_SZ 20

/* getGaussianKernel(n, sigma, CV_32F) for sigma > 0 */
static void gaussian_kernel(int n, double sigma, float *out)
{
    double sum = 0, tmp[32];
    const double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; tmp[i] = exp(scale2x * x * x); sum += tmp[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) out[i] = (float)(tmp[i] * sum);
}

/* gray: rows x cols uint8.  Returns the number of keypoints written (at most max_kp, strongest first);
 * kp_out: 7 floats per keypoint (x, y, size, angle, response, octave, class_id); desc_out: 64 floats per keypoint. */
int esfm_ref_surf(const uint8_t *gray, int rows, int cols, double hessian_threshold, int max_kp, float *kp_out, float *desc_out)
{
    const int sr = rows + 1, sc = cols + 1;
    int *sum = (int *)calloc((size_t)sr * sc, sizeof(int));
    for (int y = 0; y < rows; ++y) {
        int rs = 0;
        for (int x = 0; x < cols; ++x) { rs += gray[(size_t)y * cols + x]; sum[(size_t)(y + 1) * sc + x + 1] = sum[(size_t)y * sc + x + 1] + rs; }
    }
    /* ---- fastHessianDetector */
    const int n_total = (N_LAYERS + 2) * N_OCT;
    float *dets[20], *traces[20]; int sizes[20], steps[20], lrows[20], lcols[20];
    int step = 1, index = 0;
    for (int oct = 0; oct < N_OCT; ++oct) {
        for (int layer = 0; layer < N_LAYERS + 2; ++layer) {
            lrows[index] = (sr - 1) / step; lcols[index] = (sc - 1) / step;
            dets[index] = (float *)calloc((size_t)(lrows[index] > 0 ? lrows[index] : 1) * (lcols[index] > 0 ? lcols[index] : 1), sizeof(float));
            traces[index] = (float *)calloc((size_t)(lrows[index] > 0 ? lrows[index] : 1) * (lcols[index] > 0 ? lcols[index] : 1), sizeof(float));
            sizes[index] = (9 + 6 * layer) << oct; steps[index] = step;
            ++index;
        }
        step *= 2;
    }
    static const int dx_s[3][5] = { {0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1} };
    static const int dy_s[3][5] = { {2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1} };
    static const int dxy_s[4][5] = { {1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1} };
    for (int L = 0; L < n_total; ++L) {
        const int size = sizes[L], st = steps[L];
        if (size > sr - 1 || size > sc - 1) continue;
        surf_hf Dx[3], Dy[3], Dxy[4];
        resize_haar(dx_s, Dx, 3, 9, size, sc); resize_haar(dy_s, Dy, 3, 9, size, sc); resize_haar(dxy_s, Dxy, 4, 9, size, sc);
        const int samples_i = 1 + (sr - 1 - size) / st, samples_j = 1 + (sc - 1 - size) / st, margin = (size / 2) / st;
        for (int i = 0; i < samples_i; ++i) {
            const int *sp = sum + (size_t)(i * st) * sc;
            float *dp = dets[L] + (size_t)(i + margin) * lcols[L] + margin, *tp = traces[L] + (size_t)(i + margin) * lcols[L] + margin;
            for (int j = 0; j < samples_j; ++j) {
                const float dx = calc_haar(sp, Dx, 3), dy = calc_haar(sp, Dy, 3), dxy = calc_haar(sp, Dxy, 4);
                sp += st;
                dp[j] = dx * dy - 0.81f * dxy * dxy; tp[j] = dx + dy;
            }
        }
    }
    size_t cap = 4096, nk = 0;
    surf_kp *kps = (surf_kp *)malloc(sizeof(surf_kp) * cap);
    for (int oct = 0; oct < N_OCT; ++oct)
        for (int layer = 1; layer <= N_LAYERS; ++layer) {
            const int L = oct * (N_LAYERS + 2) + layer;
            const int size = sizes[L], st = steps[L];
            const int layer_rows = (sr - 1) / st, layer_cols = (sc - 1) / st;
            const int margin = (sizes[L + 1] / 2) / st + 1;
            const int stp = lcols[L];
            for (int i = margin; i < layer_rows - margin; ++i)
                for (int j = margin; j < layer_cols - margin; ++j) {
                    const float val0 = dets[L][(size_t)i * stp + j];
                    if (!(val0 > (float)hessian_threshold)) continue;
                    const int sum_i = st * (i - (size / 2) / st), sum_j = st * (j - (size / 2) / st);
                    float N9[3][9];
                    for (int l = 0; l < 3; ++l) {
                        const float *d = dets[L - 1 + l] + (size_t)i * stp + j;
                        N9[l][0] = d[-stp - 1]; N9[l][1] = d[-stp]; N9[l][2] = d[-stp + 1]; N9[l][3] = d[-1]; N9[l][4] = d[0]; N9[l][5] = d[1];
                        N9[l][6] = d[stp - 1]; N9[l][7] = d[stp]; N9[l][8] = d[stp + 1];
                    }
                    int is_max = 1;
                    for (int l = 0; l < 3 && is_max; ++l) for (int q = 0; q < 9; ++q) { if (l == 1 && q == 4) continue; if (!(val0 > N9[l][q])) { is_max = 0; break; } }
                    if (!is_max) continue;
                    const float center_i = sum_i + (size - 1) * 0.5f, center_j = sum_j + (size - 1) * 0.5f;
                    surf_kp kp; kp.x = center_j; kp.y = center_i; kp.size = (float)size; kp.angle = -1; kp.response = val0; kp.octave = oct;
                    const float tr = traces[L][(size_t)i * stp + j];
                    kp.class_id = (tr > 0) - (tr < 0);
                    const int ds = size - sizes[L - 1];
                    float b[3] = { -(N9[1][5] - N9[1][3]) / 2, -(N9[1][7] - N9[1][1]) / 2, -(N9[2][4] - N9[0][4]) / 2 };
                    float A[3][3] = { { N9[1][3] - 2 * N9[1][4] + N9[1][5], (N9[1][8] - N9[1][6] - N9[1][2] + N9[1][0]) / 4, (N9[2][5] - N9[2][3] - N9[0][5] + N9[0][3]) / 4 },
                                      { (N9[1][8] - N9[1][6] - N9[1][2] + N9[1][0]) / 4, N9[1][1] - 2 * N9[1][4] + N9[1][7], (N9[2][7] - N9[2][1] - N9[0][7] + N9[0][1]) / 4 },
                                      { (N9[2][5] - N9[2][3] - N9[0][5] + N9[0][3]) / 4, (N9[2][7] - N9[2][1] - N9[0][7] + N9[0][1]) / 4, N9[0][4] - 2 * N9[1][4] + N9[2][4] } };
                    float x[3] = {0, 0, 0};
                    if (!solve3f(A, b, x)) continue;
                    const int ok = (x[0] != 0 || x[1] != 0 || x[2] != 0) && fabsf(x[0]) <= 1 && fabsf(x[1]) <= 1 && fabsf(x[2]) <= 1;
                    if (!ok) continue;
                    kp.x += x[0] * st; kp.y += x[1] * st; kp.size = (float)cv_round_f(kp.size + x[2] * ds);
                    if (nk == cap) { cap *= 2; kps = (surf_kp *)realloc(kps, sizeof(surf_kp) * cap); }
                    kps[nk++] = kp;
                }
        }
    qsort(kps, nk, sizeof(surf_kp), kp_greater);
    /* ---- orientation + descriptor (SURFInvoker) */
    float G_ori[2 * ORI_RADIUS + 1], G_desc[PATCH_SZ], DW[PATCH_SZ * PATCH_SZ], aptw[169];
    int aptx[169], apty[169], n_ori = 0;
    gaussian_kernel(2 * ORI_RADIUS + 1, 2.5, G_ori);
    for (int i = -ORI_RADIUS; i <= ORI_RADIUS; ++i) for (int j = -ORI_RADIUS; j <= ORI_RADIUS; ++j)
        if (i * i + j * j <= ORI_RADIUS * ORI_RADIUS) { aptx[n_ori] = i; apty[n_ori] = j; aptw[n_ori++] = G_ori[i + ORI_RADIUS] * G_ori[j + ORI_RADIUS]; }
    gaussian_kernel(PATCH_SZ, 3.3, G_desc);
    for (int i = 0; i < PATCH_SZ; ++i) for (int j = 0; j < PATCH_SZ; ++j) DW[i * PATCH_SZ + j] = G_desc[i] * G_desc[j];
    static const int gx_s[2][5] = { {0, 0, 2, 4, -1}, {2, 0, 4, 4, 1} }, gy_s[2][5] = { {0, 0, 4, 2, 1}, {0, 2, 4, 4, -1} };
    int n_out = 0;
    uint8_t *win = NULL; size_t win_cap = 0;
    for (size_t k = 0; k < nk && n_out < max_kp; ++k) {
        surf_kp kp = kps[k];
        const float s = kp.size * 1.2f / 9.0f;
        const int grad_wav_size = 2 * cv_round_f(2 * s);
        if (sr < grad_wav_size || sc < grad_wav_size) continue;
        surf_hf dx_t[2], dy_t[2];
        resize_haar(gx_s, dx_t, 2, 4, grad_wav_size, sc); resize_haar(gy_s, dy_t, 2, 4, grad_wav_size, sc);
        float X[169], Y[169], angle[169];
        int nangle = 0;
        for (int kk = 0; kk < n_ori; ++kk) {
            const int x = cv_round_f(kp.x + aptx[kk] * s - (float)(grad_wav_size - 1) / 2), y = cv_round_f(kp.y + apty[kk] * s - (float)(grad_wav_size - 1) / 2);
            if (y < 0 || y >= sr - grad_wav_size || x < 0 || x >= sc - grad_wav_size) continue;
            const int *ptr = sum + (size_t)y * sc + x;
            const float vx = calc_haar(ptr, dx_t, 2), vy = calc_haar(ptr, dy_t, 2);
            X[nangle] = vx * aptw[kk]; Y[nangle] = vy * aptw[kk]; ++nangle;
        }
        if (nangle == 0) continue;
        for (int j = 0; j < nangle; ++j) angle[j] = fast_atan2(Y[j], X[j]);
        float bestx = 0, besty = 0, descriptor_mod = 0;
        for (int i = 0; i < 360; i += 5) {
            float sumx = 0, sumy = 0;
            for (int j = 0; j < nangle; ++j) { const int d = abs(cv_round_f(angle[j]) - i); if (d < 30 || d > 330) { sumx += X[j]; sumy += Y[j]; } }
            const float temp_mod = sumx * sumx + sumy * sumy;
            if (temp_mod > descriptor_mod) { descriptor_mod = temp_mod; bestx = sumx; besty = sumy; }
        }
        float descriptor_dir = fast_atan2(-besty, bestx);
        kp.angle = descriptor_dir;
        /* window */
        const int win_size = (int)((PATCH_SZ + 1) * s);
        if ((size_t)win_size * win_size > win_cap) { win_cap = (size_t)win_size * win_size; win = (uint8_t *)realloc(win, win_cap); }
        descriptor_dir *= (float)(3.14159265358979323846 / 180);
        const float sin_dir = -(float)sin((double)descriptor_dir), cos_dir = (float)cos((double)descriptor_dir);
        const float win_offset = -(float)(win_size - 1) / 2;
        float start_x = kp.x + win_offset * cos_dir + win_offset * sin_dir, start_y = kp.y - win_offset * sin_dir + win_offset * cos_dir;
        const int ncols1 = cols - 1, nrows1 = rows - 1;
        for (int i = 0; i < win_size; ++i, start_x += sin_dir, start_y += cos_dir) {
            double pixel_x = start_x, pixel_y = start_y;
            for (int j = 0; j < win_size; ++j, pixel_x += cos_dir, pixel_y -= sin_dir) {
                const int ix = (int)floor(pixel_x), iy = (int)floor(pixel_y);
                if ((unsigned)ix < (unsigned)ncols1 && (unsigned)iy < (unsigned)nrows1) {
                    const float a = (float)(pixel_x - ix), b = (float)(pixel_y - iy);
                    const uint8_t *p = gray + (size_t)iy * cols + ix;
                    win[(size_t)i * win_size + j] = (uint8_t)cv_round_f(p[0] * (1.f - a) * (1.f - b) + p[1] * a * (1.f - b) + p[cols] * (1.f - a) * b + p[cols + 1] * a * b);
                } else {
                    int x = cv_round_d(pixel_x), y = cv_round_d(pixel_y);
                    x = x < 0 ? 0 : (x > ncols1 ? ncols1 : x); y = y < 0 ? 0 : (y > nrows1 ? nrows1 : y);
                    win[(size_t)i * win_size + j] = gray[(size_t)y * cols + x];
                }
            }
        }
        /* INTER_AREA shrink to 21 x 21 (general weighted form): out = sum_y beta_y (sum_x alpha_x S[y][x]) */
        uint8_t PATCH[PATCH_SZ + 1][PATCH_SZ + 1];
        {
            const int D = PATCH_SZ + 1;
            const double scale = (double)win_size / D;
            int tsi[21][64], tn[21]; float talpha[21][64];
            for (int dx = 0; dx < D; ++dx) {
                const double fsx1 = dx * scale, fsx2 = fsx1 + scale, cell = fmin(scale, win_size - fsx1);
                int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2), n = 0;
                sx2 = sx2 < win_size - 1 ? sx2 : win_size - 1; sx1 = sx1 < sx2 ? sx1 : sx2;
                if (sx1 - fsx1 > 1e-3) { tsi[dx][n] = sx1 - 1; talpha[dx][n++] = (float)((sx1 - fsx1) / cell); }
                for (int sx = sx1; sx < sx2; ++sx) { tsi[dx][n] = sx; talpha[dx][n++] = (float)(1.0 / cell); }
                if (fsx2 - sx2 > 1e-3) { tsi[dx][n] = sx2; talpha[dx][n++] = (float)(fmin(fmin(fsx2 - sx2, 1.), cell) / cell); }
                tn[dx] = n;
            }
            for (int dy = 0; dy < D; ++dy)
                for (int dx = 0; dx < D; ++dx) {
                    float acc = 0;
                    for (int a = 0; a < tn[dy]; ++a) {
                        const uint8_t *row = win + (size_t)tsi[dy][a] * win_size;
                        float buf = 0;
                        for (int b = 0; b < tn[dx]; ++b) buf += row[tsi[dx][b]] * talpha[dx][b];
                        acc += buf * talpha[dy][a];
                    }
                    int v = cv_round_f(acc);
                    PATCH[dy][dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
        }
        float DX[PATCH_SZ][PATCH_SZ], DY[PATCH_SZ][PATCH_SZ];
        for (int i = 0; i < PATCH_SZ; ++i) for (int j = 0; j < PATCH_SZ; ++j) {
            const float dw = DW[i * PATCH_SZ + j];
            DX[i][j] = (PATCH[i][j + 1] - PATCH[i][j] + PATCH[i + 1][j + 1] - PATCH[i + 1][j]) * dw;
            DY[i][j] = (PATCH[i + 1][j] - PATCH[i][j] + PATCH[i + 1][j + 1] - PATCH[i][j + 1]) * dw;
        }
        float *vec = desc_out + 64 * (size_t)n_out;
        for (int kk = 0; kk < 64; ++kk) vec[kk] = 0;
        double square_mag = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            float *v = vec + 4 * (4 * i + j);
            for (int y = i * 5; y < i * 5 + 5; ++y) for (int x = j * 5; x < j * 5 + 5; ++x) { const float tx = DX[y][x], ty = DY[y][x]; v[0] += tx; v[1] += ty; v[2] += (float)fabs(tx); v[3] += (float)fabs(ty); }
            for (int kk = 0; kk < 4; ++kk) square_mag += v[kk] * v[kk];
        }
        const float scl = (float)(1. / (sqrt(square_mag) + DBL_EPSILON));
        for (int kk = 0; kk < 64; ++kk) vec[kk] *= scl;
        float *ko = kp_out + 7 * (size_t)n_out;
        ko[0] = kp.x; ko[1] = kp.y; ko[2] = kp.size; ko[3] = kp.angle; ko[4] = kp.response; ko[5] = (float)kp.octave; ko[6] = (float)kp.class_id;
        ++n_out;
    }
    for (int L = 0; L < n_total; ++L) { free(dets[L]); free(traces[L]); }
    free(kps); free(sum); free(win);
    return n_out;
}
