/*
 * oracle/orb_ref.c -- CPU restatement of the ORB detector + 256-bit descriptor the reference extracts with feature type 'O'
 * (cv::ORB::create(max_num)->detect + ->compute, reference cpp_code/src/feature_matching.cpp:14-41, called at
 * cpp_code/test/sfm.cpp:116).  SURVEY.md section 8 row f-2 (ORB half).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link, call or execute this file
 * (see oracle/match_ref.c header).
 *
 * PARITY UNPINNED: OpenCV is absent here and the reference holds no fixture.  Restated from memory of OpenCV 3.4
 * [upstream modules/features2d/src/orb.cpp, fast.cpp, fast_score.cpp, keypoint.cpp; modules/imgproc resize / smooth],
 * cv::ORB::create(nfeatures) defaults: scaleFactor 1.2, nlevels 8, edgeThreshold 31, firstLevel 0, WTA_K 2, HARRIS_SCORE,
 * patchSize 31, fastThreshold 20.
 *   pyramid     level l has size cvRound(cols / 1.2^l) x cvRound(rows / 1.2^l), each level resized from the one above it
 *               (bilinear, source coordinate (x + 0.5) s - 0.5, weights rounded to 1/256, ((h0 (256 - ay) + h1 ay) + 2^15) >> 16,
 *               edge samples replicated), BORDER_REFLECT_101 outside
 *   quota       n_l = cvRound(N (1 - f) / (1 - f^8) f^l), f = 1 / 1.2, the last level takes the remainder
 *   FAST        9 of 16, threshold 20, score = (largest t for which the pixel is still a corner) = max over the 16 arcs of 9 of
 *               min |difference| - 1, strict 3 x 3 non-maximum suppression, keypoints within 31 px of the border dropped,
 *               the best 2 n_l by FAST score kept (all ties with the last kept one stay, as KeyPointsFilter::retainBest does)
 *   Harris      7 x 7 block of Sobel-like 3 x 3 gradients, ((float)a b - (float)c c - 0.04 ((float)a + b)^2) / (4 * 7 * 255)^4;
 *               the best n_l kept (ties as above)
 *   angle       intensity centroid over the radius-15 disc (OpenCV's umax table), cv::fastAtan2(m01, m10) in degrees
 *   keypoint    pt = level pt * 1.2^l (float), size = 31 * 1.2^l, octave = l, response = Harris, class_id = -1
 *   descriptor  level image blurred 7 x 7, sigma 2 (weights rounded to 1/256 with the centre taking the remainder, separable,
 *               (sum + 2^15) >> 16, BORDER_REFLECT_101); centre = cvRound(pt / 1.2^l); bit = I(R p0) < I(R p1) with
 *               R p = (cvRound(x a - y b), cvRound(x b + y a)), a = (float)cos, b = (float)sin of the angle in radians (float)
 * Deviations, documented: (1) the 256 x 2 test point pairs.  OpenCV's bit_pattern_31_ is a LEARNED table (rBRIEF) that ships
 * only inside OpenCV; the pairs here come from esfm_orb_pattern() below -- a seeded generator, isotropic, sigma = 31 / 5 clipped
 * to +-13 like the original BRIEF -- so descriptors match OpenCV's in kind, not bit for bit.  (2) keypoints come out level by
 * level ordered by (Harris response descending, y, x); OpenCV's order inside a level is whatever std::nth_element leaves.
 * (3) resize and blur use the fixed-point forms stated above (OpenCV's INTER_LINEAR_EXACT / fixed-point GaussianBlur are of this
 * kind; their exact rounding is not claimed).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORB_LEVELS 8
#define ORB_EDGE 31
#define ORB_HP 15
#define ORB_FAST_T 20

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

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

/* The 512 test points (x0, y0, x1, y1 per bit): sum of four uniforms (variance 1/3 each side) scaled to sigma 6.2, rounded,
 * clipped to [-13, 13]; a pair whose two points coincide is redrawn.  LCG: x <- 1664525 x + 1013904223 (mod 2^32), seed 31. */
void esfm_ref_orb_pattern(int8_t *out /* 1024 */)
{
    uint32_t s = 31u;
    int n = 0;
    while (n < 256) {
        int v[4];
        for (int k = 0; k < 4; ++k) {
            double acc = 0.0;
            for (int u = 0; u < 4; ++u) { s = s * 1664525u + 1013904223u; acc += (double)(s >> 8) / 16777216.0 - 0.5; }
            int q = cv_round_d(acc * (6.2 / 0.57735026918962576));   /* sum of 4 U(-1/2, 1/2): sigma = sqrt(4 / 12) */
            if (q > 13) q = 13;
            if (q < -13) q = -13;
            v[k] = q;
        }
        if (v[0] == v[2] && v[1] == v[3]) continue;
        for (int k = 0; k < 4; ++k) out[4 * n + k] = (int8_t)v[k];
        ++n;
    }
}

static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

static void resize_linear(const uint8_t *src, int sr, int sc, uint8_t *dst, int dr, int dc)
{
    const double fy = (double)sr / dr, fx = (double)sc / dc;
    int *x0 = (int *)malloc(sizeof(int) * dc), *ax = (int *)malloc(sizeof(int) * dc);
    for (int x = 0; x < dc; ++x) {
        double f = (x + 0.5) * fx - 0.5;
        int i = (int)floor(f);
        f -= i;
        if (i < 0) { i = 0; f = 0; }
        if (i >= sc - 1) { i = sc - 1; f = 0; }
        x0[x] = i; ax[x] = cv_round_d(f * 256.0);
    }
    for (int y = 0; y < dr; ++y) {
        double f = (y + 0.5) * fy - 0.5;
        int i = (int)floor(f);
        f -= i;
        if (i < 0) { i = 0; f = 0; }
        if (i >= sr - 1) { i = sr - 1; f = 0; }
        const int ay = cv_round_d(f * 256.0);
        const uint8_t *r0 = src + (size_t)i * sc, *r1 = src + (size_t)(i + 1 < sr ? i + 1 : i) * sc;
        for (int x = 0; x < dc; ++x) {
            const int j = x0[x], j1 = j + 1 < sc ? j + 1 : j, a = ax[x];
            const int h0 = r0[j] * (256 - a) + r0[j1] * a, h1 = r1[j] * (256 - a) + r1[j1] * a;
            dst[(size_t)y * dc + x] = (uint8_t)((h0 * (256 - ay) + h1 * ay + 32768) >> 16);
        }
    }
    free(x0); free(ax);
}

static void gauss7_weights(int w[7])
{
    double g[7], sum = 0;
    for (int i = 0; i < 7; ++i) { const double x = i - 3; g[i] = exp(-0.5 * x * x / 4.0); sum += g[i]; }
    int tot = 0;
    for (int i = 0; i < 7; ++i) { w[i] = cv_round_d(256.0 * g[i] / sum); tot += w[i]; }
    w[3] += 256 - tot;
}

static void blur7(const uint8_t *src, int rows, int cols, uint8_t *dst)
{
    int w[7];
    gauss7_weights(w);
    int *h = (int *)malloc(sizeof(int) * (size_t)rows * cols);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int k = -3; k <= 3; ++k) s += w[k + 3] * src[(size_t)y * cols + reflect101(x + k, cols)];
            h[(size_t)y * cols + x] = s;
        }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int s = 0;
            for (int k = -3; k <= 3; ++k) s += w[k + 3] * h[(size_t)reflect101(y + k, rows) * cols + x];
            dst[(size_t)y * cols + x] = (uint8_t)((s + 32768) >> 16);
        }
    free(h);
}

static const int ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* 0 when the pixel is not a FAST-9 corner at threshold t, else the corner score */
static int fast_score(const uint8_t *img, int cols, int x, int y, int t)
{
    const int v = img[(size_t)y * cols + x];
    int d[25];
    for (int k = 0; k < 16; ++k) d[k] = v - img[(size_t)(y + ring_dy[k]) * cols + x + ring_dx[k]];
    for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
    int best = 0;
    for (int s = 0; s < 16; ++s) {
        int mn = d[s], mx = d[s];
        for (int k = 1; k < 9; ++k) { if (d[s + k] < mn) mn = d[s + k]; if (d[s + k] > mx) mx = d[s + k]; }
        if (mn > best) best = mn;            /* all nine darker than the centre by at least mn */
        if (-mx > best) best = -mx;          /* all nine brighter by at least -mx */
    }
    return best > t ? best - 1 : 0;
}

typedef struct { int x, y, level; float resp; } orb_cand;

static int cand_cmp(const void *a, const void *b)
{
    const orb_cand *p = (const orb_cand *)a, *q = (const orb_cand *)b;
    if (p->resp != q->resp) return p->resp > q->resp ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return p->x < q->x ? -1 : (p->x > q->x ? 1 : 0);
}

/* KeyPointsFilter::retainBest on a list sorted by cand_cmp: the first n and every later one that ties with the n-th */
static int retain_best(const orb_cand *c, int count, int n)
{
    if (n >= count) return count;
    if (n <= 0) return 0;
    const float amb = c[n - 1].resp;
    int k = n;
    while (k < count && c[k].resp >= amb) ++k;
    return k;
}

static float harris_response(const uint8_t *img, int cols, int x0, int y0)
{
    int a = 0, b = 0, c = 0;
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) {
            const uint8_t *p = img + (size_t)(y0 + dy) * cols + x0 + dx;
            const int Ix = (p[1] - p[-1]) * 2 + (p[-cols + 1] - p[-cols - 1]) + (p[cols + 1] - p[cols - 1]);
            const int Iy = (p[cols] - p[-cols]) * 2 + (p[cols - 1] - p[-cols - 1]) + (p[cols + 1] - p[-cols + 1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    const float scale = 1.f / (4 * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

static void make_umax(int umax[ORB_HP + 2])
{
    const int vmax = (int)floor(ORB_HP * sqrt(2.0) / 2 + 1), vmin = (int)ceil(ORB_HP * sqrt(2.0) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt((double)ORB_HP * ORB_HP - v * v));
    for (int v = ORB_HP, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

static float ic_angle(const uint8_t *img, int cols, int x0, int y0, const int *umax)
{
    const uint8_t *center = img + (size_t)y0 * cols + x0;
    int m01 = 0, m10 = 0;
    for (int u = -ORB_HP; u <= ORB_HP; ++u) m10 += u * center[u];
    for (int v = 1; v <= ORB_HP; ++v) {
        int vsum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int vp = center[u + v * cols], vm = center[u - v * cols];
            vsum += vp - vm;
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    return fast_atan2((float)m01, (float)m10);
}

/* keypoints: 7 floats each (x, y, size, angle, response, octave, class_id); descriptors: 32 bytes each.  Returns the count. */
int esfm_ref_orb(const uint8_t *gray, int rows, int cols, int nfeatures, int max_kp, float *kp_out, uint8_t *desc_out)
{
    uint8_t *lvl[ORB_LEVELS], *blr[ORB_LEVELS];
    int lr[ORB_LEVELS], lc[ORB_LEVELS], quota[ORB_LEVELS];
    float scale[ORB_LEVELS];
    int8_t pattern[1024];
    int umax[ORB_HP + 2];
    esfm_ref_orb_pattern(pattern);
    make_umax(umax);
    for (int l = 0; l < ORB_LEVELS; ++l) {
        scale[l] = (float)pow(1.2, (double)l);
        lc[l] = cv_round_f(cols / scale[l]); lr[l] = cv_round_f(rows / scale[l]);
        if (lc[l] < 1) lc[l] = 1;
        if (lr[l] < 1) lr[l] = 1;
        lvl[l] = (uint8_t *)malloc((size_t)lr[l] * lc[l]);
        blr[l] = (uint8_t *)malloc((size_t)lr[l] * lc[l]);
        if (l == 0) memcpy(lvl[0], gray, (size_t)rows * cols);
        else resize_linear(lvl[l - 1], lr[l - 1], lc[l - 1], lvl[l], lr[l], lc[l]);
        blur7(lvl[l], lr[l], lc[l], blr[l]);
    }
    {
        const float factor = 1.f / 1.2f;
        float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)ORB_LEVELS));
        int sum = 0;
        for (int l = 0; l < ORB_LEVELS - 1; ++l) { quota[l] = cv_round_f(nd); sum += quota[l]; nd *= factor; }
        quota[ORB_LEVELS - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    }
    int n_out = 0;
    for (int l = 0; l < ORB_LEVELS; ++l) {
        const int R = lr[l], C = lc[l];
        if (R <= 2 * ORB_EDGE || C <= 2 * ORB_EDGE) continue;
        int *score = (int *)calloc((size_t)R * C, sizeof(int));
        for (int y = 3; y < R - 3; ++y)
            for (int x = 3; x < C - 3; ++x) score[(size_t)y * C + x] = fast_score(lvl[l], C, x, y, ORB_FAST_T);
        orb_cand *cand = (orb_cand *)malloc(sizeof(orb_cand) * ((size_t)R * C / 4 + 16));
        int nc = 0;
        for (int y = ORB_EDGE; y < R - ORB_EDGE; ++y)
            for (int x = ORB_EDGE; x < C - ORB_EDGE; ++x) {
                const int s = score[(size_t)y * C + x];
                if (s == 0) continue;
                const int *p = score + (size_t)y * C + x;
                if (s > p[-1] && s > p[1] && s > p[-C - 1] && s > p[-C] && s > p[-C + 1] && s > p[C - 1] && s > p[C] && s > p[C + 1]) {
                    cand[nc].x = x; cand[nc].y = y; cand[nc].level = l; cand[nc].resp = (float)s; ++nc;
                }
            }
        free(score);
        qsort(cand, (size_t)nc, sizeof(orb_cand), cand_cmp);
        nc = retain_best(cand, nc, 2 * quota[l]);
        for (int k = 0; k < nc; ++k) cand[k].resp = harris_response(lvl[l], C, cand[k].x, cand[k].y);
        qsort(cand, (size_t)nc, sizeof(orb_cand), cand_cmp);
        nc = retain_best(cand, nc, quota[l]);
        const float inv = 1.f / scale[l];
        for (int k = 0; k < nc && n_out < max_kp; ++k, ++n_out) {
            const float angle = ic_angle(lvl[l], C, cand[k].x, cand[k].y, umax);
            float *ko = kp_out + 7 * (size_t)n_out;
            const float px = cand[k].x * scale[l], py = cand[k].y * scale[l];
            ko[0] = px; ko[1] = py; ko[2] = 31.f * scale[l]; ko[3] = angle; ko[4] = cand[k].resp; ko[5] = (float)l; ko[6] = -1.f;
            const int cx = cv_round_f(px * inv), cy = cv_round_f(py * inv);
            float ang = angle;
            ang *= (float)(3.14159265358979323846 / 180.f);
            const float a = (float)cos(ang), b = (float)sin(ang);
            uint8_t *dsc = desc_out + 32 * (size_t)n_out;
            for (int by = 0; by < 32; ++by) {
                int val = 0;
                for (int bit = 0; bit < 8; ++bit) {
                    const int8_t *pp = pattern + 4 * (8 * by + bit);
                    const int x0 = cv_round_f(pp[0] * a - pp[1] * b), y0 = cv_round_f(pp[0] * b + pp[1] * a);
                    const int x1 = cv_round_f(pp[2] * a - pp[3] * b), y1 = cv_round_f(pp[2] * b + pp[3] * a);
                    const int t0 = blr[l][(size_t)reflect101(cy + y0, R) * C + reflect101(cx + x0, C)];
                    const int t1 = blr[l][(size_t)reflect101(cy + y1, R) * C + reflect101(cx + x1, C)];
                    val |= (t0 < t1) << bit;
                }
                dsc[by] = (uint8_t)val;
            }
        }
        free(cand);
    }
    for (int l = 0; l < ORB_LEVELS; ++l) { free(lvl[l]); free(blr[l]); }
    return n_out;
}
