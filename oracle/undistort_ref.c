/* TEST INFRASTRUCTURE ONLY -- CPU restatement of cv::undistort as MotionEstimator::doUnDistort calls it
 * (cpp_code/src/estimate_motion.cpp:431-441: cv::undistort(rgb_image, out, K, distort_coeff), once per imported frame,
 * cpp_code/test/sfm.cpp:97-98).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use this file.
 *
 * OpenCV is a dependency that is absent from /root/reference (OpenCV 3.x, unpinned), so this follows the
 * published algorithm of imgproc's undistort.cpp / imgwarp.cpp (3.4 line, scalar paths):
 *   cv::undistort           -- stripes of min(max(1, 4096 / cols), rows) rows; per stripe the new camera matrix gets
 *                              cy' = cy - y0 and initUndistortRectifyMap(A, dist, I, Ar, CV_16SC2) + remap(INTER_LINEAR,
 *                              BORDER_CONSTANT 0) are run on it;
 *   initUndistortRectifyMap -- iR = inv(Ar) by the closed 3 x 3 cofactor formula (cv::invert's n == 3 branch), the normalised
 *                              coordinate _x accumulated along the row (_x += iR[0] per pixel), the k1 k2 p1 p2 model in double,
 *                              u, v rounded to 1/32 pixel (cvRound(u * 32)) and split into integer part and a 5 + 5 bit fraction;
 *   remap                   -- 8-bit bilinear with 15-bit fixed-point weights (32 - fx)(32 - fy) * 32 ... (exact products, they
 *                              sum to 32768), (sum + 16384) >> 15, zero outside the image.
 * PARITY UNPINNED against OpenCV itself (no OpenCV in this image, the reference ships no fixtures); pinned against an
 * independent numpy restatement in tests/golden/make_golden.py (undistort_cases.npz). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static void invert3(const double *S, double *t)
{
    /* cv::invert, DECOMP_LU, n == 3: det3 and cofactors in this exact order */
    const double d0 = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d0 == 0.0) { memset(t, 0, 9 * sizeof(double)); return; }
    const double d = 1. / d0;
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

static int sat_int(double v)
{
    /* saturate_cast<int>(double) = cvRound = cvtsd2si on x86: round half to even; NaN and anything that rounds outside the
     * int range give the "integer indefinite" value 0x80000000 */
    if (!(v < 2147483647.5 && v >= -2147483648.5)) return INT32_MIN;
    return (int)lrint(v);
}

/* map1 (sx, sy shorts) and map2 (fy * 32 + fx) of one stripe of `srows` rows starting at image row y0 */
void esfm_ref_undistort_maps(int cols, int srows, int y0, const double *K4 /*fx cx fy cy*/, const double *dist /*k1 k2 p1 p2*/,
                             int16_t *map1, uint16_t *map2)
{
    const double fx = K4[0], u0 = K4[1], fy = K4[2], v0 = K4[3];
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
    const double Ar[9] = {fx, 0, u0, 0, fy, v0 - y0, 0, 0, 1};
    double ir[9];
    invert3(Ar, ir);
    for (int i = 0; i < srows; ++i) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < cols; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((0 * r2 + k2) * r2 + k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
            const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + 0 * r2 + 0 * r2 * r2);
            const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + 0 * r2 + 0 * r2 * r2);
            /* identity tilt: vecTilt = (xd, yd, 1), invProj = 1 */
            const double u = fx * 1. * xd + u0;
            const double v = fy * 1. * yd + v0;
            const int iu = sat_int(u * 32), iv = sat_int(v * 32);
            map1[2 * ((size_t)i * cols + j)] = (int16_t)(iu >> 5);
            map1[2 * ((size_t)i * cols + j) + 1] = (int16_t)(iv >> 5);
            map2[(size_t)i * cols + j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}

/* remapBilinear<FixedPtCast<int, uchar, 15>, short>, BORDER_CONSTANT 0, `ch` interleaved channels */
static void remap_rows(const uint8_t *src, int rows, int cols, int ch, const int16_t *map1, const uint16_t *map2, int drows, uint8_t *dst)
{
    for (int i = 0; i < drows; ++i)
        for (int j = 0; j < cols; ++j) {
            const int sx = map1[2 * ((size_t)i * cols + j)], sy = map1[2 * ((size_t)i * cols + j) + 1];
            const int f = map2[(size_t)i * cols + j], ax = f & 31, ay = f >> 5;
            const int w[4] = {(32 - ax) * (32 - ay) * 32, ax * (32 - ay) * 32, (32 - ax) * ay * 32, ax * ay * 32};
            uint8_t *D = dst + ((size_t)i * cols + j) * ch;
            if (sx >= cols || sx + 1 < 0 || sy >= rows || sy + 1 < 0) {
                for (int c = 0; c < ch; ++c) D[c] = 0;
                continue;
            }
            for (int c = 0; c < ch; ++c) {
                int v[4];
                for (int q = 0; q < 4; ++q) {
                    const int xx = sx + (q & 1), yy = sy + (q >> 1);
                    v[q] = (xx >= 0 && xx < cols && yy >= 0 && yy < rows) ? src[((size_t)yy * cols + xx) * ch + c] : 0;
                }
                const int s = v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
                const int r = (s + (1 << 14)) >> 15;
                D[c] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
        }
}

/* cv::undistort(src, dst, K, dist) for an 8-bit image of `ch` interleaved channels.  Returns 0, or -1 without memory. */
int esfm_ref_undistort(const uint8_t *src, int rows, int cols, int ch, const double *K4, const double *dist, uint8_t *dst)
{
    if (rows <= 0 || cols <= 0) return 0;
    int stripe0 = (1 << 12) / cols;
    if (stripe0 < 1) stripe0 = 1;
    if (stripe0 > rows) stripe0 = rows;
    int16_t *map1 = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)stripe0 * cols);
    uint16_t *map2 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)stripe0 * cols);
    if (!map1 || !map2) { free(map1); free(map2); return -1; }
    for (int y = 0; y < rows; y += stripe0) {
        const int sr = stripe0 < rows - y ? stripe0 : rows - y;
        esfm_ref_undistort_maps(cols, sr, y, K4, dist, map1, map2);
        remap_rows(src, rows, cols, ch, map1, map2, sr, dst + (size_t)y * cols * ch);
    }
    free(map1);
    free(map2);
    return 0;
}
