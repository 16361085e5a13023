// cv::undistort (initUndistortRectifyMap in CV_16SC2 form + remap INTER_LINEAR / BORDER_CONSTANT) fused into one pass over
// the destination image: reference cpp_code/src/estimate_motion.cpp:431-441 (MotionEstimator::doUnDistort).
// One thread per destination pixel.  The distortion model runs in double with OpenCV's operation order (the library is built
// with -ffp-contract=off; f64 division and sqrt are IEEE on gfx950), the source position is rounded to 1/32 pixel exactly as
// the 16SC2 maps do, and the bilinear blend uses the 15-bit integer weights of OpenCV's table, so the result is bit-exact
// against the CPU restatement.  The kernel is HBM-bound byte work: 3 B written and about 3 B read per pixel (the four taps
// of neighbouring pixels share cache lines).
#include "undistort_kernels.hpp"

namespace esfm {

namespace {

__device__ __forceinline__ int round_sat(double v)
{
    // cvRound on x86 (cvtsd2si): round half to even, "integer indefinite" outside the int range and for NaN
    if (!(v < 2147483647.5 && v >= -2147483648.5)) return (int)0x80000000;
    return (int)__double2ll_rn(v);
}

// source position of destination pixel (i, j): 1/32-pixel fixed point, as the CV_16SC2 maps hold it
__device__ __forceinline__ void undistort_source(const UndistortParams &P, double xs, double w, double y, int &sx, int &sy, int &ax, int &ay)
{
    // w = 1 / _w and y = _y * w depend on the row only and come from the host (IEEE division there and here give the same
    // bits).  OpenCV divides the radial numerator by 1 + ((k6 r2 + k5) r2 + k4) r2, which is exactly 1 for a 4-coefficient
    // model and finite r2 (and NaN together with the numerator otherwise), so the division is dropped.
    const double x = xs * w;
    const double x2 = x * x, y2 = y * y;
    const double r2 = x2 + y2, _2xy = 2 * x * y;
    const double kr = 1 + ((0 * r2 + P.k2) * r2 + P.k1) * r2;
    const double xd = (x * kr + P.p1 * _2xy + P.p2 * (r2 + 2 * x2) + 0 * r2 + 0 * r2 * r2);
    const double yd = (y * kr + P.p1 * (r2 + 2 * y2) + P.p2 * _2xy + 0 * r2 + 0 * r2 * r2);
    const double u = P.fx * 1. * xd + P.u0;
    const double v = P.fy * 1. * yd + P.v0;
    const int iu = round_sat(u * 32), iv = round_sat(v * 32);
    sx = (short)(iu >> 5); sy = (short)(iv >> 5);       // the maps hold shorts
    ax = iu & 31; ay = iv & 31;
}

// one destination pixel, any position: the four taps byte by byte, BORDER_CONSTANT zeros outside the image
template <int CH>
__device__ __forceinline__ void undistort_pixel_bytes(const UndistortParams &P, const uint8_t *__restrict__ src, int sx, int sy, int ax, int ay, uint8_t (&o)[CH])
{
#pragma unroll
    for (int c = 0; c < CH; ++c) o[c] = 0;
    if (sx >= P.cols || sx + 1 < 0 || sy >= P.rows || sy + 1 < 0) return;
    const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
    const bool x0 = sx >= 0, x1 = sx + 1 < P.cols, y0 = sy >= 0, y1 = sy + 1 < P.rows;
    const uint8_t *S0 = src + ((ptrdiff_t)sy * P.cols + sx) * CH;
    const uint8_t *S1 = S0 + (size_t)P.cols * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int v00 = (x0 && y0) ? S0[c] : 0, v01 = (x1 && y0) ? S0[CH + c] : 0;
        const int v10 = (x0 && y1) ? S1[c] : 0, v11 = (x1 && y1) ? S1[CH + c] : 0;
        const int s = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
        o[c] = (uint8_t)((s + (1 << 14)) >> 15);      // the weights sum to 32768: always within 0..255
    }
}

// One thread per destination pixel (any width, one or three channels).
template <int CH>
__global__ __launch_bounds__(256) void undistort_remap_kernel(UndistortParams P, const uint8_t *__restrict__ src, const double *__restrict__ xseq,
                                                              const double *__restrict__ yrow, const double *__restrict__ wrow,
                                                              uint8_t *__restrict__ dst)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= P.cols) return;
    int sx, sy, ax, ay;
    undistort_source(P, xseq[j], wrow[i], yrow[i], sx, sy, ax, ay);
    uint8_t o[CH];
    undistort_pixel_bytes<CH>(P, src, sx, sy, ax, ay, o);
    uint8_t *D = dst + ((size_t)i * P.cols + j) * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) D[c] = o[c];
}

// Three channels, width a multiple of four: FOUR destination pixels per thread.  A pixel whose four taps lie inside the image reads
// them as two 8-byte loads (the two source pixels of a row are six consecutive bytes; any alignment; the source buffer has 16 spare
// bytes behind the image) instead of twelve byte loads, and the thread's twelve result bytes leave in one 12-byte store (a wave
// writes 768 contiguous bytes) instead of twelve byte stores: the byte accesses were what the one-pixel form spent its time on
// (55 us for 37.7 MB: 0.086 of HBM).  Same arithmetic, same bits.
__global__ __launch_bounds__(256) void undistort_remap4_kernel(UndistortParams P, const uint8_t *__restrict__ src, const double *__restrict__ xseq,
                                                               const double *__restrict__ yrow, const double *__restrict__ wrow,
                                                               uint8_t *__restrict__ dst)
{
    const int j4 = (blockIdx.x * 256 + threadIdx.x) * 4, i = blockIdx.y;
    if (j4 >= P.cols) return;
    const double w = wrow[i], y = yrow[i];
    uint32_t packed[3] = {0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int sx, sy, ax, ay;
        undistort_source(P, xseq[j4 + u], w, y, sx, sy, ax, ay);
        uint8_t o[3];
        if (sx >= 0 && sx + 1 < P.cols && sy >= 0 && sy + 1 < P.rows) {
            const uint8_t *S0 = src + ((size_t)sy * P.cols + sx) * 3;
            uint64_t r0, r1;
            __builtin_memcpy(&r0, S0, 8); __builtin_memcpy(&r1, S0 + (size_t)P.cols * 3, 8);
            const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int v00 = (int)((r0 >> (8 * c)) & 255), v01 = (int)((r0 >> (8 * (3 + c))) & 255);
                const int v10 = (int)((r1 >> (8 * c)) & 255), v11 = (int)((r1 >> (8 * (3 + c))) & 255);
                const int s = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
                o[c] = (uint8_t)((s + (1 << 14)) >> 15);
            }
        } else {
            undistort_pixel_bytes<3>(P, src, sx, sy, ax, ay, o);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int b = 3 * u + c;
            packed[b >> 2] |= (uint32_t)o[c] << (8 * (b & 3));
        }
    }
    uint32_t *D = reinterpret_cast<uint32_t *>(dst + ((size_t)i * P.cols + j4) * 3);      // (cols % 4 == 0: a multiple of twelve bytes)
    D[0] = packed[0]; D[1] = packed[1]; D[2] = packed[2];
}

}  // namespace

int launch_undistort(hipStream_t st, const UndistortParams &P, const uint8_t *src, const double *xseq, const double *yrow,
                     const double *wrow, uint8_t *dst)
{
    if (P.rows <= 0 || P.cols <= 0) return ESFM_OK;
    const dim3 grid((unsigned)((P.cols + 255) / 256), (unsigned)P.rows);
    if (P.channels == 3 && P.cols % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0)
        hipLaunchKernelGGL(undistort_remap4_kernel, dim3((unsigned)((P.cols / 4 + 255) / 256), (unsigned)P.rows), dim3(256), 0, st, P, src, xseq, yrow, wrow, dst);
    else if (P.channels == 3)
        hipLaunchKernelGGL(undistort_remap_kernel<3>, grid, dim3(256), 0, st, P, src, xseq, yrow, wrow, dst);
    else
        hipLaunchKernelGGL(undistort_remap_kernel<1>, grid, dim3(256), 0, st, P, src, xseq, yrow, wrow, dst);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm
