// C-ABI entry point for image undistortion (include/esfm.h): the replacement for
// cv::undistort(rgb_image, out, K, distort_coeff) at reference cpp_code/src/estimate_motion.cpp:436.  Host side: the inverse of
// the (stripe-shifted) camera matrix and the three coordinate sequences OpenCV builds incrementally -- rows + cols doubles.
#include <algorithm>
#include <cstring>
#include <vector>

#include "undistort_kernels.hpp"

namespace {

// cv::invert, DECOMP_LU, n == 3: determinant and cofactors in OpenCV's order
void invert3(const double *S, double *t)
{
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

}  // namespace

extern "C" {

int esfm_undistort(esfm_ctx *ctx, const uint8_t *image, int rows, int cols, int channels, const double *K4, const double *dist4,
                   uint8_t *out)
{
    if (!ctx) { esfm::set_error("ctx is NULL"); return ESFM_ERR_INVALID_ARG; }
    ESFM_REQUIRE(image && out && K4 && dist4, "NULL argument");
    ESFM_REQUIRE(image != out, "cv::undistort does not work in place");
    ESFM_REQUIRE(rows > 0 && cols > 0 && (channels == 1 || channels == 3), "image must be rows x cols x {1, 3}");
    ESFM_REQUIRE(rows < 32768 && cols < 32768, "the 16-bit source coordinates of the maps limit the image to 32767 x 32767");
    if (int rc = esfm::set_device(ctx)) return rc;
    hipStream_t st = ctx->stream;

    esfm::UndistortParams P;
    memset(&P, 0, sizeof(P));
    P.rows = rows; P.cols = cols; P.channels = channels;
    P.fx = K4[0]; P.u0 = K4[1]; P.fy = K4[2]; P.v0 = K4[3];
    P.k1 = dist4[0]; P.k2 = dist4[1]; P.p1 = dist4[2]; P.p2 = dist4[3];

    // cv::undistort works in stripes of stripe0 rows, each with its own new camera matrix (cy' = cy - y0) and inverse; within a
    // stripe initUndistortRectifyMap accumulates _x, _y, _w along the row by += ir[0], ir[3], ir[6].  For a camera matrix without
    // skew ir[1] = ir[3] = ir[6] = ir[7] = +-0 exactly, so _y and _w are constant along a row and _x is the same in every row.
    const int stripe0 = std::min(std::max(1, (1 << 12) / cols), rows);
    std::vector<double> seq((size_t)cols + 2 * (size_t)rows);
    double *xseq = seq.data(), *yrow = xseq + cols, *wrow = yrow + rows;
    for (int y0 = 0; y0 < rows; y0 += stripe0) {
        const double Ar[9] = {P.fx, 0, P.u0, 0, P.fy, P.v0 - y0, 0, 0, 1};
        double ir[9];
        invert3(Ar, ir);
        ESFM_REQUIRE(ir[1] == 0 && ir[3] == 0 && ir[6] == 0 && ir[7] == 0, "camera matrix is not finite");
        const int sr = std::min(stripe0, rows - y0);
        for (int i = 0; i < sr; ++i) {
            const double _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
            const double w = 1. / _w;
            wrow[y0 + i] = w;
            yrow[y0 + i] = _y * w;
        }
        if (y0 == 0) {
            double _x = 0 * ir[1] + ir[2];
            for (int j = 0; j < cols; ++j, _x += ir[0]) xseq[j] = _x;
        }
    }

    const size_t n_bytes = (size_t)rows * cols * channels;
    esfm::DevBuf &b_src = ctx->stage_a, &b_dst = ctx->stage_b, &b_seq = ctx->stage_c;
    if (int rc = b_src.reserve(n_bytes + 16)) return rc;
    if (int rc = b_dst.reserve(n_bytes + 16)) return rc;
    if (int rc = b_seq.reserve(sizeof(double) * seq.size())) return rc;
    ESFM_HIP_TRY(esfm::copy_h2d(b_src.ptr, image, n_bytes, st));
    ESFM_HIP_TRY(esfm::copy_h2d(b_seq.ptr, seq.data(), sizeof(double) * seq.size(), st));
    const double *d_seq = b_seq.as<double>();
    {
        esfm::KernelTimer tm(ctx, ESFM_K_UNDISTORT);
        if (int rc = esfm::launch_undistort(st, P, b_src.as<uint8_t>(), d_seq, d_seq + cols, d_seq + cols + rows, b_dst.as<uint8_t>())) return rc;
    }
    ESFM_HIP_TRY(esfm::copy_d2h(out, b_dst.ptr, n_bytes, st));
    ESFM_HIP_TRY(hipStreamSynchronize(st));
    return ESFM_OK;
}

}  // extern "C"
