// cv::undistort on the GPU (undistort part).  Launcher for undistort_kernels.hip.
#pragma once
#include "common.hpp"

namespace esfm {

struct UndistortParams {
    int rows, cols, channels, pad;
    double fx, u0, fy, v0;      // camera matrix (also the new camera matrix)
    double k1, k2, p1, p2;
};

// xseq[cols]: the normalised x numerator accumulated along a row as OpenCV does; yrow[rows], wrow[rows]: the per-row y
// numerator and homogeneous w of the stripe the row belongs to.  All three are computed on the host (rows + cols doubles).
int launch_undistort(hipStream_t st, const UndistortParams &P, const uint8_t *src, const double *xseq, const double *yrow,
                     const double *wrow, uint8_t *dst);

}  // namespace esfm
