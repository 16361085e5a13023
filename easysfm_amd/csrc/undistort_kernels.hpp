// cv::undistort on the GPU (undistort part).  Launcher for undistort_kernels.hip.
#pragma once
#include "common.hpp"

namespace esfm {

struct UndistortParams {
    int rows, cols, channels, pad;
    double fx, u0, fy, v0;      // camera matrix (also the new camera matrix)
    double k1, k2, p1, p2;
};

// xseq[cols]: the normalised x numerator accumulated along a row as OpenCV does; wrow[rows] = 1 / _w and yrow[rows] = _y * w:
// the reciprocal homogeneous coordinate and the normalised y of each row, from the inverse camera matrix of the stripe the row
// belongs to.  All three are computed on the host (rows + cols doubles).
int launch_undistort(hipStream_t st, const UndistortParams &P, const uint8_t *src, const double *xseq, const double *yrow,
                     const double *wrow, uint8_t *dst);

}  // namespace esfm
