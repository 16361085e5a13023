/*
 * oracle/geometry_ref.c -- CPU restatement of the two-view triangulation the reference calls through OpenCV
 * (triangulation part).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link, call or execute this file
 * (see oracle/match_ref.c header).
 *
 * PARITY UNPINNED: cv::triangulatePoints is called at cpp_code/src/estimate_motion.cpp:263 (getDepthFast) and :333
 * (doTriangulation) with CV_32F 3x4 projection matrices and Point2f normalised image points (pixel2cam,
 * cpp_code/include/estimate_motion.h:41-46); OpenCV (>= 3, unpinned) is absent here and the reference holds no fixture.
 * Restated from memory [upstream opencv/modules/calib3d/src/triangulate.cpp cvTriangulatePoints]:
 *   per point i, in double:  A (4x4), rows 2j, 2j+1 for view j:
 *       A[2j][k]   = x_j * P_j[2][k] - P_j[0][k]
 *       A[2j+1][k] = y_j * P_j[2][k] - P_j[1][k]
 *   cvSVD(A, W, 0, V, CV_SVD_V_T); the homogeneous point is the right singular vector of the smallest singular value
 *   (row 3 of V^T), written to a CV_32F 4 x N matrix.
 * The singular vector is computed here by one-sided (Hestenes) Jacobi rotations on the columns of A, as OpenCV's
 * JacobiSVD does; it is unique up to sign, and the sign cancels in the de-homogenisation the callers perform
 * (estimate_motion.cpp:271, :341).  Output: out[4*i + 0..3] = X, Y, Z, W as floats.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

void esfm_ref_triangulate_points(const float *P1, const float *P2, const float *pts1, const float *pts2, int n, float *out)
{
    const float *P[2] = { P1, P2 };
    const float *pts[2] = { pts1, pts2 };
    for (int i = 0; i < n; ++i) {
        double A[4][4];
        for (int j = 0; j < 2; ++j) {
            const double x = (double)pts[j][2 * (size_t)i], y = (double)pts[j][2 * (size_t)i + 1];
            for (int k = 0; k < 4; ++k) {
                A[2 * j][k] = x * (double)P[j][8 + k] - (double)P[j][k];
                A[2 * j + 1][k] = y * (double)P[j][8 + k] - (double)P[j][4 + k];
            }
        }
        /* one-sided Jacobi: orthogonalise the columns of A, accumulating the rotations in V */
        double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
        for (int sweep = 0; sweep < 60; ++sweep) {
            int changed = 0;
            for (int p = 0; p < 3; ++p)
                for (int q = p + 1; q < 4; ++q) {
                    double a = 0, b = 0, c = 0;
                    for (int r = 0; r < 4; ++r) { a += A[r][p] * A[r][p]; b += A[r][q] * A[r][q]; c += A[r][p] * A[r][q]; }
                    if (fabs(c) <= DBL_EPSILON * sqrt(a * b) || c == 0.0) continue;
                    changed = 1;
                    const double zeta = (b - a) / (2.0 * c);
                    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                    for (int r = 0; r < 4; ++r) {
                        const double u = A[r][p], v = A[r][q];
                        A[r][p] = cs * u - sn * v; A[r][q] = sn * u + cs * v;
                        const double vu = V[r][p], vv = V[r][q];
                        V[r][p] = cs * vu - sn * vv; V[r][q] = sn * vu + cs * vv;
                    }
                }
            if (!changed) break;
        }
        int best = 0; double bn = DBL_MAX;
        for (int k = 0; k < 4; ++k) {
            double s = 0;
            for (int r = 0; r < 4; ++r) s += A[r][k] * A[r][k];
            if (s < bn) { bn = s; best = k; }
        }
        for (int r = 0; r < 4; ++r) out[4 * (size_t)i + r] = (float)V[r][best];
    }
}
