// Host-side pieces of OpenCV's RANSAC point-set registrator shared by the essential-matrix and the PnP entry points
// [upstream opencv/modules/calib3d/src/ptsetreg.cpp, modules/core/include/opencv2/core/operations.hpp]: the cv::RNG
// recurrence, getSubset's sampling rule and RANSACUpdateNumIters.  Pure host arithmetic.
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace esfm {
namespace ransac {

constexpr int kModelPoints = 5;       // both the 5-point essential kernel and solvePnPRansac's EPnP kernel draw 5
constexpr int kMaxIters = 1000;       // RANSACPointSetRegistrator default (findEssentialMat does not override it)

struct CvRng {                        // cv::RNG((uint64)-1)
    uint64_t state = 0xFFFFFFFFFFFFFFFFull;
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// RANSACPointSetRegistrator::getSubset for a model without checkSubset: distinct indices, a repeat is redrawn
inline void draw_subset(CvRng &rng, int count, int32_t idx[kModelPoints])
{
    for (int i = 0; i < kModelPoints;) {
        int v;
        for (;;) {
            v = idx[i] = rng.uniform(0, count);
            int j = 0;
            for (; j < i; ++j) if (v == idx[j]) break;
            if (j == i) break;
        }
        ++i;
    }
}

inline int update_num_iters(double p, double ep, int model_points, int max_iters)   // cv::RANSACUpdateNumIters
{
    p = std::max(p, 0.0); p = std::min(p, 1.0);
    ep = std::max(ep, 0.0); ep = std::min(ep, 1.0);
    double num = std::max(1.0 - p, DBL_MIN);
    double denom = 1.0 - std::pow(1.0 - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num); denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

}  // namespace ransac
}  // namespace esfm
